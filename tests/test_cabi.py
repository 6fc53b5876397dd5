"""CPU: libr3d_hip.so loads, exports every symbol include/r3d.h declares (and the ctypes table
covers the header one-to-one), and its no-GPU error behaviour is loud and well-formed.
No compute entry point is called here."""
import ctypes as C
import importlib
import os
import re

import numpy as np
import pytest

from helpers import PKG, ROOT


def header_symbols(names=("r3d.h", "r3d_internal_api.h")):
    """Every function include/*.h declares: the drop-in boundary (r3d.h) and the ICP driver's building blocks
    (r3d_internal_api.h)."""
    out = set()
    for name in names:
        text = open(os.path.join(ROOT, "include", name)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        out |= set(re.findall(r"\b(r3d_[a-z0-9_A-Z]+)\s*\(", text))
    return sorted(out)


def test_library_exports_every_header_symbol():
    L = importlib.import_module(PKG + "._lib")
    lib = L.load()
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), "libr3d_hip.so does not export %s" % s
    assert sorted(L.SIGNATURES) == syms, set(L.SIGNATURES) ^ set(syms)
    assert sorted(os.listdir(os.path.join(ROOT, "include"))) == ["r3d.h", "r3d_internal_api.h"]
    # the stable header stays small: what a drop-in host needs; selection / partial-sum / reordering helpers live in the other
    stable, internal = header_symbols(("r3d.h",)), header_symbols(("r3d_internal_api.h",))
    assert not set(stable) & set(internal) and len(internal) >= 15
    for s in ("r3d_selftest_magic_div", "r3d_permutation_invert", "r3d_remap_u32", "r3d_gather_rows_strided", "r3d_select_quantile_f32"):
        assert s in internal and s not in stable
    assert lib.r3d_version() == 200


def test_library_carries_gfx950_code_object():
    blob = open(os.path.join(ROOT, PKG, "libr3d_hip.so"), "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx90a", b"gfx942", b"sm_80"):
        assert other not in blob


def test_no_device_fails_loudly():
    """On a box without an MI355X the product raises; it never falls back to a CPU path."""
    r3d = importlib.import_module(PKG)
    L = importlib.import_module(PKG + "._lib")
    lib = L.load()
    n = C.c_int(-1)
    rc = lib.r3d_device_count(C.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is visible; the no-device behaviour is covered on the CPU box")
    assert rc == L.ERR_NODEVICE and n.value == 0
    with pytest.raises(r3d.R3DError) as e:
        r3d.Context(0)
    assert e.value.code == L.ERR_NODEVICE
    with pytest.raises(r3d.R3DError):
        r3d.fuse_frames(np.zeros((1, 4, 4), np.uint8), [[0, 0, 0, 1]], [[0, 0, 0]])
    assert "no HIP device" in L.last_error() or "device" in L.last_error()


def test_argument_errors_without_gpu():
    L = importlib.import_module(PKG + "._lib")
    lib = L.load()
    assert lib.r3d_ctx_create(0, None, 0, None) == L.ERR_INVALID
    assert lib.r3d_ctx_destroy(None) == 0
    assert lib.r3d_camera_destroy(None) == 0
    n = C.c_size_t()
    assert lib.r3d_format_ply(None, 7, 0, None, 0, C.byref(n)) == L.ERR_INVALID
    assert lib.r3d_format_ply(None, L.F32, -1, None, 0, C.byref(n)) == L.ERR_INVALID
    assert lib.r3d_write_ply(None, None, L.F32, 0) == L.ERR_INVALID


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under the package may import it."""
    for root, _, files in os.walk(os.path.join(ROOT, PKG)):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(root, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), fn
                assert "oracle/" not in text and "fusion_ref" not in text and "icp_ref" not in text, fn


def test_magic_division_is_exact():
    """The kernels split pixel -> (row, col) and tile -> (frame, tile) with host-made magic numbers: exact for every
    divisor >= 1 and every x < 2^31 (checked on edge values and 200k random pairs, both evaluation forms)."""
    L = importlib.import_module(PKG + "._lib")
    lib = L.load()
    rng = np.random.default_rng(0)
    q = C.c_uint32()
    ds = [1, 2, 3, 4, 5, 6, 7, 640, 1279, 1280, 1281, 1920, 4099, 65535, 65536, 65537, 2**30 - 1, 2**30, 2**30 + 1, 2**31 - 1]
    xs = [0, 1, 2, 1279, 1280, 491519, 491520, 2**31 - 2, 2**31 - 1]
    for d in ds:
        for x in xs + [d - 1, d, d + 1, 2 * d - 1, 2 * d, 3 * d + 1, (2**31 - 1) // d * d, (2**31 - 1) // d * d - 1]:
            if 0 <= x < 2**31:
                assert lib.r3d_selftest_magic_div(d, x, C.byref(q)) == 0, L.last_error()
                assert q.value == x // d, (d, x, q.value)
    dd = rng.integers(1, 2**31, size=200000)
    dd[:100000] = rng.integers(1, 5000, size=100000)          # realistic widths / tile counts
    xx = rng.integers(0, 2**31, size=200000)
    for d, x in zip(dd.tolist(), xx.tolist()):
        lib.r3d_selftest_magic_div(d, x, C.byref(q))
        if q.value != x // d:
            raise AssertionError((d, x, q.value, x // d))
    assert lib.r3d_selftest_magic_div(0, 1, C.byref(q)) == L.ERR_INVALID
