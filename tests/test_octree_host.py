"""CPU: the .bt serialiser (host C++ in libr3d_hip.so) against the OctoMap restatement in oracle/octomap_ref.py,
plus hand-derived known answers.  Parity unpinned by the reference (third-party library absent)."""
import importlib

import numpy as np
import pytest

from helpers import PKG
from oracle import octomap_ref as OM


@pytest.fixture(scope="module")
def V():
    return importlib.import_module(PKG + ".voxelmap")


def test_single_voxel_known_answer(V):
    codes, dropped = OM.occupied_set(np.array([[0.05, 0.05, 0.05]], np.float32))
    assert dropped == 0 and codes.tolist() == [7 << 45]             # keys (32768,)*3: only the top bit set
    data, nodes = V.format_bt(codes)
    assert nodes == 17                                              # root + 15 inner + 1 leaf
    head = (b"# Octomap OcTree binary file\n# (feel free to add / change comments, but leave the first line as it is!)\n#\n"
            b"id OcTree\nsize 17\nres 0.1\ndata\n")
    assert data == head + b"\x00\xc0" + b"\x03\x00" * 14 + b"\x02\x00"
    assert (data, nodes) == OM.write_bt_bytes(codes)


def test_full_octant_prunes(V):
    g = np.array([[x, y, z] for x in (0.05, 0.15) for y in (0.05, 0.15) for z in (0.05, 0.15)], np.float32)
    codes, _ = OM.occupied_set(g)
    data, nodes = V.format_bt(codes)
    assert nodes == 16                                              # the 8 leaves collapse into their parent
    assert data.endswith(b"\x03\x00" * 13 + b"\x02\x00")
    res, size, count, leaves = OM.read_bt_leaves(data)
    assert size == count == 16 and leaves == [(codes[0] >> 3, 15)]
    # seven of the eight: no pruning
    data7, nodes7 = V.format_bt(codes[:7])
    assert nodes7 == 16 + 7 and data7[-2:] == bytes([0xaa, 0x2a])


@pytest.mark.parametrize("seed,n,spread", [(0, 1, 1.0), (1, 200, 0.3), (2, 5000, 2.0), (3, 60000, 6.0), (4, 30000, 0.5)])
def test_serialiser_matches_oracle(V, seed, n, spread):
    rng = np.random.default_rng(seed)
    pts = (rng.normal(size=(n, 3)) * spread).astype(np.float32)
    codes, _ = OM.occupied_set(pts)
    data, nodes = V.format_bt(codes)
    want, want_nodes = OM.write_bt_bytes(codes)
    assert nodes == want_nodes and data == want
    res, size, count, leaves = OM.read_bt_leaves(data)
    assert res == 0.1 and size == count == nodes
    # expanding the leaves (pruned ones cover 8^(16-depth) voxels) gives back exactly the input set
    total = sum(8 ** (16 - d) for _, d in leaves)
    assert total == len(codes)


def test_empty_and_invalid(V):
    data, nodes = V.format_bt(np.zeros(0, np.uint64))
    assert nodes == 0 and data.endswith(b"size 0\nres 0.1\ndata\n")
    assert (data, nodes) == OM.write_bt_bytes(np.zeros(0, np.uint64))
    R = importlib.import_module(PKG)
    with pytest.raises(R.R3DError):
        V.format_bt(np.array([5, 5], np.uint64))                     # not strictly ascending
    with pytest.raises(R.R3DError):
        V.format_bt(np.array([1 << 50], np.uint64))                  # beyond 48 bits
    d2, _ = V.format_bt(np.array([0], np.uint64), resolution=0.25)
    assert b"res 0.25\n" in d2


def test_key_rules_of_the_oracle():
    # floor, float rounding of the coordinate, range check and non-finite points
    pts = np.array([[-0.05, 0.0, 0.1], [3276.75, 0, 0], [3276.85, 0, 0], [-3276.75, 0, 0], [-3276.9, 0, 0],
                    [np.nan, 0, 0], [np.inf, 0, 0]], np.float32)
    k, ok = OM.voxel_keys(pts)
    assert k[0].tolist() == [32767, 32768, 32769]                    # float32(0.1)*10 = 1.0000000149 -> 1
    assert ok.tolist() == [True, True, False, True, False, False, False]
    assert k[1, 0] == 65535 and k[3, 0] == 0
