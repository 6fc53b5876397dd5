"""GPU: f1 on the device (csrc/r3d_textfmt.hip) -- the reference's txt lines (`str(x)+','+str(y)+','+str(z)+'\\n'`,
camera_to_world.py:80-81, 103-104) and PLY rows ("%.4f %.4f %.4f \\n", :129-131) formatted by the GPU -- against the HOST
formatter of the same library (csrc/r3d_format.cpp), which tests/test_host_logic.py pins to Python's repr() and "%.4f" and to
the files the reference itself wrote (tests/golden).  Byte for byte: every binary exponent, powers of ten, notation switch
points, ties of the fifth decimal, subnormals, infinities and NaNs, 3 M random bit patterns, ragged segment sizes."""
import importlib
import os

import numpy as np
import pytest

from helpers import PKG

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    return importlib.import_module(PKG)


@pytest.fixture(scope="module")
def ctx(R):
    c = R.Context(0)
    yield c
    c.close()


def device_text(R, ctx, kind, xyz, aux=None, aux_code=0, segment_points=0):
    """(text bytes, segment offsets) of a host cloud through the device formatter."""
    T = R.device_text
    xyz = np.ascontiguousarray(xyz)
    n = xyz.shape[0]
    d_xyz = ctx.alloc(max(xyz.nbytes, 16)).upload(xyz)
    d_aux = ctx.alloc(max(aux.nbytes, 16)).upload(aux) if aux is not None else None
    try:
        total, offs = T.format_text(ctx, kind, d_xyz.ptr, xyz.dtype, n, d_aux.ptr if d_aux else None, aux_code, segment_points)
        d_text = ctx.alloc(total + 64)
        ctx.lib.r3d_memset(ctx.handle, d_text.ptr, 0x7e, total + 64)
        # the text at an odd offset of its buffer: the kernel aligns its stores to the ADDRESS, not to the offset
        got, offs2 = T.format_text(ctx, kind, d_xyz.ptr, xyz.dtype, n, d_aux.ptr if d_aux else None, aux_code, segment_points,
                                   d_text.ptr + 13, total)
        assert got == total and np.array_equal(offs, offs2)
        raw = d_text.download(np.uint8, total + 64)
        assert bytes(raw[:13]) == b"~" * 13 and bytes(raw[13 + total:]) == b"~" * (51)      # nothing outside its range
        d_text.free()
        return bytes(raw[13:13 + total]), offs
    finally:
        d_xyz.free()
        if d_aux:
            d_aux.free()


def host_ply_rows(R, xyz):
    body = R.cloud_io.format_ply(xyz).split(b"end_header\n    ", 1)[1]
    return body[:-5]                                         # without the "\n    " trailer


def first_difference(got, want):
    gl, wl = got.split(b"\n"), want.split(b"\n")
    for k, (g, w) in enumerate(zip(gl, wl)):
        if g != w:
            return k, g, w
    return min(len(gl), len(wl)), b"<end>", b"<end>"


def boundary_doubles():
    vals = []
    for e in range(-1074, 1024):
        p = float(2.0 ** e) if e >= -1022 else float(np.ldexp(1.0, e))
        vals += [p, np.nextafter(p, np.inf), np.nextafter(p, 0.0)]
        if -1022 <= e < 1023:
            vals += [float(np.ldexp(1.5, e)), float(np.ldexp(1.0 + 2.0 ** -52 * 0x5555555555555, e)), float(np.ldexp(2.0 - 2.0 ** -52, e))]
    for k in range(-323, 309):
        t = float("1e%d" % k)
        vals += [t, np.nextafter(t, np.inf), np.nextafter(t, 0.0), float("9.5e%d" % k), float("1.2345678901234567e%d" % k), float("5e%d" % k)]
    vals += [5e-324, 1.7976931348623157e308, 2.2250738585072014e-308, 2.225073858507201e-308, 9007199254740991.0, 9007199254740992.0,
             9007199254740994.0, 1e16, 9999999999999998.0, 1e-4, 0.0001, 0.00009999999999999999, 123456789012345680.0, 1e22, 1e23,
             100.0, 1000000.0, 120000.0, 0.1, 0.2, 0.30000000000000004, 1 / 3, 2 / 3, 1e15, 123.0, 0.5, 4.35, 4.350000000000001,
             0.0, -0.0, np.inf, -np.inf, np.nan]
    vals = np.array(vals, np.float64)
    return np.concatenate([vals, -vals[::7]])


def test_repr_lines_equal_the_host_formatter_on_every_exponent_and_boundary(R, ctx):
    vals = boundary_doubles()
    vals = vals[: len(vals) // 3 * 3].reshape(-1, 3)
    got, offs = device_text(R, ctx, R.device_text.TEXT_XYZ_TXT, vals)
    want = R.cloud_io.format_xyz_txt(vals)
    assert got == want, first_difference(got, want)
    assert offs.tolist() == [0, len(want)]
    # ... and against Python itself on a slice (the host formatter is pinned to repr() on all of them on the CPU side)
    lines = got.decode().split("\n")
    for row, ln in zip(vals[:2000].tolist(), lines):
        assert ln == ",".join(repr(v) for v in row)


def test_repr_lines_on_random_bit_patterns_and_camera_values(R, ctx):
    rng = np.random.default_rng(20261004)
    bits = rng.integers(0, 1 << 64, 3_000_000, dtype=np.uint64)
    vals = bits.view(np.float64).reshape(-1, 3)               # NaNs and infinities included
    got, _ = device_text(R, ctx, R.device_text.TEXT_XYZ_TXT, vals)
    want = R.cloud_io.format_xyz_txt(vals)
    assert got == want, first_difference(got, want)
    fx, fy, cx, cy = R.REF_INTRINSICS
    z = rng.integers(0, 256, 600_000)
    cam = np.stack([((rng.integers(0, 1280, 600_000) - cx) / fx) * z, ((rng.integers(0, 384, 600_000) - cy) / fy) * z, z.astype(np.float64)], 1)
    for dtype in (np.float64, np.float32):
        for z_raw, code in ((None, 0), (z.astype(np.uint8), R.DEPTH_U8), ((z * 257).astype(np.uint16), R.DEPTH_U16)):
            a = cam.astype(dtype)
            got, _ = device_text(R, ctx, R.device_text.TEXT_XYZ_TXT, a, z_raw, code)
            want = R.cloud_io.format_xyz_txt(a, z_raw=z_raw)
            assert got == want, (dtype, code, first_difference(got, want))


def test_percent_4f_rows_on_ties_near_ties_and_random_values(R, ctx):
    rng = np.random.default_rng(7)
    ties = (2 * rng.integers(0, 1 << 20, 200_000) + 1) / 32.0 * rng.choice([1.0, 0.5, 0.25, 0.125, 2.0, 1024.0], 200_000)
    ties = np.concatenate([ties, np.arange(1, 20001, 2) / 32.0])
    near = np.concatenate([np.nextafter(ties, np.inf), np.nextafter(ties, -np.inf)])
    almost = (rng.integers(0, 10 ** 9, 400_000) + 0.5) / 1e4
    big = rng.uniform(2.0 ** 30, 2.0 ** 40, 100_000)
    big = np.concatenate([big, np.floor(big) + 0.5, [2.0 ** 40 - 2.0 ** -13, np.nextafter(2.0 ** 40, 0.0)]])
    tiny = np.concatenate([rng.uniform(0, 1e-3, 100_000), [5e-5, 4.9999999999999996e-5, 5.000000000000001e-5, 1.5e-4, 2.5e-4, 5e-324, 0.0,
                                                             -0.0, np.inf, -np.inf, np.nan, 2.2250738585072014e-308]])
    rand = rng.normal(0, 1, 2_000_000) * 10.0 ** rng.integers(-6, 9, 2_000_000)
    vals = np.concatenate([ties, near, almost, big, tiny, rand])
    vals = np.concatenate([vals, -vals[::5]])
    vals = vals[: len(vals) // 3 * 3].reshape(-1, 3)
    got, _ = device_text(R, ctx, R.device_text.TEXT_PLY_ROWS, vals)
    want = host_ply_rows(R, vals)
    assert got == want, first_difference(got, want)
    flat = vals.reshape(-1)[:30000].tolist()
    assert got.decode().split()[:30000] == ["%.4f" % v for v in flat]
    got32, _ = device_text(R, ctx, R.device_text.TEXT_PLY_ROWS, vals.astype(np.float32)[:400_000])
    assert got32 == host_ply_rows(R, vals.astype(np.float32)[:400_000])


def test_giants_are_refused_and_the_writer_takes_the_host_formatter_for_them(R, ctx, tmp_path):
    """|x| >= 2^40 in a "%.4f" row needs more digits than a device row holds: R3D_ERR_UNSUPPORTED from the C ABI, and
    TextWriter writes that file with the host formatter -- same bytes as cloud_io.write_ply."""
    L = importlib.import_module(PKG + "._lib")
    xyz = np.random.default_rng(1).normal(0, 10, (5000, 3))
    xyz[4321, 1] = 2.0 ** 40
    d = ctx.alloc(xyz.nbytes).upload(xyz)
    with pytest.raises(R.R3DError) as e:
        R.device_text.format_text(ctx, R.device_text.TEXT_PLY_ROWS, d.ptr, np.float64, 5000)
    assert e.value.code == L.ERR_UNSUPPORTED and "2^40" in str(e.value)
    total, _ = R.device_text.format_text(ctx, R.device_text.TEXT_XYZ_TXT, d.ptr, np.float64, 5000)       # repr() has no such limit
    assert total == len(R.cloud_io.format_xyz_txt(xyz))
    w = R.device_text.TextWriter(ctx)
    w.add_ply(tmp_path / "g.ply", d.ptr, np.float64, 5000)
    w.write()
    R.cloud_io.write_ply(str(tmp_path / "h.ply"), xyz)
    assert (tmp_path / "g.ply").read_bytes() == (tmp_path / "h.ply").read_bytes()
    d.free()


@pytest.mark.parametrize("n,seg", [(1, 0), (255, 0), (256, 0), (257, 0), (1000, 7), (5000, 256), (5000, 1000), (70_001, 10_000), (4 * 2240, 2240)])
def test_segments_start_where_their_first_row_does(R, ctx, n, seg):
    """Tiles never straddle a segment: offsets[s] is the byte where point s * segment_points' row starts, for sizes around
    the 256-point tile, segments smaller than a tile, a ragged last segment and 40 x 56 frames."""
    rng = np.random.default_rng(n * 31 + seg)
    xyz = rng.normal(0, 30, (n, 3)) * 10.0 ** rng.integers(-5, 6, (n, 1))
    z = rng.integers(0, 256, n).astype(np.uint8)
    got, offs = device_text(R, ctx, R.device_text.TEXT_XYZ_TXT, xyz, z, R.DEPTH_U8, seg)
    want = R.cloud_io.format_xyz_txt(xyz, z_raw=z)
    assert got == want
    per = seg or n
    n_seg = -(-n // per)
    assert len(offs) == n_seg + 1 and offs[0] == 0 and offs[-1] == len(want)
    for s in range(n_seg):
        part = R.cloud_io.format_xyz_txt(xyz[s * per:(s + 1) * per], z_raw=z[s * per:(s + 1) * per])
        assert got[offs[s]:offs[s + 1]] == part, s
    rows, _ = device_text(R, ctx, R.device_text.TEXT_PLY_ROWS, xyz, segment_points=seg)
    assert rows == host_ply_rows(R, xyz)


def test_coloured_rows(R, ctx):
    rng = np.random.default_rng(3)
    n = 10_000
    xyz = rng.normal(0, 100, (n, 3))
    rgb = rng.integers(0, 256, (n, 3), dtype=np.uint8)
    rgba = np.concatenate([rgb, np.full((n, 1), 77, np.uint8)], 1)          # the fourth byte is not printed (the row ends in " 0")
    want = b"".join(b"%s %s %s %d %d %d 0\n" % (tuple(("%.4f" % v).encode() for v in p) + tuple(int(c) for c in col)) for p, col in zip(xyz.tolist(), rgb))
    for col, stride in ((rgb, 3), (rgba, 4)):
        got, _ = device_text(R, ctx, R.device_text.TEXT_PLY_ROWS_RGB, xyz, np.ascontiguousarray(col), stride)
        assert got == want, first_difference(got, want)


def test_writer_reproduces_the_reference_generated_files(R, ctx, golden_dir, tmp_path):
    """The fixtures the reference itself wrote (tests/golden/make_golden.py): the fused PLY of the 3-frame scene and KAT-1's
    camera txt, from device-resident clouds through TextWriter -- byte for byte."""
    from oracle import fusion_ref as O
    import json
    scene = os.path.join(golden_dir, "scene3")
    names, quats, ts = R.read_pose_file(os.path.join(scene, "camera_pose", "image_colmap_simi_2.txt"))
    depths = R.cloud_io.read_depth_batch([os.path.join(scene, "depth", n) for n in names])
    F, H, W = depths.shape
    n = F * H * W
    cam = ctx.camera(H, W, *R.REF_INTRINSICS)
    tab = R.pose_table(quats, ts)
    d_depth, d_pose = ctx.alloc(depths.nbytes).upload(depths), ctx.alloc(tab.nbytes).upload(tab)
    d_world, d_cam = ctx.alloc(n * 24), ctx.alloc(n * 24)
    R.fuse_frames_device(ctx, cam, d_depth.ptr, np.uint8, F, d_pose.ptr, d_world.ptr, np.float64)
    R.unproject_device(ctx, cam, d_depth.ptr, np.uint8, F, d_cam.ptr, np.float64)
    w = R.device_text.TextWriter(ctx)
    w.add_ply(tmp_path / "fused.ply", d_world.ptr, np.float64, n)
    w.add_xyz_txt([tmp_path / (nm[:-4] + ".txt") for nm in names], d_cam.ptr, np.float64, n, d_z_raw=d_depth.ptr, z_dtype=np.uint8)
    w.add_xyz_txt([tmp_path / "world_last.txt"], d_world.ptr + (F - 1) * H * W * 24, np.float64, H * W)
    total = w.write()
    assert (tmp_path / "fused.ply").read_bytes() == open(os.path.join(scene, "ply", "small_035_p8.ply"), "rb").read()
    for nm in names:
        ref = os.path.join(scene, "point", nm[:-4] + ".txt")
        assert (tmp_path / (nm[:-4] + ".txt")).read_bytes() == open(ref, "rb").read(), nm
    # the world txt differs from the reference's in the last bit of some coordinates (its BLAS's dgemv tail):
    # compared with the host formatter on the library's own cloud, and with the reference's numbers to 1e-12
    world = d_world.download(np.float64, n * 3).reshape(-1, 3)
    assert (tmp_path / "world_last.txt").read_bytes() == R.cloud_io.format_xyz_txt(world[(F - 1) * H * W:])
    assert total == sum(os.path.getsize(tmp_path / f) for f in os.listdir(tmp_path)) - len(R.device_text.ply_header(n)) - 5
    for b in (d_depth, d_pose, d_world, d_cam):
        b.free()


def test_writer_many_files_and_errors(R, ctx, tmp_path):
    """100 files side by side (more files than writer threads), file sizes that are not multiples of the 1 MiB piece, an
    empty cloud, a path that cannot be opened."""
    rng = np.random.default_rng(9)
    per, F = 30_011, 100
    xyz = rng.normal(0, 50, (per * F, 3))
    z = rng.integers(0, 65536, per * F).astype(np.uint16)
    d_xyz, d_z = ctx.alloc(xyz.nbytes).upload(xyz), ctx.alloc(z.nbytes).upload(z)
    w = R.device_text.TextWriter(ctx)
    paths = [tmp_path / ("f%03d.txt" % k) for k in range(F)]
    w.add_xyz_txt(paths, d_xyz.ptr, np.float64, per * F, d_z_raw=d_z.ptr, z_dtype=np.uint16)
    w.add_ply(tmp_path / "all.ply", d_xyz.ptr, np.float64, per * F)
    w.add_ply(tmp_path / "empty.ply", None, np.float64, 0)
    xyz32 = xyz.astype(np.float32)
    d_32 = ctx.alloc(xyz32.nbytes).upload(xyz32)
    w.add_ply_binary(tmp_path / "bin.ply", d_32.ptr, per * F)           # f1's optional flag: the cloud's own bytes behind a header
    w.add_ply_binary(tmp_path / "bin0.ply", None, 0)
    w.write()
    R.cloud_io.write_ply_binary(str(tmp_path / "bin_host.ply"), xyz32)
    assert (tmp_path / "bin.ply").read_bytes() == (tmp_path / "bin_host.ply").read_bytes()
    assert (tmp_path / "bin0.ply").read_bytes() == R.device_text.ply_header_binary(0)
    d_32.free()
    for k in (0, 1, 57, 99):
        assert paths[k].read_bytes() == R.cloud_io.format_xyz_txt(xyz[k * per:(k + 1) * per], z_raw=z[k * per:(k + 1) * per])
    assert (tmp_path / "all.ply").read_bytes() == R.cloud_io.format_ply(xyz)
    assert (tmp_path / "empty.ply").read_bytes() == R.cloud_io.format_ply(np.zeros((0, 3)))
    w.add_ply(tmp_path / "no_such_dir" / "x.ply", d_xyz.ptr, np.float64, 1000)
    with pytest.raises(R.R3DError) as e:
        w.write()
    assert "cannot open" in str(e.value)
    with pytest.raises(ValueError):
        w.add_xyz_txt([tmp_path / "a.txt", tmp_path / "b.txt"], d_xyz.ptr, np.float64, 3)
    d_xyz.free()
    d_z.free()
