"""GPU parity tests proper: the HIP path (through the C ABI) against the oracle on the same
seeded inputs, against the golden fixtures, and through size-independent properties at
BASELINE.json's full size.  Tolerance (north_star <= 1e-4 relative):
  f32 output: point-wise ||d||/||ref|| <= 1e-6 and per-component |d| <= 1e-4*max(|ref|,1e-3||ref||)
              (the kernel rounds the fp64 result once, so the observed error is ~6e-8);
  f64 output: |d| <= 1e-12*(1+||ref||)  (summation order only).
"""
import importlib
import json
import os

import numpy as np
import pytest

from helpers import r3d as _r3d
from oracle import fusion_ref as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    return _r3d()


@pytest.fixture(scope="module")
def ctx(R):
    c = R.Context(0)
    yield c
    c.close()


def check(got, want, out_dtype):
    assert got.shape == want.shape and got.dtype == out_dtype
    if got.size == 0:
        return
    if out_dtype == np.float32:
        e_norm, e_comp = O.parity_errors(got, want)
        assert e_norm <= 1e-6, e_norm
        assert e_comp <= 1e-4, e_comp
    else:
        scale = 1.0 + np.linalg.norm(want, axis=1, keepdims=True)
        assert (np.abs(got - want) / scale).max() <= 1e-12


def make_depth(rng, shape, dtype):
    if dtype == np.uint8:
        return rng.integers(0, 256, size=shape, dtype=np.uint8)
    if dtype == np.uint16:
        return rng.integers(0, 65536, size=shape, dtype=np.uint16)
    return (rng.random(size=shape) * 99.5 + 0.5).astype(np.float32)


SHAPES = [(1, 4, 6), (3, 24, 32), (2, 37, 52), (1, 192, 640), (5, 33, 128), (2, 3, 2), (1, 1, 2), (1, 480, 640)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("ddtype", [np.uint8, np.uint16, np.float32])
@pytest.mark.parametrize("odtype", [np.float32, np.float64])
def test_fuse_matches_oracle(R, ctx, shape, ddtype, odtype):
    rng = np.random.default_rng(hash((shape, str(ddtype))) % 2**32)
    d = make_depth(rng, shape, ddtype)
    q = rng.normal(size=(shape[0], 4))
    t = rng.normal(size=(shape[0], 3)) * 10
    got = R.fuse_frames(d, q, t, out_dtype=odtype, ctx=ctx)
    check(got, O.fuse_frames(d, q, t), odtype)
    got_u = R.unproject(d, out_dtype=odtype, ctx=ctx)
    want_u = np.concatenate([O.unproject(f) for f in d])
    check(got_u, want_u, odtype)
    if odtype == np.float64:  # unprojection is a single fp64 product per coordinate: bit exact
        np.testing.assert_array_equal(got_u, want_u)


@pytest.mark.parametrize("shape", [(3, 24, 32), (2, 37, 52), (2, 100, 1280)])
@pytest.mark.parametrize("odtype", [np.float32, np.float64])
def test_launch_geometry_never_changes_a_bit(R, ctx, shape, odtype):
    """The only tuning knob of the fused kernels is the workgroup count: grid-stride over tiles, same bits."""
    rng = np.random.default_rng(5)
    d = make_depth(rng, shape, np.uint8)
    q = rng.normal(size=(shape[0], 4))
    t = rng.normal(size=(shape[0], 3)) * 10
    outs = []
    for blocks in (0, 1, 7, 300, 100000):
        ctx.set_tuning("fuse_blocks", blocks)
        outs.append(R.fuse_frames(d, q, t, out_dtype=odtype, ctx=ctx))
    ctx.set_tuning("fuse_blocks", 0)
    assert ctx.get_tuning("fuse_blocks") == 0
    for o in outs[1:]:
        np.testing.assert_array_equal(o, outs[0])
    # the f32 cloud is the f64 cloud rounded once: the two kernels (lane-per-pixel / lane-pair) share their arithmetic
    if odtype == np.float64:
        np.testing.assert_array_equal(R.fuse_frames(d, q, t, out_dtype=np.float32, ctx=ctx), outs[0].astype(np.float32))


def test_intrinsics_and_scale(R, ctx):
    rng = np.random.default_rng(9)
    d = make_depth(rng, (2, 20, 28), np.uint16)
    q = rng.normal(size=(2, 4))
    t = rng.normal(size=(2, 3))
    K = (0.58 * 28, 1.92 * 20, 13.5, 9.5)  # monodepth2-KITTI style, non-integer principal point
    got = R.fuse_frames(d, q, t, intrinsics=K, out_dtype=np.float64, depth_scale=1.0 / 256.0, ctx=ctx)
    want = O.fuse_frames(d.astype(np.float64) / 256.0, q, t, *K)  # /256 is exact, so scale commutes
    check(got, want, np.float64)


def test_golden_scene3_and_kat(R, ctx, golden_dir):
    from PIL import Image
    scene = os.path.join(golden_dir, "scene3")
    names, quats, ts = R.read_pose_file(os.path.join(scene, "camera_pose", "image_colmap_simi_2.txt"))
    depths = np.stack([np.array(Image.open(os.path.join(scene, "depth", n)).convert("L")) for n in names])
    want = R.cloud_io.read_ply(os.path.join(scene, "ply", "small_035_p8.ply"))  # reference output, 4 decimals
    got64 = R.fuse_frames(depths, quats, ts, out_dtype=np.float64, ctx=ctx)
    got32 = R.fuse_frames(depths, quats, ts, out_dtype=np.float32, ctx=ctx)
    assert np.abs(got64 - want).max() <= 0.5001e-4
    assert np.abs(got32.astype(np.float64) - want).max() <= 0.5e-4 + 1e-4 * np.abs(want).max()
    # the reference's own PLY bytes, reproduced from the GPU's fp64 output through the native writer
    assert R.cloud_io.format_ply(got64) == open(os.path.join(scene, "ply", "small_035_p8.ply"), "rb").read()
    # world txt of the last frame, numerically
    world = R.cloud_io.read_xyz_txt(os.path.join(scene, "point_world", "small_worldpoint_5_23_5.txt"))
    check(got64[-768:], world, np.float64)
    # KAT-1 camera txt, byte exact through the GPU (fp64 product == reference's)
    j, i = np.mgrid[0:4, 0:6]
    kat = ((7 * j + 3 * i + 1) % 256).astype(np.uint8)
    cam = R.unproject(kat, out_dtype=np.float64, ctx=ctx)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        R.cloud_io.write_xyz_txt(os.path.join(td, "k.txt"), cam, z_raw=kat)
        assert open(os.path.join(td, "k.txt")).read() == open(os.path.join(golden_dir, "kat_unproject_4x6.txt")).read()


def test_golden_c1_world_points(R, ctx, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "c1_192x640.json")))
    depth = np.random.default_rng(g["seed"]).integers(1, 256, tuple(g["shape"]), dtype=np.uint8)
    got = R.fuse_frames(depth, [g["q_xyzw"]], [g["t"]], out_dtype=np.float64, ctx=ctx)
    for k, xyz in g["world_xyz"].items():
        np.testing.assert_allclose(got[int(k)], xyz, rtol=0, atol=1e-11)
    np.testing.assert_allclose(got.sum(0), g["world_sum_xyz"], rtol=1e-11)


@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 1023, 1024, 1025, 4099, 250001])
@pytest.mark.parametrize("idt,odt", [(np.float32, np.float32), (np.float64, np.float64), (np.float32, np.float64),
                                     (np.float64, np.float32)])
def test_apply_T_and_se3(R, ctx, n, idt, odt):
    rng = np.random.default_rng(n + 1)
    p = (rng.normal(size=(n, 3)) * 50).astype(idt)
    T = np.eye(4)
    T[:3, :3] = 1.7 * np.asarray(R.scipy_transfer(rng.normal(size=4)))
    T[:3, 3] = rng.normal(size=3) * 5
    rinv = np.asarray(R.scipy_transfer(rng.normal(size=4)))
    t = rng.normal(size=3) * 10
    outs = []
    for blocks in (0, 3):  # one tile per workgroup (default) and a grid-stride walk: bit-identical results
        ctx.set_tuning("apply_blocks", blocks)
        got = R.apply_T(p, T, out_dtype=odt, ctx=ctx)
        check(got, O.apply_T(p, T), odt)
        got2 = R.se3_apply(p, rinv, t, out_dtype=odt, ctx=ctx)
        check(got2, O.se3_apply(p, rinv, t), odt)
        outs.append((got, got2))
    ctx.set_tuning("apply_blocks", 0)
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    if odt == np.float64:   # lane-per-point f32 kernel == lane-pair f64 kernel rounded once
        np.testing.assert_array_equal(R.apply_T(p, T, out_dtype=np.float32, ctx=ctx), outs[0][0].astype(np.float32))


def test_fuse_equals_unproject_then_se3_bitwise(R, ctx):
    rng = np.random.default_rng(77)
    d = make_depth(rng, (3, 40, 64), np.uint8)
    q = rng.normal(size=(3, 4))
    t = rng.normal(size=(3, 3)) * 10
    fused = R.fuse_frames(d, q, t, out_dtype=np.float64, ctx=ctx)
    cam = R.unproject(d, out_dtype=np.float64, ctx=ctx).reshape(3, -1, 3)
    for k in range(3):
        w = R.se3_apply(cam[k], R.scipy_transfer(q[k]), t[k], ctx=ctx)
        np.testing.assert_array_equal(w, fused.reshape(3, -1, 3)[k])


def test_identity_pose_is_unproject_bitwise(R, ctx):
    rng = np.random.default_rng(78)
    d = make_depth(rng, (2, 16, 20), np.uint8)
    fused = R.fuse_frames(d, [[0, 0, 0, 1]] * 2, np.zeros((2, 3)), out_dtype=np.float32, ctx=ctx)
    np.testing.assert_array_equal(fused, R.unproject(d, out_dtype=np.float32, ctx=ctx))


def test_empty_and_errors(R, ctx):
    assert R.fuse_frames(np.zeros((0, 8, 8), np.uint8), np.zeros((0, 4)), np.zeros((0, 3)), ctx=ctx).shape == (0, 3)
    assert R.apply_T(np.zeros((0, 3), np.float32), np.eye(4), ctx=ctx).shape == (0, 3)
    with pytest.raises(TypeError):
        R.fuse_frames(np.zeros((1, 8, 8), np.int32), [[0, 0, 0, 1]], [[0, 0, 0]], ctx=ctx)
    with pytest.raises(ValueError):
        R.fuse_frames(np.zeros((2, 8, 8), np.uint8), [[0, 0, 0, 1]], [[0, 0, 0]], ctx=ctx)
    with pytest.raises(ValueError):
        R.fuse_frames(np.zeros((1, 8, 8), np.uint8), [[0, 0, 0, 0]], [[0, 0, 0]], ctx=ctx)
    with pytest.raises(R.R3DError):
        ctx.set_tuning("no_such_knob", 1)
    with pytest.raises(R.R3DError):
        R.Context(99)


def test_full_size_c2_properties(R, ctx):
    """BASELINE config 2: 100 frames of 1280x384 in one launch (49,152,000 points).
    Size-independent checks: (1) the batch equals frame-by-frame launches bit for bit, (2) a
    permutation of the frames permutes the output blocks, (3) linearity in Z along each ray:
    fuse(2*depth) - fuse(depth) == Rinv.(p_cam), (4) oracle agreement on 3 whole frames."""
    F, H, W = 100, 384, 1280
    rng = np.random.default_rng(1234)
    d = rng.integers(1, 128, size=(F, H, W), dtype=np.uint8)
    q = rng.normal(size=(F, 4))
    t = rng.normal(size=(F, 3)) * 10
    full = R.fuse_frames(d, q, t, out_dtype=np.float32, ctx=ctx).reshape(F, H * W, 3)
    for k in (0, 41, 99):
        one = R.fuse_frames(d[k], q[k:k + 1], t[k:k + 1], out_dtype=np.float32, ctx=ctx)
        np.testing.assert_array_equal(one, full[k])
        check(one, O.fuse_frames(d[k:k + 1], q[k:k + 1], t[k:k + 1]), np.float32)
    perm = rng.permutation(F)
    np.testing.assert_array_equal(R.fuse_frames(d[perm], q[perm], t[perm], ctx=ctx).reshape(F, H * W, 3), full[perm])
    G = 16  # fp64 clouds are 1.2 GB per 100 frames on the host: linearity on the first 16 frames
    d, q, t = d[:G], q[:G], t[:G]
    twice = R.fuse_frames((2 * d).astype(np.uint8), q, t, out_dtype=np.float64, ctx=ctx).reshape(G, H * W, 3)
    once = R.fuse_frames(d, q, t, out_dtype=np.float64, ctx=ctx).reshape(G, H * W, 3)
    zero_t = R.fuse_frames(d, q, np.zeros_like(t), out_dtype=np.float64, ctx=ctx).reshape(G, H * W, 3)
    lin = np.abs((twice - once) - zero_t)
    assert (lin / (1.0 + np.linalg.norm(zero_t, axis=2, keepdims=True))).max() <= 1e-12


def test_c4_sized_batch_64bit_indexing(R, ctx):
    """BASELINE config 4's per-node total on ONE GPU: 1000 frames of 1280x384 = 491,520,000 points, 5.9 GB of xyz in one
    launch (byte offsets far beyond 2^32).  Frames are tiled copies of 8 distinct rasters with distinct poses; the first,
    a middle and the last frame are checked against the oracle, and a per-frame checksum against per-frame launches."""
    F, H, W, D = 1000, 384, 1280, 8
    rng = np.random.default_rng(44)
    base = rng.integers(1, 256, size=(D, H, W), dtype=np.uint8)
    q = rng.normal(size=(F, 4))
    t = rng.normal(size=(F, 3)) * 10
    n = F * H * W
    d_depth = ctx.alloc(n)
    import ctypes as C
    L = __import__("importlib").import_module(R.__name__ + "._lib")
    for k in range(F):                      # upload frame by frame: the host never holds the 491 MB batch twice
        fr = np.ascontiguousarray(base[k % D])
        L.check(ctx.lib.r3d_memcpy_h2d(ctx.handle, d_depth.ptr + k * H * W, fr.ctypes.data, fr.nbytes))
    ctx.sync()
    tab = R.pose_table(q, t)
    d_pose = ctx.alloc(tab.nbytes).upload(tab)
    d_out = ctx.alloc(n * 12)
    cam = ctx.camera(H, W, *R.REF_INTRINSICS)
    R.fuse_frames_device(ctx, cam, d_depth.ptr, np.uint8, F, d_pose.ptr, d_out.ptr, np.float32)
    ctx.sync()
    per = H * W
    for k in (0, 499, 999):
        got = np.empty((per, 3), np.float32)
        L.check(ctx.lib.r3d_memcpy_d2h(ctx.handle, got.ctypes.data, d_out.ptr + k * per * 12, got.nbytes))
        ctx.sync()
        check(got, O.fuse_frames(base[k % D][None], q[k:k + 1], t[k:k + 1]), np.float32)
        one = R.fuse_frames(base[k % D], q[k:k + 1], t[k:k + 1], out_dtype=np.float32, ctx=ctx)
        np.testing.assert_array_equal(one, got)
    for b in (d_depth, d_pose, d_out):
        b.free()


def test_pinned_and_preallocated_outputs(R, ctx):
    rng = np.random.default_rng(12)
    d = make_depth(rng, (37, 120, 160), np.uint8)              # 37 frames: several pipeline chunks, ragged last chunk
    q = rng.normal(size=(37, 4))
    t = rng.normal(size=(37, 3)) * 10
    want = R.fuse_frames(d, q, t, out_dtype=np.float32, ctx=ctx)
    check(want, O.fuse_frames(d, q, t), np.float32)
    pin = ctx.pinned_empty(want.shape, np.float32)
    got = R.fuse_frames(d, q, t, out_dtype=np.float32, ctx=ctx, out=pin)
    assert got is pin
    np.testing.assert_array_equal(pin, want)
    pre = np.zeros(want.shape, np.float32)
    R.fuse_frames(d, q, t, out_dtype=np.float32, ctx=ctx, out=pre)
    np.testing.assert_array_equal(pre, want)
    pin_in = ctx.pinned_empty(d.shape, np.uint8)               # pinned INPUT as well: no staging in either direction
    pin_in[...] = d
    pin[...] = 0
    R.fuse_frames(pin_in, q, t, out_dtype=np.float32, ctx=ctx, out=pin)
    np.testing.assert_array_equal(pin, want)
    with pytest.raises(ValueError):
        R.fuse_frames(d, q, t, ctx=ctx, out=np.zeros((5, 3), np.float32))
    del pin, pin_in, got


@pytest.mark.gpu
@pytest.mark.parametrize("nbytes", [65536 * 192 + 12, 65536 * 128 + 1, 65536 * 130 + 15, (32 << 20) + 65536 * 64 + 7, 4 * 65536 * 16 + 3])
def test_staging_copies_reach_the_last_byte(R, ctx, nbytes):
    """The pageable <-> pinned copies of the host pipeline are split over host threads in page-rounded shares.  Sizes whose
    per-thread share is a whole number of pages while the size is not a multiple of the thread count used to lose their last
    few bytes (floor before rounding: found in round 4 by a race-detector test of the new crew).  Download into pageable
    memory and the *_host fused path on a ragged single-row raster, byte for byte."""
    rng = np.random.default_rng(nbytes)
    src = rng.integers(0, 256, nbytes, dtype=np.uint8)
    buf = ctx.alloc(nbytes).upload(src)
    try:
        for _ in range(2):
            np.testing.assert_array_equal(buf.download(np.uint8, nbytes), src)
    finally:
        buf.free()
    if nbytes < 16 << 20:
        depth = src.reshape(1, 1, nbytes)                                   # one frame, one row: the chunk IS the raster
        q, t = np.array([[0.1, 0.2, 0.3, 0.9]]), np.array([[1.0, 2.0, 3.0]])
        pin_in = ctx.pinned_empty(depth.shape, np.uint8)                    # pinned both ways: no staging copies at all
        pin_in[...] = depth
        pin_out = ctx.pinned_empty((nbytes, 3), np.float32)
        R.fuse_frames(pin_in, q, t, out_dtype=np.float32, ctx=ctx, out=pin_out)
        got = R.fuse_frames(depth, q, t, out_dtype=np.float32, ctx=ctx)      # pageable both ways
        np.testing.assert_array_equal(got, pin_out)
        del pin_in, pin_out


def test_nonfinite_and_negative_f32_depth_propagate_like_numpy(R, ctx):
    """f32 rasters may carry NaN / inf / negative / zero: no masking in the reference, none here; IEEE semantics match."""
    rng = np.random.default_rng(13)
    d = (rng.random((2, 16, 24)) * 50).astype(np.float32)
    d[0, 0, :6] = [np.nan, np.inf, -np.inf, 0.0, -3.5, 1e30]
    d[1, 5, 7] = np.nan
    q = rng.normal(size=(2, 4))
    t = rng.normal(size=(2, 3))
    got = R.fuse_frames(d, q, t, out_dtype=np.float64, ctx=ctx)
    with np.errstate(invalid="ignore", over="ignore"):
        want = O.fuse_frames(d, q, t)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(np.isinf(got), np.isinf(want))
    fin = np.isfinite(want)
    assert (np.abs(got[fin] - want[fin]) <= 1e-12 * (1 + np.abs(want[fin]))).all()
    got32 = R.fuse_frames(d, q, t, out_dtype=np.float32, ctx=ctx)
    assert np.array_equal(np.isnan(got32), np.isnan(want))


@pytest.mark.parametrize("hw", [(3, 1), (2, 3), (3, 5), (2, 1023), (3, 1025), (2, 4099), (5, 257), (1, 8192)])
@pytest.mark.parametrize("odtype", [np.float32, np.float64])
def test_odd_widths(R, ctx, hw, odtype):
    """Rasters whose size is not a multiple of anything convenient (and a 1-pixel-wide one): ragged last tiles."""
    h, w = hw
    rng = np.random.default_rng(h * 10007 + w)
    d = make_depth(rng, (3, h, w), np.uint16)
    q = rng.normal(size=(3, 4))
    t = rng.normal(size=(3, 3)) * 10
    check(R.fuse_frames(d, q, t, out_dtype=odtype, ctx=ctx), O.fuse_frames(d, q, t), odtype)


# ---- colour carried through the fused launch (f4; BASELINE config 5 "RGBD") ---------------------------------------
def _rgba_words(rgb):
    r = rgb.reshape(-1, 3).astype(np.uint32)
    return r[:, 0] | (r[:, 1] << 8) | (r[:, 2] << 16)


@pytest.mark.parametrize("shape", [(1, 4, 6), (3, 24, 32), (2, 37, 52), (2, 64, 128), (1, 192, 640), (3, 33, 1000)])
@pytest.mark.parametrize("ddtype", [np.uint8, np.uint16, np.float32])
@pytest.mark.parametrize("odtype", [np.float32, np.float64])
def test_fuse_rgb_matches_plain_fuse_and_carries_colour(R, ctx, shape, ddtype, odtype):
    """xyz of the colour-carrying launch == the plain launch bit for bit; point k gets pixel k's colour, alpha 0.
    Shapes cover whole 1024-pixel tiles (16-byte staged colour loads), ragged tiles and unaligned frames (byte loads)."""
    rng = np.random.default_rng(sum(shape) + 3)
    d = make_depth(rng, shape, ddtype)
    rgb = rng.integers(0, 256, size=shape + (3,), dtype=np.uint8)
    q = rng.normal(size=(shape[0], 4))
    t = rng.normal(size=(shape[0], 3)) * 10
    xyz, rgba = R.fuse_frames_rgb(d, rgb, q, t, out_dtype=odtype, ctx=ctx)
    np.testing.assert_array_equal(xyz, R.fuse_frames(d, q, t, out_dtype=odtype, ctx=ctx))
    np.testing.assert_array_equal(rgba, _rgba_words(rgb))
    cam_xyz, rgba2 = R.fuse_frames_rgb(d, rgb, out_dtype=odtype, ctx=ctx)          # no pose: pixel_to_camera.py
    np.testing.assert_array_equal(cam_xyz, R.unproject(d, out_dtype=odtype, ctx=ctx))
    np.testing.assert_array_equal(rgba2, rgba)


def test_fuse_rgb_reproduces_reference_coloured_ply(R, ctx, golden_dir, tmp_path):
    """genply_noRGB's file (pixel_to_camera.py:55-91), fixture made with the reference itself (tests/golden/make_golden.py):
    the first 6 points of the seeded 480x640 raster with the colours of p2c_rgb_2x3.png -> xyz + colour through the GPU
    -> byte-identical PLY."""
    import json
    import os
    from PIL import Image
    g = json.load(open(os.path.join(golden_dir, "p2c_480x640.json")))
    depth = np.random.default_rng(g["seed"]).integers(1, 256, tuple(g["shape"]), dtype=np.uint8)
    six = np.array(Image.open(os.path.join(golden_dir, "p2c_rgb_2x3.png")).convert("RGB")).reshape(-1, 3)
    img = np.zeros(depth.shape + (3,), np.uint8)
    img.reshape(-1, 3)[:6] = six
    want = open(os.path.join(golden_dir, "p2c_first6_rgb.ply"), "rb").read()
    for odt in (np.float64, np.float32):
        xyz, rgba = R.fuse_frames_rgb(depth, img, out_dtype=odt, ctx=ctx)
        out = tmp_path / ("c%d.ply" % np.dtype(odt).itemsize)
        R.cloud_io.write_ply_rgb(str(out), xyz[:6], rgba[:6])
        assert out.read_bytes() == want


def test_config5_geometry_1080p_f32_rgbd(R, ctx):
    """BASELINE config 5 geometry (1920x1080 f32 depth + RGB), 50 frames = 103.7 M points, HBM-resident: properties
    that need no oracle pass over 100 M points -- per-frame launches == one batched launch, colour == input, and a
    strided sample against the oracle."""
    F, H, W = 50, 1080, 1920
    rng = np.random.default_rng(5)
    depth = (rng.random((F, H, W), dtype=np.float32) * 99.5 + 0.5)
    rgb = rng.integers(0, 256, size=(F, H, W, 3), dtype=np.uint8)
    q = rng.normal(size=(F, 4))
    t = rng.normal(size=(F, 3)) * 10
    intr = (960.0, 960.0, 959.5, 539.5)
    cam = ctx.camera(H, W, *intr)
    n = F * H * W
    table = R.pose_table(q, t)
    d_depth, d_rgb, d_pose = ctx.alloc(depth.nbytes).upload(depth), ctx.alloc(rgb.nbytes).upload(rgb), ctx.alloc(table.nbytes).upload(table)
    d_xyz, d_rgba, d_xyz2, d_rgba2 = ctx.alloc(n * 12), ctx.alloc(n * 4), ctx.alloc(n * 12), ctx.alloc(n * 4)
    R.fuse_frames_rgb_device(ctx, cam, d_depth.ptr, np.float32, F, d_pose.ptr, d_rgb.ptr, d_xyz.ptr, np.float32, d_rgba.ptr)
    per = H * W
    for f in range(F):      # frame by frame into the second pair of buffers
        R.fuse_frames_rgb_device(ctx, cam, d_depth.ptr + f * per * 4, np.float32, 1, d_pose.ptr + f * 96, d_rgb.ptr + f * per * 3,
                                 d_xyz2.ptr + f * per * 12, np.float32, d_rgba2.ptr + f * per * 4)
    xyz = d_xyz.download(np.float32, n * 3).reshape(-1, 3)
    np.testing.assert_array_equal(xyz, d_xyz2.download(np.float32, n * 3).reshape(-1, 3))
    rgba = d_rgba.download(np.uint32, n)
    np.testing.assert_array_equal(rgba, d_rgba2.download(np.uint32, n))
    np.testing.assert_array_equal(rgba, _rgba_words(rgb))
    for f in (0, 17, 49):   # oracle on three whole frames
        want = O.fuse_frames(depth[f:f + 1], q[f:f + 1], t[f:f + 1], *intr)
        check(xyz[f * per:(f + 1) * per], want, np.float32)
    for b in (d_depth, d_rgb, d_pose, d_xyz, d_rgba, d_xyz2, d_rgba2):
        b.free()


def test_config5_per_gpu_share_250_frames_single_launch(R, ctx):
    """BASELINE config 5 per GPU at 8 GPUs: 250 frames of 1920x1080 f32 depth + RGB = 518.4 M points in ONE launch
    (6.2 GB of xyz, 2.1 GB of rgba; 506,250 tiles).  The rasters are 10 distinct host frames replicated on the device; the
    250 poses are all different.  Checks: one launch == 25 launches of 10 frames, bit for bit (xyz and colour); three frames
    against the oracle; 64-bit addressing past 4 G bytes."""
    F, H, W, U = 250, 1080, 1920, 10
    rng = np.random.default_rng(55)
    depth = (rng.random((U, H, W), dtype=np.float32) * 99.5 + 0.5)
    rgb = rng.integers(0, 256, size=(U, H, W, 3), dtype=np.uint8)
    q, t = rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10
    intr = (960.0, 960.0, 959.5, 539.5)
    cam = ctx.camera(H, W, *intr)
    per = H * W
    n = F * per
    table = R.pose_table(q, t)
    lib = R.load_library()
    d_depth, d_rgb, d_pose = ctx.alloc(n * 4), ctx.alloc(n * 3), ctx.alloc(table.nbytes).upload(table)
    d_depth.upload(depth)
    d_rgb.upload(rgb)
    for k in range(1, F // U):      # replicate the 10 frames 25 times in HBM
        lib.r3d_memcpy_d2d(ctx.handle, d_depth.ptr + k * U * per * 4, d_depth.ptr, U * per * 4)
        lib.r3d_memcpy_d2d(ctx.handle, d_rgb.ptr + k * U * per * 3, d_rgb.ptr, U * per * 3)
    d_xyz, d_rgba = ctx.alloc(n * 12), ctx.alloc(n * 4)
    R.fuse_frames_rgb_device(ctx, cam, d_depth.ptr, np.float32, F, d_pose.ptr, d_rgb.ptr, d_xyz.ptr, np.float32, d_rgba.ptr)
    d_xyz2, d_rgba2 = ctx.alloc(U * per * 12), ctx.alloc(U * per * 4)
    L_ = importlib.import_module(R.__name__ + "._lib")
    want_rgba = (rgb.reshape(-1, 3).astype(np.uint32) * np.array([1, 256, 65536], dtype=np.uint32)).sum(1, dtype=np.uint32)
    for k in range(F // U):
        R.fuse_frames_rgb_device(ctx, cam, d_depth.ptr + k * U * per * 4, np.float32, U, d_pose.ptr + k * U * 96,
                                 d_rgb.ptr + k * U * per * 3, d_xyz2.ptr, np.float32, d_rgba2.ptr)
        part = d_xyz2.download(np.float32, U * per * 3)
        whole = np.empty(U * per * 3, np.float32)
        L_.check(lib.r3d_memcpy_d2h(ctx.handle, whole.ctypes.data, d_xyz.ptr + k * U * per * 12, whole.nbytes))
        ctx.sync()
        assert np.array_equal(part, whole), k
        if k in (0, 24):
            col = np.empty(U * per, np.uint32)
            L_.check(lib.r3d_memcpy_d2h(ctx.handle, col.ctypes.data, d_rgba.ptr + k * U * per * 4, col.nbytes))
            ctx.sync()
            assert np.array_equal(col, want_rgba)
    for f in (0, 137, 249):
        one = np.empty(per * 3, np.float32)
        L_.check(lib.r3d_memcpy_d2h(ctx.handle, one.ctypes.data, d_xyz.ptr + f * per * 12, one.nbytes))
        ctx.sync()
        want = O.fuse_frames(depth[f % U:f % U + 1], q[f:f + 1], t[f:f + 1], *intr)
        check(one.reshape(-1, 3), want, np.float32)
    for b in (d_depth, d_rgb, d_pose, d_xyz, d_rgba, d_xyz2, d_rgba2):
        b.free()


@pytest.mark.parametrize("odtype", [np.float32, np.float64])
def test_fuse_rgb_host_pipeline_many_chunks(R, ctx, odtype):
    """The host entry point streams depth + colour in and xyz + rgba out through the pinned pipeline: a batch large enough
    for many chunks (40 frames of 1280x384), pageable and pinned outputs, against the device-pointer path."""
    F, H, W = 40, 384, 1280
    rng = np.random.default_rng(8)
    d = rng.integers(0, 256, size=(F, H, W), dtype=np.uint8)
    rgb = rng.integers(0, 256, size=(F, H, W, 3), dtype=np.uint8)
    q, t = rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10
    xyz, rgba = R.fuse_frames_rgb(d, rgb, q, t, out_dtype=odtype, ctx=ctx)
    np.testing.assert_array_equal(xyz, R.fuse_frames(d, q, t, out_dtype=odtype, ctx=ctx))
    np.testing.assert_array_equal(rgba, _rgba_words(rgb))
    for f in (0, 39):
        check(xyz[f * H * W:(f + 1) * H * W], O.fuse_frames(d[f:f + 1], q[f:f + 1], t[f:f + 1]), odtype)


def test_seeded_random_shape_sweep_every_kernel(R, ctx):
    """120 seeded random (F, H, W) -- frame sizes around the 1024-pixel tile (1023, 1024, 1025, multiples, primes), 1..7
    frames -- through every shipped kernel (3 depth types x f32/f64 xyz x pose / no pose x plain / colour) against the oracle.
    Random shapes find what hand-picked ones do not: tile tails, frames that are not a multiple of the tile, 1-wide rasters.
    (tools/stress_random.py runs the same sweep under other seeds for as long as it is given.)"""
    random_shape_sweep(R, ctx, 20260, 120)


def random_shape_sweep(R, ctx, seed, cases):
    rng = np.random.default_rng(seed)
    specials = [1, 2, 3, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1021, 1023, 1024, 1025, 2047, 2048, 2049, 3071,
                3072, 3073, 4095, 4096, 4097, 5003]
    for case in range(cases):
        if case % 3 == 0:                                  # a frame of exactly / nearly k tiles, factored at random
            px = int(rng.choice(specials))
            divs = [d for d in range(1, px + 1) if px % d == 0]
            h = int(rng.choice(divs))
            w = px // h
        else:
            h, w = int(rng.integers(1, 70)), int(rng.integers(1, 300))
        f = int(rng.integers(1, 8))
        ddtype = [np.uint8, np.uint16, np.float32][case % 3 if case % 2 else int(rng.integers(0, 3))]
        odtype = [np.float32, np.float64][int(rng.integers(0, 2))]
        d = make_depth(rng, (f, h, w), ddtype)
        q = rng.normal(size=(f, 4))
        t = rng.normal(size=(f, 3)) * 10
        K = (float(rng.uniform(50, 900)), float(rng.uniform(50, 900)), float(rng.uniform(0, w)), float(rng.uniform(0, h)))
        want = O.fuse_frames(d, q, t, *K)
        got = R.fuse_frames(d, q, t, intrinsics=K, out_dtype=odtype, ctx=ctx)
        check(got, want, odtype)
        got_u = R.unproject(d, intrinsics=K, out_dtype=odtype, ctx=ctx)
        want_u = np.concatenate([O.unproject(fr, *K) for fr in d])
        check(got_u, want_u, odtype)
        if odtype == np.float64:
            np.testing.assert_array_equal(got_u, want_u, err_msg="case %d %s" % (case, (f, h, w, ddtype)))
        rgb = rng.integers(0, 256, size=(f, h, w, 3), dtype=np.uint8)
        xyz, rgba = R.fuse_frames_rgb(d, rgb, q, t, intrinsics=K, out_dtype=odtype, ctx=ctx)
        np.testing.assert_array_equal(xyz, got, err_msg="case %d %s" % (case, (f, h, w, ddtype)))
        np.testing.assert_array_equal(rgba, _rgba_words(rgb))
        cam_xyz, rgba2 = R.fuse_frames_rgb(d, rgb, intrinsics=K, out_dtype=odtype, ctx=ctx)
        np.testing.assert_array_equal(cam_xyz, got_u)
        np.testing.assert_array_equal(rgba2, rgba)


@pytest.mark.parametrize("ddtype", [np.uint8, np.uint16, np.float32])
@pytest.mark.parametrize("odtype", [np.float32, np.float64])
def test_input_staging_in_chunks_never_changes_a_bit(R, ctx, ddtype, odtype):
    """Big batches have their inputs swept into the Infinity Cache chunk by chunk before each chunk is fused (fuse_prefetch /
    fuse_chunk_mb; auto above fuse_stage_auto_mb = 8 MB of inputs).  Forced on with 1 MB chunks here: several chunks, a ragged last one, frame
    offsets into every array (raster, pose table, xyz, colour, rgba) -- the results must equal the single launch bit for bit."""
    rng = np.random.default_rng(5)
    F, H, W = 37, 96, 130                                       # 12,480 pixels per frame: not a multiple of the 1024-pixel tile
    d = make_depth(rng, (F, H, W), ddtype)
    rgb = rng.integers(0, 256, size=(F, H, W, 3), dtype=np.uint8)
    q, t = rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10
    try:
        ctx.set_tuning("fuse_prefetch", 1)
        want = R.fuse_frames(d, q, t, out_dtype=odtype, ctx=ctx)
        want_u = R.unproject(d, out_dtype=odtype, ctx=ctx)
        want_c = R.fuse_frames_rgb(d, rgb, q, t, out_dtype=odtype, ctx=ctx)
        ctx.set_tuning("fuse_prefetch", 2)
        ctx.set_tuning("fuse_chunk_mb", 1)
        np.testing.assert_array_equal(R.fuse_frames(d, q, t, out_dtype=odtype, ctx=ctx), want)
        np.testing.assert_array_equal(R.unproject(d, out_dtype=odtype, ctx=ctx), want_u)
        got_c = R.fuse_frames_rgb(d, rgb, q, t, out_dtype=odtype, ctx=ctx)
        np.testing.assert_array_equal(got_c[0], want_c[0])
        np.testing.assert_array_equal(got_c[1], want_c[1])
        # the device entry points with everything resident (the host ones above go through the 32 MiB pipeline chunks)
        cam = ctx.camera(H, W, *R.REF_INTRINSICS)
        tab = R.pose_table(q, t)
        d_depth, d_pose = ctx.alloc(d.nbytes).upload(d), ctx.alloc(tab.nbytes).upload(tab)
        d_rgb, d_rgba = ctx.alloc(rgb.nbytes).upload(rgb), ctx.alloc(F * H * W * 4)
        d_out = ctx.alloc(F * H * W * 3 * np.dtype(odtype).itemsize)
        R.fuse_frames_device(ctx, cam, d_depth.ptr, ddtype, F, d_pose.ptr, d_out.ptr, odtype)
        np.testing.assert_array_equal(d_out.download(odtype, F * H * W * 3).reshape(-1, 3), want)
        R.fuse_frames_rgb_device(ctx, cam, d_depth.ptr, ddtype, F, d_pose.ptr, d_rgb.ptr, d_out.ptr, odtype, d_rgba.ptr)
        np.testing.assert_array_equal(d_out.download(odtype, F * H * W * 3).reshape(-1, 3), want_c[0])
        np.testing.assert_array_equal(d_rgba.download(np.uint32, F * H * W), want_c[1])
        for b in (d_depth, d_pose, d_rgb, d_rgba, d_out):
            b.free()
    finally:
        ctx.set_tuning("fuse_prefetch", 0)
        ctx.set_tuning("fuse_chunk_mb", 0)


def test_cache_prefetch_is_a_pure_read(R, ctx):
    """r3d_cache_prefetch sweeps a device buffer (any alignment, any length) and changes nothing."""
    L = importlib.import_module(R.__name__ + "._lib")
    rng = np.random.default_rng(8)
    data = rng.integers(0, 256, 1_000_003, dtype=np.uint8)
    buf = ctx.alloc(data.nbytes).upload(data)
    for off, nbytes in ((0, data.nbytes), (1, data.nbytes - 1), (7, 9), (3, 0), (16, 4096), (5, 100000)):
        L.check(ctx.lib.r3d_cache_prefetch(ctx.handle, buf.ptr + off, nbytes))
    np.testing.assert_array_equal(buf.download(np.uint8, data.nbytes), data)
    with pytest.raises(R.R3DError):
        L.check(ctx.lib.r3d_cache_prefetch(ctx.handle, None, 16))
    buf.free()


def test_auto_staging_above_64_mb_of_inputs_is_bit_identical(R, ctx):
    """150 frames of 1280x384 u8 = 73.7 MB of raster: above the auto threshold, so the default call stages the inputs through
    the Infinity Cache in 96 MB steps (here one step) -- and with colour (295 MB of inputs) in four.  Same bits as with the
    staging switched off."""
    rng = np.random.default_rng(9)
    F, H, W = 150, 384, 1280
    n = F * H * W
    d = rng.integers(0, 256, size=(F, H, W), dtype=np.uint8)
    rgb = rng.integers(0, 256, size=(F, H, W, 3), dtype=np.uint8)
    tab = R.pose_table(rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10)
    cam = ctx.camera(H, W, *R.REF_INTRINSICS)
    d_depth, d_pose, d_rgb = ctx.alloc(d.nbytes).upload(d), ctx.alloc(tab.nbytes).upload(tab), ctx.alloc(rgb.nbytes).upload(rgb)
    del rgb
    d_a, d_b, d_ca, d_cb = ctx.alloc(n * 12), ctx.alloc(n * 12), ctx.alloc(n * 4), ctx.alloc(n * 4)
    try:
        for colour in (False, True):
            for knob, out, rgba in ((1, d_a, d_ca), (0, d_b, d_cb)):
                ctx.set_tuning("fuse_prefetch", knob)
                if colour:
                    R.fuse_frames_rgb_device(ctx, cam, d_depth.ptr, np.uint8, F, d_pose.ptr, d_rgb.ptr, out.ptr, np.float32, rgba.ptr)
                else:
                    R.fuse_frames_device(ctx, cam, d_depth.ptr, np.uint8, F, d_pose.ptr, out.ptr, np.float32)
            a, b = d_a.download(np.uint32, n * 3), d_b.download(np.uint32, n * 3)
            assert np.array_equal(a, b)
            if colour:
                assert np.array_equal(d_ca.download(np.uint32, n), d_cb.download(np.uint32, n))
    finally:
        ctx.set_tuning("fuse_prefetch", 0)
        for buf in (d_depth, d_pose, d_rgb, d_a, d_b, d_ca, d_cb):
            buf.free()


def test_staging_follows_provenance_not_size(R, ctx):
    """`fuse_prefetch` auto since round 4: a launch's inputs are swept into the Infinity Cache first UNLESS the library has
    reason to presume they are there -- a launch on this device read those very bytes and little else since.  The library's
    own counter ("fuse_sweeps") says which launches were staged; every cloud is the same bit for bit."""
    L = importlib.import_module(R.__name__ + "._lib")
    rng = np.random.default_rng(12)
    F, H, W = 24, 384, 1280                                      # 11.8 MB of raster: above the 8 MB floor of the policy
    n = F * H * W
    d = rng.integers(0, 256, size=(F, H, W), dtype=np.uint8)
    tab = R.pose_table(rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10)
    cam = ctx.camera(H, W, *R.REF_INTRINSICS)
    d_depth, d_pose, d_out = ctx.alloc(n).upload(d), ctx.alloc(tab.nbytes).upload(tab), ctx.alloc(n * 12)
    d_other = ctx.alloc(n).upload(d)
    ctx.set_tuning("fuse_prefetch", 1)
    R.fuse_frames_device(ctx, cam, d_depth.ptr, np.uint8, F, d_pose.ptr, d_out.ptr, np.float32)
    want = d_out.download(np.uint32, n * 3)
    ctx.set_tuning("fuse_prefetch", 0)
    ctx.inputs_fresh()

    def launch(src=None):
        src = src or d_depth
        s0 = ctx.get_tuning("fuse_sweeps")
        L.check(ctx.lib.r3d_memset(ctx.handle, d_out.ptr, 0, n * 12))
        R.fuse_frames_device(ctx, cam, src.ptr, np.uint8, F, d_pose.ptr, d_out.ptr, np.float32)
        assert np.array_equal(d_out.download(np.uint32, n * 3), want)
        return ctx.get_tuning("fuse_sweeps") - s0

    try:
        assert launch() == 1                                     # never seen: staged
        assert launch() == 0 and launch() == 0                   # the previous launch has just read it: not again
        d_depth.upload(d)
        assert launch() == 1 and launch() == 0                   # an H2D copy rewrote it: DMA does not land in the cache
        L.check(ctx.lib.r3d_memcpy_d2d(ctx.handle, d_depth.ptr + 4096, d_other.ptr + 4096, 1 << 20))
        assert launch() == 1 and launch() == 0                   # any write through the library that overlaps the range
        ctx.inputs_fresh()
        assert launch() == 1 and launch() == 0                   # a foreign producer said so
        assert launch(d_other) == 1 and launch() == 0 and launch(d_other) == 0    # two rasters fit the budget side by side
        L.check(ctx.lib.r3d_cache_prefetch(ctx.handle, d_depth.ptr, n))
        ctx.inputs_fresh()
        L.check(ctx.lib.r3d_cache_prefetch(ctx.handle, d_depth.ptr, n))
        assert launch() == 0                                     # the caller swept it himself
        ctx.set_tuning("fuse_resident_mb", 16)                   # a cache that keeps 16 MB: the two rasters evict each other
        assert launch(d_other) == 1 and launch() == 1 and launch(d_other) == 1 and launch(d_other) == 0
        ctx.set_tuning("fuse_resident_mb", 8)                    # ... and one that cannot keep even one of them
        assert launch() == 1 and launch() == 1
        ctx.set_tuning("fuse_resident_mb", 128)
        # the record belongs to the DEVICE: a second context on it sees the first one's reads and writes
        other = R.Context(0)
        cam2 = other.camera(H, W, *R.REF_INTRINSICS)
        assert launch() == 1 and launch() == 0
        s0 = other.get_tuning("fuse_sweeps")
        R.fuse_frames_device(other, cam2, d_depth.ptr, np.uint8, F, d_pose.ptr, d_out.ptr, np.float32)
        other.sync()
        assert other.get_tuning("fuse_sweeps") - s0 == 0
        L.check(other.lib.r3d_memcpy_h2d(other.handle, d_depth.ptr, d.ctypes.data, n))
        other.sync()
        assert launch() == 1
        other.close()
        # forced on / off ignore the record
        ctx.set_tuning("fuse_prefetch", 2)
        assert launch() == 1 and launch() == 1
        ctx.set_tuning("fuse_prefetch", 1)
        ctx.inputs_fresh()
        assert launch() == 0
        # a freed range is forgotten: the next allocation may get the same addresses
        ctx.set_tuning("fuse_prefetch", 0)
        assert launch() in (0, 1) and launch() == 0
        assert ctx.get_tuning("fuse_inputs_fresh") >= 1          # ranges on record
        d_depth.free()
        d_depth = ctx.alloc(n).upload(d)
        assert launch(d_depth) == 1
        # below the policy's floor nothing is staged
        small = 12
        s0 = ctx.get_tuning("fuse_sweeps")
        ctx.inputs_fresh()
        R.fuse_frames_device(ctx, cam, d_depth.ptr, np.uint8, small, d_pose.ptr, d_out.ptr, np.float32)
        ctx.sync()
        assert ctx.get_tuning("fuse_sweeps") == s0
    finally:
        ctx.set_tuning("fuse_prefetch", 0)
        ctx.set_tuning("fuse_resident_mb", 128)
        for b in (d_depth, d_pose, d_out, d_other):
            b.free()
