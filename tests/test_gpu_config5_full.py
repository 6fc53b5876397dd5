"""GPU: BASELINE config 5 WHOLE on one MI355X, inside the driver's gate -- 2000 frames of 1920x1080 f32 depth + RGB
(4,147,200,000 points), the fused cloud carrying colour AND its occupied-voxel set from ONE call (r3d_fuse_frames_voxel),
95 GB of the 288 GB of HBM.  (BASELINE splits these frames over 8 GPUs; one GPU holds them all, which is the harder case for
the addressing.)  The rasters are 10 distinct host frames of a smooth scene replicated 200 times on the device -- surfaces,
like a scan: tens of points per voxel -- and the 2000 poses are all different.

What is pinned, and against what:
  * the cloud: three whole frames (first, one beyond byte 2^35 of the cloud, last) against oracle/fusion_ref.py, colour words
    against the rasters;
  * the map on a 20.7 M-point sub-range (10 frames): the same entry point into a fresh set -> its codes EQUAL
    oracle/octomap_ref.py's occupied set of the downloaded cloud, ignored points equal, overflow 0;
  * the map of all 4.1 G points: count / ignored / overflow equal between the one-call form and fuse + insert as two calls
    (a different kernel pair), the codes equal between the two, the sub-range's set contained in it, and the counters
    equal to the code list's length.
Ordered after every parity file (tests/conftest.py)."""
import importlib
import time

import numpy as np
import pytest

from helpers import PKG
from oracle import fusion_ref as O
from oracle import octomap_ref as OM

pytestmark = pytest.mark.gpu


def test_config5_whole_2000_frames_cloud_colour_and_voxel_map_from_one_call():
    R = importlib.import_module(PKG)
    L = importlib.import_module(PKG + "._lib")
    V = importlib.import_module(PKG + ".voxelmap")
    t_start = time.perf_counter()
    F, H, W, U = 2000, 1080, 1920, 10
    per = H * W
    n = F * per
    rng = np.random.default_rng(555)
    jj, ii = np.mgrid[0:H, 0:W]
    depth = np.stack([8.0 + 3.0 * np.sin(ii / (90.0 + 7 * k)) * np.cos(jj / (70.0 + 5 * k)) + 0.02 * rng.random((H, W)) for k in range(U)]).astype(np.float32)
    rgb = rng.integers(0, 256, size=(U, H, W, 3), dtype=np.uint8)
    q, t = rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10
    intr = (960.0, 960.0, 959.5, 539.5)
    ctx = R.Context(0)
    lib = ctx.lib
    cam = ctx.camera(H, W, *intr)
    table = R.pose_table(q, t)
    d_depth, d_rgb, d_pose = ctx.alloc(n * 4), ctx.alloc(n * 3), ctx.alloc(table.nbytes).upload(table)
    d_depth.upload(depth)
    d_rgb.upload(rgb)
    for k in range(1, F // U):      # replicate the 10 frames 200 times in HBM
        L.check(lib.r3d_memcpy_d2d(ctx.handle, d_depth.ptr + k * U * per * 4, d_depth.ptr, U * per * 4))
        L.check(lib.r3d_memcpy_d2d(ctx.handle, d_rgb.ptr + k * U * per * 3, d_rgb.ptr, U * per * 3))
    d_xyz, d_rgba = ctx.alloc(n * 12), ctx.alloc(n * 4)
    vs = V.VoxelSet(0.1, 1 << 28, ctx)
    R.fuse_frames_voxel_device(ctx, cam, d_depth.ptr, np.float32, F, d_pose.ptr, d_rgb.ptr, d_xyz.ptr, d_rgba.ptr, vs)   # THE call
    st = vs.stats()
    codes = vs.codes()
    assert st["overflow"] == 0 and st["voxels"] == len(codes) > 1_000_000 and np.all(codes[1:] > codes[:-1])

    def frames_of(buf, elem_bytes, dtype, f0, k):
        out = np.empty(k * per * elem_bytes // np.dtype(dtype).itemsize, dtype)
        L.check(lib.r3d_download(ctx.handle, out.ctypes.data, buf.ptr + f0 * per * elem_bytes, out.nbytes))
        return out

    # -- the cloud and its colour: three whole frames against the oracle (frame 1440 starts at byte 35.8e9 > 2^35 of the cloud)
    want_rgba = (rgb.reshape(U, per, 3).astype(np.uint32) * np.array([1, 256, 65536], dtype=np.uint32)).sum(2, dtype=np.uint32)
    for f in (0, 1440, F - 1):
        got = frames_of(d_xyz, 12, np.float32, f, 1).reshape(-1, 3)
        want = O.fuse_frames(depth[f % U:f % U + 1], q[f:f + 1], t[f:f + 1], *intr)
        e_norm, e_comp = O.parity_errors(got, want)
        assert e_norm <= 1e-6 and e_comp <= 1e-4, (f, e_norm, e_comp)
        assert np.array_equal(frames_of(d_rgba, 4, np.uint32, f, 1), want_rgba[f % U]), f

    # -- the map on a 20.7 M-point sub-range, pinned by the oracle: frames [1000, 1010) through the same entry point
    f0, k = 1000, 10
    sub = V.VoxelSet(0.1, 1 << 24, ctx)
    d_xyz2, d_rgba2 = ctx.alloc(k * per * 12), ctx.alloc(k * per * 4)
    R.fuse_frames_voxel_device(ctx, cam, d_depth.ptr + f0 * per * 4, np.float32, k, d_pose.ptr + f0 * 96, d_rgb.ptr + f0 * per * 3,
                               d_xyz2.ptr, d_rgba2.ptr, sub)
    cloud_sub = frames_of(d_xyz2, 12, np.float32, 0, k).reshape(-1, 3)
    assert np.array_equal(cloud_sub, frames_of(d_xyz, 12, np.float32, f0, k).reshape(-1, 3))      # the big launch wrote the same bits there
    want_codes, dropped = OM.occupied_set(cloud_sub, 0.1)
    sub_codes, sub_st = sub.codes(), sub.stats()
    assert np.array_equal(sub_codes, want_codes) and sub_st["ignored_points"] == dropped and sub_st["overflow"] == 0
    assert sub_st["voxels"] == len(want_codes) and cloud_sub.shape[0] >= 20_000_000
    at = np.searchsorted(codes, want_codes)                                                        # ... and all of them are in the whole map
    assert at.max() < len(codes) and np.array_equal(codes[at], want_codes)
    sub.close()
    for b in (d_xyz2, d_rgba2):
        b.free()

    # -- the whole map a second way: the cloud that is already in HBM through r3d_voxelset_insert (other kernels)
    two = V.VoxelSet(0.1, 1 << 28, ctx)
    two.insert_device(d_xyz.ptr, n)
    st2 = two.stats()
    assert st2 == st, (st2, st)
    assert np.array_equal(two.codes(), codes)
    two.close()
    vs.close()
    for b in (d_depth, d_rgb, d_pose, d_xyz, d_rgba):
        b.free()
    ctx.close()
    print("config 5 whole: %d voxels, %.0f s" % (st["voxels"], time.perf_counter() - t_start))
