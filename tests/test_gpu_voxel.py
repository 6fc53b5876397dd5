"""GPU: occupied-voxel set kernel + the OctoMap drop-in scripts against oracle/octomap_ref.py
(parity unpinned by the reference: the OctoMap library is absent; see the oracle's header)."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import PKG, ROOT, r3d as _r3d
from oracle import octomap_ref as OM

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def V():
    return importlib.import_module(PKG + ".voxelmap")


@pytest.fixture(scope="module")
def ctx():
    c = _r3d().Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("n,spread", [(1, 1.0), (7, 0.2), (1000, 1.0), (4097, 3.0), (300000, 8.0), (1000000, 2.0)])
def test_voxel_set_matches_oracle(V, ctx, n, spread):
    rng = np.random.default_rng(n)
    pts = (rng.normal(size=(n, 3)) * spread).astype(np.float32)
    vs = V.VoxelSet(0.1, max(1024, 2 * n), ctx)
    vs.insert(pts)
    st = vs.stats()
    want, dropped = OM.occupied_set(pts)
    assert st == {"voxels": len(want), "ignored_points": dropped, "overflow": 0}
    np.testing.assert_array_equal(vs.codes(), want)
    vs.insert(pts[: n // 2])                                   # re-inserting changes nothing
    assert vs.stats()["voxels"] == len(want)
    vs.clear()
    assert vs.stats()["voxels"] == 0
    vs.close()


def test_out_of_range_nonfinite_and_duplicates(V, ctx):
    pts = np.array([[-0.05, 0.0, 0.1], [3276.75, 0, 0], [3276.85, 0, 0], [-3276.75, 0, 0], [-3276.9, 0, 0],
                    [np.nan, 0, 0], [np.inf, 0, 0], [0.01, 0.02, 0.03], [0.04, 0.05, 0.06]] * 50, np.float32)
    vs = V.VoxelSet(0.1, 4096, ctx)
    vs.insert(pts)
    want, dropped = OM.occupied_set(pts)
    assert vs.stats() == {"voxels": len(want), "ignored_points": dropped, "overflow": 0}
    assert dropped == 4 * 50 and len(want) == 4
    np.testing.assert_array_equal(vs.codes(), want)
    vs.close()


def test_overflow_is_reported_and_voxelize_regrows(V, ctx):
    R = _r3d()
    rng = np.random.default_rng(0)
    pts = (rng.random((20000, 3)) * 50).astype(np.float32)
    vs = V.VoxelSet(0.1, 1024, ctx)
    vs.insert(pts)
    assert vs.stats()["overflow"] > 0
    with pytest.raises(R.R3DError):
        vs.codes()
    vs.close()
    codes, st = V.voxelize(pts, 0.1, ctx)
    np.testing.assert_array_equal(codes, OM.occupied_set(pts)[0])


def test_other_resolution(V, ctx):
    rng = np.random.default_rng(1)
    pts = (rng.normal(size=(5000, 3)) * 4).astype(np.float32)
    codes, _ = V.voxelize(pts, 0.25, ctx)
    np.testing.assert_array_equal(codes, OM.occupied_set(pts, 0.25)[0])
    assert V.format_bt(codes, 0.25) == OM.write_bt_bytes(codes, 0.25)


def test_fused_cloud_to_bt_on_device(V, ctx):
    """depth -> world cloud -> voxel set without leaving HBM, .bt equals the oracle's for the oracle's cloud."""
    R = _r3d()
    from oracle import fusion_ref as O
    rng = np.random.default_rng(2)
    F, H, W = 4, 48, 64
    d = rng.integers(1, 60, size=(F, H, W), dtype=np.uint8)
    q = rng.normal(size=(F, 4))
    t = rng.normal(size=(F, 3))
    K = (60.0, 60.0, 32, 24)
    cam = ctx.camera(H, W, *K)
    d_depth = ctx.alloc(d.nbytes).upload(d)
    tab = R.pose_table(q, t)
    d_pose = ctx.alloc(tab.nbytes).upload(tab)
    d_out = ctx.alloc(F * H * W * 12)
    R.fuse_frames_device(ctx, cam, d_depth.ptr, np.uint8, F, d_pose.ptr, d_out.ptr, np.float32)
    vs = V.VoxelSet(0.1, 1 << 16, ctx)
    vs.insert_device(d_out.ptr, F * H * W)
    codes = vs.codes()
    world32 = O.fuse_frames(d, q, t, *K).astype(np.float32)
    got32 = d_out.download(np.float32, F * H * W * 3).reshape(-1, 3)
    np.testing.assert_array_equal(codes, OM.occupied_set(got32)[0])
    # against the oracle's own cloud: identical unless a coordinate sits within an f32 ulp of a voxel face
    want = OM.occupied_set(world32)[0]
    assert len(np.setxor1d(codes, want)) <= 4
    vs.close()


def run_script(rel, cwd, *args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, PKG, rel)] + list(args), cwd=cwd, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_octomap_dropin_scripts(tmp_path, golden_dir):
    R = _r3d()
    scene = os.path.join(golden_dir, "scene3")
    os.makedirs(tmp_path / "bt")
    world_txt = os.path.join(scene, "point_world", "small_worldpoint_5_23_5.txt")
    run_script("octomap/txt_transfer_octomap.py", str(tmp_path), world_txt, "bt/w.bt")
    pts = R.cloud_io.read_xyz_txt(world_txt).astype(np.float32)
    assert (tmp_path / "bt" / "w.bt").read_bytes() == OM.write_bt_bytes(OM.occupied_set(pts)[0])[0]
    ply = os.path.join(scene, "ply", "small_035_p8.ply")
    run_script("octomap/ply_transfer_octomap.py", str(tmp_path), ply, "bt/p.bt")
    verts = R.cloud_io.read_ply(ply).astype(np.float32)[1:]         # the reference skips 8 lines: first vertex lost
    assert (tmp_path / "bt" / "p.bt").read_bytes() == OM.write_bt_bytes(OM.occupied_set(verts)[0])[0]
    run_script("other_tools/ply_transfer_octomap.py", str(tmp_path), ply, "bt/p2.bt")
    assert (tmp_path / "bt" / "p2.bt").read_bytes() == (tmp_path / "bt" / "p.bt").read_bytes()


@pytest.mark.parametrize("n", [0, 1, 2, 255, 256, 4095, 4096, 4097, 70001, 1500000])
@pytest.mark.parametrize("bits", [48, 64, 20])
def test_radix_sort_u64(ctx, n, bits):
    import ctypes as C
    L = importlib.import_module(PKG + "._lib")
    rng = np.random.default_rng(n + bits)
    keys = rng.integers(0, 2 ** 63, size=n, dtype=np.uint64) * 2 + rng.integers(0, 2, size=n, dtype=np.uint64)
    if bits < 64:
        keys &= np.uint64((1 << bits) - 1)
    if n > 10:
        keys[n // 2:n // 2 + 5] = keys[0]                        # duplicates
    buf = ctx.alloc(max(n * 8, 16))
    if n:
        buf.upload(keys)
    L.check(ctx.lib.r3d_sort_u64(ctx.handle, buf.ptr, n, bits))
    got = buf.download(np.uint64, n)
    np.testing.assert_array_equal(got, np.sort(keys))
    buf.free()


def test_insert_codes_and_single_rank_union(V, ctx, real_rccl):
    """Codes of one set folded into another == the set built from both clouds; codes beyond 48 bits are ignored; a world
    of one rank leaves the set as it is."""
    CM = importlib.import_module(PKG + ".comm")
    rng = np.random.default_rng(2)
    a = (rng.normal(size=(40000, 3)) * 3).astype(np.float32)
    b = (rng.normal(size=(30000, 3)) * 3 + 2).astype(np.float32)
    va, vb, vab = V.VoxelSet(0.1, 1 << 18, ctx), V.VoxelSet(0.1, 1 << 18, ctx), V.VoxelSet(0.1, 1 << 18, ctx)
    va.insert(a)
    vb.insert(b)
    vab.insert(np.concatenate([a, b]))
    cb = vb.codes()
    bad = np.array([1 << 50, (1 << 48) | 5], dtype=np.uint64)
    both = np.concatenate([cb, cb[:100], bad])                      # repeats and two non-keys
    d = ctx.alloc(both.nbytes).upload(both)
    va.insert_codes_device(d.ptr, both.shape[0])
    np.testing.assert_array_equal(va.codes(), vab.codes())
    assert va.stats()["ignored_points"] == 2
    comm = CM.Comm(ctx, CM.Comm.unique_id(), 0, 1)
    va.union_across(comm)
    np.testing.assert_array_equal(va.codes(), vab.codes())
    comm.close()
    d.free()
    for v in (va, vb, vab):
        v.close()


@pytest.mark.parametrize("res", [0.1, 0.05, 0.25, 0.3, 1.0 / 3.0])
def test_keys_of_points_on_and_next_to_voxel_faces(V, ctx, res):
    """Keys are OctoMap's (int)floor(factor * (double)x) + 32768.  Points exactly ON voxel faces, one and a few float ulps
    either side, at the range limits, non-finite and ordinary ones -- every key must equal the fp64 oracle's, at several
    resolutions (any cheaper key arithmetic has to pass this first; an fp32 fast path with an fp64 fallback did, and bought
    nothing: DESIGN.md 4.5)."""
    rng = np.random.default_rng(int(res * 1000))
    k = rng.integers(-32768, 32768, 60000)
    base = (k * res).astype(np.float32)                                    # (nearly) on a face
    pts = []
    for step in (0, 1, -1, 2, -2, 7, -7, 50, -50):
        x = base.copy()
        for _ in range(abs(step)):
            x = np.nextafter(x, np.float32(np.inf if step > 0 else -np.inf), dtype=np.float32)
        pts.append(x)
    x = np.concatenate(pts)
    xyz = np.stack([x, np.roll(x, 1), np.roll(x, 2)], 1).astype(np.float32)
    xyz = np.concatenate([xyz, (rng.normal(size=(50000, 3)) * 50).astype(np.float32),
                          np.array([[-32768 * res, 0, 0], [32768 * res, 0, 0], [np.nextafter(np.float32(32768 * res), np.float32(0)), 0, 0],
                                    [1e30, 0, 0], [-1e30, 0, 0], [0, np.nan, 0], [0, 0, np.inf]], np.float32)])
    want, dropped = OM.occupied_set(xyz, res)
    vs = V.VoxelSet(res, 1 << 21, ctx)
    vs.insert(xyz)
    st = vs.stats()
    assert st["overflow"] == 0 and st["ignored_points"] == dropped
    np.testing.assert_array_equal(vs.codes(), want)
    vs.close()


# ---- config 5 sizes: the multi-tile path of voxel_insert_kernel (round-2 kernel: contiguous tile runs per workgroup, the
# LDS set kept from tile to tile, the rotating fill counters and the wipe) only runs above num_cus*8*1024 = 2.1 M points.
# Everything below compares EXACT code sets + ignored / overflow counts with oracle/octomap_ref.occupied_set, and the
# kernel with the LDS set (voxel_dedupe 0/2) against the one without it (1) bitwise.

def room_views(n_frames, h, w, seed=0):
    """Views of the inside of a box room (3d_reconstruction_system_amd/synthetic.py): neighbouring pixels and rows land in the
    same 10 cm voxels (duplicate-heavy, like an indoor scan)."""
    return importlib.import_module(PKG + ".synthetic").room_views(n_frames, h, w, seed)


def fuse_on_device(ctx, depth, q, t, K, rgb=None):
    R = _r3d()
    F, H, W = depth.shape
    cam = ctx.camera(H, W, *K)
    d_depth = ctx.alloc(depth.nbytes).upload(depth)
    tab = R.pose_table(q, t)
    d_pose = ctx.alloc(tab.nbytes).upload(tab)
    d_xyz = ctx.alloc(F * H * W * 12)
    if rgb is None:
        R.fuse_frames_device(ctx, cam, d_depth.ptr, depth.dtype.type, F, d_pose.ptr, d_xyz.ptr, np.float32)
    else:
        d_rgb = ctx.alloc(rgb.nbytes).upload(rgb)
        d_rgba = ctx.alloc(F * H * W * 4)
        R.fuse_frames_rgb_device(ctx, cam, d_depth.ptr, depth.dtype.type, F, d_pose.ptr, d_rgb.ptr, d_xyz.ptr, np.float32, d_rgba.ptr)
        ctx.sync()
        d_rgb.free()
        d_rgba.free()
    ctx.sync()
    d_depth.free()
    d_pose.free()
    return d_xyz


def check_device_cloud_against_oracle(V, ctx, d_xyz, n, res, capacity):
    """Every way in: the CAS path with the LDS set and without it, the sort-merge path, and the library's own choice between
    them -- all must equal the oracle's set of the very same f32 cloud."""
    cloud = d_xyz.download(np.float32, n * 3).reshape(-1, 3)
    want, dropped = OM.occupied_set(cloud, res)
    got = {}
    for path, dedupe in ((1, 0), (1, 1), (2, 0), (0, 0)):
        ctx.set_tuning("voxel_path", path)
        ctx.set_tuning("voxel_dedupe", dedupe)
        try:
            vs = V.VoxelSet(res, capacity, ctx)
            vs.insert_device(d_xyz.ptr, n)
            assert vs.stats() == {"voxels": len(want), "ignored_points": dropped, "overflow": 0}, (path, dedupe, vs.stats(), len(want))
            if path:
                assert ctx.get_tuning("voxel_last_path") == path
            got[path, dedupe] = vs.codes()
            vs.insert_device(d_xyz.ptr, n)                        # idempotent at size
            assert vs.stats()["voxels"] == len(want)
            vs.close()
        finally:
            ctx.set_tuning("voxel_dedupe", 0)
            ctx.set_tuning("voxel_path", 0)
    for k in got:
        np.testing.assert_array_equal(got[k], want, err_msg=str(k))
    return want


@pytest.mark.parametrize("res", [0.1, 0.02, 0.005])
def test_room_scan_20M_points_multi_tile_runs_match_oracle(V, ctx, res):
    """>= 20 M points of a duplicate-heavy surface scene: 10 tiles per workgroup; at 0.1 m the LDS set persists over the whole
    run, at 0.02 m it fills and is wiped mid-run, at 0.005 m (about one voxel per pixel) nearly every tile wipes it."""
    depth, q, t, K = room_views(26, 768, 1024, seed=3)
    n = depth.size
    assert n >= 20_000_000
    d_xyz = fuse_on_device(ctx, depth, q, t, K)
    want = check_device_cloud_against_oracle(V, ctx, d_xyz, n, res, 1 << 26)
    assert len(want) > {0.1: 10_000, 0.02: 400_000, 0.005: 4_000_000}[res]
    d_xyz.free()


@pytest.mark.parametrize("order", ["shuffled", "lattice"])
def test_8M_all_distinct_points_wipe_every_tile(V, ctx, order):
    """Every point its own voxel (centres of a 256 x 256 x 128 lattice of 10 cm cells): each tile brings 1024 new codes, so
    the LDS set is wiped before every tile; lattice order also exercises runs of x-neighbours in one wave."""
    n = 256 * 256 * 128
    i = np.arange(n, dtype=np.int64)
    if order == "shuffled":
        i = np.random.default_rng(8).permutation(n)
    pts = np.stack([(i % 256) - 128, (i // 256) % 256 - 128, i // 65536 - 64], 1).astype(np.float32) * np.float32(0.1) + np.float32(0.05)
    d_xyz = ctx.alloc(n * 12).upload(pts)
    want = check_device_cloud_against_oracle(V, ctx, d_xyz, n, 0.1, 1 << 25)
    assert len(want) == n
    d_xyz.free()


@pytest.mark.parametrize("kind", ["room", "random"])
def test_fused_c5_geometry_cloud_to_voxels_matches_oracle(V, ctx, kind):
    """BASELINE config 5's own pipeline at its geometry: >= 10 frames of 1920x1080 f32 depth + RGB through the colour-carrying
    fused launch (r3d_fuse_frames_rgb), the cloud voxelised where it lies in HBM; 20.7 M points = 10 tiles per workgroup.
    'room': surfaces (about 60 points per voxel, like an indoor scan); 'random': random depth (about 1 voxel per point, what
    bench.py --workload c5 feeds)."""
    F, H, W = 10, 1080, 1920
    rng = np.random.default_rng(5)
    if kind == "room":
        depth, q, t, K = room_views(F, H, W, seed=5)
    else:
        depth = rng.random((F, H, W), dtype=np.float32) * 99.5 + 0.5
        q, t, K = rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10, (960.0, 960.0, 959.5, 539.5)
    rgb = rng.integers(0, 256, size=(F, H, W, 3), dtype=np.uint8)
    d_xyz = fuse_on_device(ctx, depth, q, t, K, rgb=rgb)
    want = check_device_cloud_against_oracle(V, ctx, d_xyz, F * H * W, 0.1, 1 << 26)
    assert len(want) > (10_000 if kind == "room" else 10_000_000)
    # and the map file: the library's serialiser on the GPU's codes == the oracle's on its own
    if kind == "room":
        assert V.format_bt(want, 0.1)[0] == OM.write_bt_bytes(want, 0.1)[0]
    d_xyz.free()


# ---- the cloud AND the map in one launch (r3d_fuse_frames_voxel): same cloud bytes, same rgba words, same set, same counters
# as r3d_fuse_frames_rgb followed by r3d_voxelset_insert -- and the set equal to the oracle's of that very cloud.

def fused_vs_two_calls(V, ctx, depth, q, t, K, rgb, res, capacity, depth_scale=1.0, oracle=True, overflowing=False):
    R = _r3d()
    F, H, W = depth.shape
    n = F * H * W
    cam = ctx.camera(H, W, *K)
    d_depth = ctx.alloc(depth.nbytes).upload(depth)
    d_pose = None
    if q is not None:
        tab = R.pose_table(q, t)
        d_pose = ctx.alloc(tab.nbytes).upload(tab)
    d_rgb = ctx.alloc(rgb.nbytes).upload(rgb) if rgb is not None else None
    bufs = []
    for fused in (False, True):
        d_xyz, d_rgba = ctx.alloc(n * 12), (ctx.alloc(n * 4) if rgb is not None else None)
        L_ = importlib.import_module(PKG + "._lib")
        L_.check(ctx.lib.r3d_memset(ctx.handle, d_xyz.ptr, 0xa5, n * 12))
        vs = V.VoxelSet(res, capacity, ctx)
        pose_ptr = d_pose.ptr if d_pose is not None else None
        if fused:
            R.fuse_frames_voxel_device(ctx, cam, d_depth.ptr, depth.dtype.type, F, pose_ptr, d_rgb.ptr if d_rgb else None, d_xyz.ptr,
                                       d_rgba.ptr if d_rgba else None, vs, depth_scale)
        else:
            if rgb is not None:
                R.fuse_frames_rgb_device(ctx, cam, d_depth.ptr, depth.dtype.type, F, pose_ptr, d_rgb.ptr, d_xyz.ptr, np.float32,
                                         d_rgba.ptr, depth_scale)
            elif d_pose is not None:
                R.fuse_frames_device(ctx, cam, d_depth.ptr, depth.dtype.type, F, pose_ptr, d_xyz.ptr, np.float32, depth_scale)
            else:
                R.unproject_device(ctx, cam, d_depth.ptr, depth.dtype.type, F, d_xyz.ptr, np.float32, depth_scale)
            vs.insert_device(d_xyz.ptr, n)
        cloud = d_xyz.download(np.uint32, n * 3)
        words = d_rgba.download(np.uint32, n) if d_rgba else None
        bufs.append((cloud, words, vs.stats(), None if overflowing else vs.codes()))
        vs.close()
        d_xyz.free()
        if d_rgba:
            d_rgba.free()
    (c0, w0, s0, k0), (c1, w1, s1, k1) = bufs
    np.testing.assert_array_equal(c1, c0)
    if rgb is not None:
        np.testing.assert_array_equal(w1, w0)
    if overflowing:   # which codes found no slot depends on the order of arrival: only the full table and the fact are common
        assert s1["voxels"] == s0["voxels"] == capacity and s1["overflow"] > 0 and s0["overflow"] > 0, (s1, s0)
        assert s1["ignored_points"] == s0["ignored_points"]
    else:
        assert s1 == s0, (s1, s0)
        np.testing.assert_array_equal(k1, k0)
    if oracle:
        want, dropped = OM.occupied_set(c1.view(np.float32).reshape(-1, 3), res)
        assert s1 == {"voxels": len(want), "ignored_points": dropped, "overflow": 0}
        np.testing.assert_array_equal(k1, want)
    for b in (d_depth, d_pose, d_rgb):
        if b is not None:
            b.free()
    return s1


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
@pytest.mark.parametrize("pose", [True, False])
@pytest.mark.parametrize("colour", [True, False])
@pytest.mark.parametrize("hw", [(48, 64), (37, 53), (1, 1), (33, 1024)])
def test_fused_cloud_and_voxels_equal_the_two_calls(V, ctx, dtype, pose, colour, hw):
    """whole tiles, ragged tiles (37 x 53: colour plane neither 16-byte aligned per frame nor a whole tile), one pixel; every depth
    type; with and without pose and colour"""
    H, W = hw
    F = 5
    rng = np.random.default_rng(H * 1000 + W)
    if dtype == np.float32:
        depth = (rng.random((F, H, W), dtype=np.float32) * 9.5 + 0.5)
        scale = 1.0
    elif dtype == np.uint16:
        depth = rng.integers(0, 65536, size=(F, H, W), dtype=np.uint16)
        scale = 1.0 / 5000.0
    else:
        depth = rng.integers(0, 256, size=(F, H, W), dtype=np.uint8)
        scale = 0.05
    q, t = (rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 3) if pose else (None, None)
    rgb = rng.integers(0, 256, size=(F, H, W, 3), dtype=np.uint8) if colour else None
    K = (W * 0.6, W * 0.6, (W - 1) / 2.0, (H - 1) / 2.0)
    st = fused_vs_two_calls(V, ctx, depth, q, t, K, rgb, 0.1, 1 << 19, scale)
    assert st["voxels"] >= 1


def test_fused_cloud_and_voxels_nonfinite_and_far_points_are_ignored(V, ctx):
    F, H, W = 3, 64, 128
    rng = np.random.default_rng(11)
    depth = rng.random((F, H, W), dtype=np.float32) * 5 + 0.5
    depth[0, :4] = np.nan
    depth[1, 10:12] = np.inf
    depth[2, 30:33] = 3.0e5                                   # far outside the +-3276.8 m key range at 0.1 m
    q, t = rng.normal(size=(F, 4)), rng.normal(size=(F, 3))
    rgb = rng.integers(0, 256, size=(F, H, W, 3), dtype=np.uint8)
    st = fused_vs_two_calls(V, ctx, depth, q, t, (80.0, 80.0, 63.5, 31.5), rgb, 0.1, 1 << 16)
    assert st["ignored_points"] >= (4 + 2 + 3) * W


def test_fused_cloud_and_voxels_overflow_is_reported(V, ctx):
    F, H, W = 2, 128, 256
    rng = np.random.default_rng(12)
    depth = rng.random((F, H, W), dtype=np.float32) * 50 + 0.5
    st = fused_vs_two_calls(V, ctx, depth, None, None, (200.0, 200.0, 127.5, 63.5), None, 0.05, 1 << 12, oracle=False, overflowing=True)
    assert st["overflow"] > 0 and st["voxels"] == 1 << 12


@pytest.mark.parametrize("kind,res", [("room", 0.1), ("room", 0.02), ("room", 0.005), ("random", 0.1)])
def test_fused_cloud_and_voxels_c5_geometry_20M_points(V, ctx, kind, res):
    """config 5's geometry, 10 frames = 20.7 M points = 10 tiles per workgroup: set persisting (0.1 m), wiped mid-run (0.02 m), wiped
    nearly every tile (0.005 m, random depth)"""
    F, H, W = 10, 1080, 1920
    rng = np.random.default_rng(6)
    if kind == "room":
        depth, q, t, K = room_views(F, H, W, seed=6)
    else:
        depth = rng.random((F, H, W), dtype=np.float32) * 99.5 + 0.5
        q, t, K = rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10, (960.0, 960.0, 959.5, 539.5)
    rgb = rng.integers(0, 256, size=(F, H, W, 3), dtype=np.uint8)
    st = fused_vs_two_calls(V, ctx, depth, q, t, K, rgb, res, 1 << 26)
    assert st["voxels"] > 10_000


def test_fused_cloud_and_voxels_staged_in_chunks(V, ctx):
    """inputs above the staging threshold go chunk by chunk (one launch per chunk, each with its own runs): same results"""
    F, H, W = 40, 384, 1280
    rng = np.random.default_rng(13)
    depth = rng.integers(1, 256, size=(F, H, W), dtype=np.uint8)
    q, t = rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 5
    ctx.set_tuning("fuse_chunk_mb", 4)
    try:
        st = fused_vs_two_calls(V, ctx, depth, q, t, (721.5, 721.5, 609.5, 172.8), None, 0.1, 1 << 24, 0.1, oracle=False)
    finally:
        ctx.set_tuning("fuse_chunk_mb", 0)
    assert st["voxels"] > 1000


@pytest.mark.parametrize("blocks", [1, 7, 301, 100000])
def test_fused_cloud_and_voxels_any_run_length(V, ctx, blocks):
    """one workgroup walking every tile (a run of 2400: many flushes, the rotating fill counters wrap hundreds of times), uneven
    runs (a last workgroup with fewer tiles, workgroups with none), one tile per workgroup"""
    depth, q, t, K = room_views(5, 384, 1280, seed=9)
    rng = np.random.default_rng(9)
    rgb = rng.integers(0, 256, size=depth.shape + (3,), dtype=np.uint8)
    ctx.set_tuning("fuse_blocks", blocks)
    try:
        st = fused_vs_two_calls(V, ctx, depth, q, t, K, rgb, 0.02, 1 << 22, oracle=(blocks == 7))
    finally:
        ctx.set_tuning("fuse_blocks", 0)
    assert st["voxels"] > 50_000 and st["overflow"] == 0


# ---- the sort-merge insert (path 2): keys into per-XCD bin segments, a second pass by piece, table regions updated in LDS -----

@pytest.mark.parametrize("n,log2cap,spread", [(1, 16, 1.0), (1000, 16, 3.0), (4097, 17, 8.0), (300_000, 20, 40.0), (300_000, 19, 40.0),
                                               (1_000_000, 21, 60.0), (2_000_003, 22, 25.0), (3_000_000, 29, 80.0)])
def test_sort_merge_insert_matches_oracle_and_the_cas_path(V, ctx, n, log2cap, spread):
    """forced at every size: one region's worth of keys up to millions, tables of 2^16 .. 2^29 slots (regions of 2048, 4096 and
    8192 slots), load factors up to ~0.55; non-finite / out-of-range points and repeats among them; then the SAME table through
    the CAS path (what one path placed the other must find) and back"""
    rng = np.random.default_rng(n + log2cap)
    pts = (rng.normal(size=(n, 3)) * spread).astype(np.float32)
    if n >= 1000:
        pts[::97] = pts[5]                                  # repeats far apart
        pts[3:200:7] = pts[2:199:7]                         # ... and in neighbouring lanes
        pts[11] = (np.nan, 0, 0)
        pts[12] = (0, np.inf, 0)
        pts[13] = (5000.0, 0, 0)
        pts[14] = (3276.75, 3276.75, 3276.75)               # key 65535 on every axis: the all-ones packed key
    want, dropped = OM.occupied_set(pts)
    d_xyz = ctx.alloc(n * 12).upload(pts)
    vs = V.VoxelSet(0.1, 1 << log2cap, ctx)
    try:
        ctx.set_tuning("voxel_path", 2)
        vs.insert_device(d_xyz.ptr, n)
        assert ctx.get_tuning("voxel_last_path") == 2
        assert vs.stats() == {"voxels": len(want), "ignored_points": dropped, "overflow": 0}, (vs.stats(), len(want), dropped)
        np.testing.assert_array_equal(vs.codes(), want)
        ctx.set_tuning("voxel_path", 1)
        vs.insert_device(d_xyz.ptr, n)                      # the CAS path finds every key the merge placed
        assert vs.stats()["voxels"] == len(want)
        ctx.set_tuning("voxel_path", 2)
        vs.insert_device(d_xyz.ptr, n)                      # ... and the merge finds its own
        assert vs.stats()["voxels"] == len(want) and vs.stats()["overflow"] == 0
        # a set that already holds half of the voxels (CAS-placed), then everything through the merge
        vs.clear()
        ctx.set_tuning("voxel_path", 1)
        vs.insert_device(d_xyz.ptr, n // 2)
        ctx.set_tuning("voxel_path", 2)
        vs.insert_device(d_xyz.ptr, n)
        assert vs.stats()["voxels"] == len(want)
        np.testing.assert_array_equal(vs.codes(), want)
    finally:
        ctx.set_tuning("voxel_path", 0)
        vs.close()
        d_xyz.free()


def test_sort_merge_into_a_crowded_2_27_slot_table_agrees_with_the_cas_path(V, ctx):
    """The 32-bit-slot merge (tables of 2^27 slots and more: a region is a piece) on a table that is NOT fresh and is crowded:
    100 M random points, half of them CAS-placed first (load 0.33: probe runs cross region ends, so regions hold keys of their
    neighbours -- the slots the merge must leave as they are), then everything through the merge (80 M distinct voxels, load 0.60:
    thousands of deferred keys).  Against the same points through the CAS path alone: same counters, same codes."""
    n = 100_000_000
    rng = np.random.default_rng(2027)
    d_xyz = ctx.alloc(n * 12)
    chunk = 10_000_000
    for k in range(n // chunk):
        pts = (rng.random((chunk, 3), dtype=np.float32) * 60.0 - 30.0)
        ctx.lib.r3d_memcpy_h2d(ctx.handle, d_xyz.ptr + k * chunk * 12, pts.ctypes.data, pts.nbytes)
        ctx.sync()
    a, b = V.VoxelSet(0.1, 1 << 27, ctx), V.VoxelSet(0.1, 1 << 27, ctx)
    try:
        ctx.set_tuning("voxel_path", 1)
        a.insert_device(d_xyz.ptr, n)
        b.insert_device(d_xyz.ptr, n // 2)
        ctx.set_tuning("voxel_path", 2)
        b.insert_device(d_xyz.ptr, n)
        assert ctx.get_tuning("voxel_last_path") == 2
        sa, sb = a.stats(), b.stats()
        assert sa["overflow"] == 0 and sb["overflow"] == 0 and sa["voxels"] == sb["voxels"] > 75_000_000, (sa, sb)
        assert sa["ignored_points"] * 3 == sb["ignored_points"] * 2            # (a saw the points once, b one and a half times)
        assert np.array_equal(a.codes(), b.codes())
        c = V.VoxelSet(0.1, 1 << 27, ctx)                                      # ... and a fresh table through the merge alone
        c.insert_device(d_xyz.ptr, n)
        assert c.stats()["voxels"] == sa["voxels"] and np.array_equal(c.codes(), a.codes())
        c.close()
    finally:
        ctx.set_tuning("voxel_path", 0)
        a.close()
        b.close()
        d_xyz.free()


@pytest.mark.parametrize("kind", ["eight_voxels", "hot_point", "one_bin_heavy", "tiny"])
def test_sort_merge_first_pass_segments_that_fill_up(V, ctx, kind):
    """The first pass writes into per-XCD bin segments sized for hashed keys (1.125 x the mean + 1024): keys that crowd into a
    bin must go in through the deferred list instead.  eight_voxels: 2 M points alternating between eight voxels (the
    neighbour-lane test removes nothing; every segment in use overflows many times over); hot_point: a cloud of distinct
    voxels with a fifth of its points at ONE place, scattered (pixels without depth end at the camera centre); one_bin_heavy:
    distinct voxels chosen so that a third of them share h48's lo byte; tiny: fewer points than there are segments."""
    rng = np.random.default_rng(len(kind))
    if kind == "eight_voxels":
        n = 2_000_000
        pts = np.zeros((n, 3), np.float32)
        pts[:, 0] = (np.arange(n) % 8) * 0.1 + 0.05
        pts[:, 1] = ((np.arange(n) // 8) % 2) * 0.1 + 0.05          # (16 voxels in fact: 8 x 2)
    elif kind == "hot_point":
        n = 3_000_000
        pts = (rng.normal(size=(n, 3)) * 60).astype(np.float32)
        pts[rng.random(n) < 0.2] = (1.234, -5.678, 9.1)
    elif kind == "one_bin_heavy":
        n = 1_500_000
        cand = (rng.normal(size=(6 * n, 3)) * 60).astype(np.float32)
        k = np.floor(cand.astype(np.float64) * 10.0).astype(np.int64) + 32768
        ok = ((k >= 0) & (k < 65536)).all(1)
        key = (k[:, 0] | (k[:, 1] << 16) | (k[:, 2] << 32)).astype(np.uint64)
        h = (key * np.uint64(0x9E3779B97F4B)) & np.uint64((1 << 48) - 1)            # r3d_voxel_dev.h: hash48
        lo = ((h >> np.uint64(32)) & np.uint64(0xff)).astype(np.int64)
        heavy = np.flatnonzero(ok & (lo == 77))[: n // 3]
        rest = np.flatnonzero(ok & (lo != 77))[: n - heavy.shape[0]]
        pts = cand[rng.permutation(np.concatenate([heavy, rest]))]
        n = pts.shape[0]
    else:
        n = 1500
        pts = (rng.normal(size=(n, 3)) * 9).astype(np.float32)
    want, dropped = OM.occupied_set(pts)
    d_xyz = ctx.alloc(n * 12).upload(pts)
    vs = V.VoxelSet(0.1, 1 << 23, ctx)
    try:
        ctx.set_tuning("voxel_path", 2)
        vs.insert_device(d_xyz.ptr, n)
        assert ctx.get_tuning("voxel_last_path") == 2
        assert vs.stats() == {"voxels": len(want), "ignored_points": dropped, "overflow": 0}, (vs.stats(), len(want), dropped)
        np.testing.assert_array_equal(vs.codes(), want)
        vs.insert_device(d_xyz.ptr, n)                      # again, into the table that now holds them all
        assert vs.stats()["voxels"] == len(want)
    finally:
        ctx.set_tuning("voxel_path", 0)
        vs.close()
        d_xyz.free()


def test_sort_merge_insert_of_more_than_one_chunk_agrees_with_the_cas_path(V, ctx):
    """The sort-merge insert walks a cloud in chunks of 2^27 points (its scratch is sized for one): 2^27 + 3 M points, nearly all
    in voxels of their own, into a table of 2^29 slots -- the second chunk meets a table that is no longer fresh (the workgroup
    form of the 32-bit merge, 8192-slot pieces).  No CPU oracle at this size: the CAS path's set of the same cloud is the check."""
    rng = np.random.default_rng(29)
    n = (1 << 27) + 3_000_000
    pts = rng.random((n, 3), dtype=np.float32)
    pts *= np.float32(900.0)
    pts -= np.float32(450.0)
    d_xyz = ctx.alloc(n * 12).upload(pts)
    del pts
    a, b = V.VoxelSet(0.1, 1 << 29, ctx), V.VoxelSet(0.1, 1 << 29, ctx)
    try:
        ctx.set_tuning("voxel_path", 2)
        a.insert_device(d_xyz.ptr, n)
        assert ctx.get_tuning("voxel_last_path") == 2
        ctx.set_tuning("voxel_path", 1)
        b.insert_device(d_xyz.ptr, n)
        sa, sb = a.stats(), b.stats()
        assert sa == sb and sa["overflow"] == 0 and sa["voxels"] > 0.9 * n, (sa, sb)
        np.testing.assert_array_equal(a.codes(), b.codes())
    finally:
        ctx.set_tuning("voxel_path", 0)
        a.close()
        b.close()
        d_xyz.free()


def test_sort_merge_insert_on_a_nearly_full_table_spills_and_still_agrees(V, ctx):
    """load factor 0.93: long probe runs cross region boundaries all the time (thousands of deferred keys), some keys were
    CAS-placed in the NEXT region before -- and the set is still the oracle's; a table that is too small reports overflow"""
    rng = np.random.default_rng(44)
    n = 62_500
    pts = (rng.random((n, 3)) * 12).astype(np.float32)
    want, dropped = OM.occupied_set(pts)
    assert 0.85 < len(want) / 65536 < 0.99
    d_xyz = ctx.alloc(n * 12).upload(pts)
    try:
        for first in (0, n // 3):
            vs = V.VoxelSet(0.1, 1 << 16, ctx)
            ctx.set_tuning("voxel_path", 1)
            vs.insert_device(d_xyz.ptr, first)
            ctx.set_tuning("voxel_path", 2)
            vs.insert_device(d_xyz.ptr, n)
            assert vs.stats() == {"voxels": len(want), "ignored_points": dropped + OM.occupied_set(pts[:first])[1], "overflow": 0}
            np.testing.assert_array_equal(vs.codes(), want)
            vs.close()
        big = (rng.random((200_000, 3)) * 30).astype(np.float32)
        d_big = ctx.alloc(big.nbytes).upload(big)
        vs = V.VoxelSet(0.1, 1 << 16, ctx)
        vs.insert_device(d_big.ptr, big.shape[0])
        st = vs.stats()
        assert st["voxels"] == 1 << 16 and st["overflow"] > 0
        vs.close()
        d_big.free()
    finally:
        ctx.set_tuning("voxel_path", 0)
        d_xyz.free()


def test_small_tables_and_small_inserts_take_the_cas_path(V, ctx):
    rng = np.random.default_rng(3)
    pts = (rng.normal(size=(50_000, 3)) * 5).astype(np.float32)
    d_xyz = ctx.alloc(pts.nbytes).upload(pts)
    want = OM.occupied_set(pts)[0]
    try:
        for cap, path, expect in ((1 << 15, 2, 1), (1 << 17, 0, 1), (1 << 17, 2, 2), (1 << 17, 1, 1)):
            ctx.set_tuning("voxel_path", path)
            vs = V.VoxelSet(0.1, cap, ctx)
            vs.insert_device(d_xyz.ptr, pts.shape[0])
            assert ctx.get_tuning("voxel_last_path") == expect, (cap, path)
            if cap > 1 << 15:
                np.testing.assert_array_equal(vs.codes(), want)
            vs.close()
    finally:
        ctx.set_tuning("voxel_path", 0)
        d_xyz.free()


@pytest.mark.parametrize("kind", ["random", "room"])
def test_the_library_picks_the_path_by_sampling_the_cloud(V, ctx, kind):
    """config 2's worst case (random depth: a voxel per point) goes through the sort-merge path, a room scan (tens of points per
    voxel) through the LDS-set kernel -- for r3d_voxelset_insert and for r3d_fuse_frames_voxel, which also chooses between its
    one-launch kernel and plain fuse + insert; clouds, colour words, sets and counters equal either way."""
    F, H, W = 12, 384, 1280                                    # 5.9 M points: above the 2^22 floor of the sampling
    rng = np.random.default_rng(17)
    if kind == "room":
        depth, q, t, K = room_views(F, H, W, seed=17)
    else:
        depth = rng.integers(1, 256, size=(F, H, W), dtype=np.uint8)
        q, t, K = rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10, _r3d().REF_INTRINSICS
    rgb = rng.integers(0, 256, size=(F, H, W, 3), dtype=np.uint8)
    expect = 2 if kind == "random" else 1
    d_xyz = fuse_on_device(ctx, depth, q, t, K)
    vs = V.VoxelSet(0.1, 1 << 24, ctx)
    vs.insert_device(d_xyz.ptr, F * H * W)
    assert ctx.get_tuning("voxel_last_path") == expect
    want, dropped = OM.occupied_set(d_xyz.download(np.float32, F * H * W * 3).reshape(-1, 3))
    assert vs.stats() == {"voxels": len(want), "ignored_points": dropped, "overflow": 0}
    np.testing.assert_array_equal(vs.codes(), want)
    vs.close()
    d_xyz.free()
    for path in (0, 1, 2):
        ctx.set_tuning("voxel_path", path)
        try:
            st = fused_vs_two_calls(V, ctx, depth, q, t, K, rgb, 0.1, 1 << 24)
            assert ctx.get_tuning("voxel_last_path") == (path or expect)
        finally:
            ctx.set_tuning("voxel_path", 0)
        assert st["voxels"] == len(want)


@pytest.mark.parametrize("res,expect", [(0.5, 2), (3.0, 1)])
def test_surfaces_with_a_few_points_per_voxel_take_the_sort_merge_path(V, ctx, res, expect):
    """The choice is a cost estimate, not a fixed ratio: slanted planes whose neighbouring points share voxels a little (2.7
    points per voxel at 0.5 m) are 3x faster through the sort-merge path and must take it; the same cloud at 3 m voxels (tens of
    points per voxel) stays with the LDS-set kernel.  Either way the set is the oracle's.  (tools/voxel_path_crossover.py)"""
    F, H, W = 12, 384, 1280
    rng = np.random.default_rng(7)
    yy, xx = np.mgrid[0:H, 0:W]
    depth = np.stack([np.clip(40 + (xx // 8 + yy // 6 + 3 * f) % 200, 1, 255) for f in range(F)]).astype(np.uint8)
    q, t = rng.normal(size=(F, 4)), rng.normal(size=(F, 3))
    d_xyz = fuse_on_device(ctx, depth, q, t, _r3d().REF_INTRINSICS)
    vs = V.VoxelSet(res, 1 << 24, ctx)
    try:
        vs.insert_device(d_xyz.ptr, F * H * W)
        assert ctx.get_tuning("voxel_last_path") == expect
        want, dropped = OM.occupied_set(d_xyz.download(np.float32, F * H * W * 3).reshape(-1, 3), res)
        assert vs.stats() == {"voxels": len(want), "ignored_points": dropped, "overflow": 0}
        np.testing.assert_array_equal(vs.codes(), want)
    finally:
        vs.close()
        d_xyz.free()
