#!/usr/bin/env python3
"""Run every kernel of the library at its BASELINE-config size with default tuning (for rocprofv3
--kernel-trace --stats) and print HIP-event timings beside the algorithmic work of each."""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
r3d = importlib.import_module("3d_reconstruction_system_amd")
L = importlib.import_module("3d_reconstruction_system_amd._lib")
icp = importlib.import_module("3d_reconstruction_system_amd.icp")
V = importlib.import_module("3d_reconstruction_system_amd.voxelmap")


def timed(ctx, fn, iters):
    """Median of 5 groups after >= 60 ms of warm-up (an idle GPU boosts, then dips for ~20 ms before it settles)."""
    import time
    t0 = time.perf_counter()
    k = 0
    while k < 3 or time.perf_counter() - t0 < 0.06:
        fn()
        k += 1
        if k % 5 == 0:
            ctx.sync()
    ctx.sync()
    per = max(iters // 5, 1)
    groups = []
    for _ in range(5):
        ctx.timer_start()
        for _ in range(per):
            fn()
        groups.append(ctx.timer_stop() / per)
    return sorted(groups)[2]


def main():
    short = "--short" in sys.argv          # PMC passes: few launches per kernel, no brute-force NN
    extra = "--extra" in sys.argv          # also the big-batch / staged launches (NOT under rocprofv3: same kernel names, other sizes)
    ctx = r3d.Context(0)
    rng = np.random.default_rng(1234)
    out = {}
    # C2: fused unproject + SE(3), 100 x 384 x 1280 u8 -> f32
    F, H, W = 100, 384, 1280
    n = F * H * W
    depth = rng.integers(1, 256, size=(F, H, W), dtype=np.uint8)
    tab = r3d.pose_table(rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10)
    d_depth, d_pose, d_xyz = ctx.alloc(n).upload(depth), ctx.alloc(tab.nbytes).upload(tab), ctx.alloc(n * 12)
    cam = ctx.camera(H, W, *r3d.REF_INTRINSICS)
    for _ in range(50):
        r3d.fuse_frames_device(ctx, cam, d_depth.ptr, np.uint8, F, d_pose.ptr, d_xyz.ptr, np.float32)
    ms = timed(ctx, lambda: r3d.fuse_frames_device(ctx, cam, d_depth.ptr, np.uint8, F, d_pose.ptr, d_xyz.ptr, np.float32), 200)
    out["fuse_u8_f32"] = {"ms": ms, "GBps": n * 13 / ms / 1e6, "Gpts": n / ms / 1e6, "bound": "hbm", "bytes_per_point": 13}
    ms = timed(ctx, lambda: r3d.unproject_device(ctx, cam, d_depth.ptr, np.uint8, F, d_xyz.ptr, np.float32), 200)
    out["unproject_u8_f32"] = {"ms": ms, "GBps": n * 13 / ms / 1e6, "Gpts": n / ms / 1e6, "bound": "hbm", "bytes_per_point": 13}
    d_xyz64 = ctx.alloc(n * 24)
    ms = timed(ctx, lambda: r3d.fuse_frames_device(ctx, cam, d_depth.ptr, np.uint8, F, d_pose.ptr, d_xyz64.ptr, np.float64), 50)
    out["fuse_u8_f64"] = {"ms": ms, "GBps": n * 25 / ms / 1e6, "Gpts": n / ms / 1e6, "bound": "hbm", "bytes_per_point": 25}
    if extra:
        # C4 on ONE GPU: 1000 frames (491.5 M points, 6.4 GB of traffic) -- far beyond the Infinity Cache, inputs staged (default)
        F4 = 1000
        n4 = F4 * H * W
        d_depth4, d_xyz4 = ctx.alloc(n4), ctx.alloc(n4 * 12)
        L.check(ctx.lib.r3d_memset(ctx.handle, d_depth4.ptr, 0x41, n4))
        tab4 = r3d.pose_table(rng.normal(size=(F4, 4)), rng.normal(size=(F4, 3)) * 10)
        d_pose4 = ctx.alloc(tab4.nbytes).upload(tab4)
        for knob, key in ((0, "fuse_u8_f32_1000_frames"), (1, "fuse_u8_f32_1000_frames_unstaged")):
            ctx.set_tuning("fuse_prefetch", knob)
            ms = timed(ctx, lambda: r3d.fuse_frames_device(ctx, cam, d_depth4.ptr, np.uint8, F4, d_pose4.ptr, d_xyz4.ptr, np.float32), 20)
            out[key] = {"ms": ms, "GBps": n4 * 13 / ms / 1e6, "Gpts": n4 / ms / 1e6, "bound": "hbm", "bytes_per_point": 13}
        ctx.set_tuning("fuse_prefetch", 0)
        for b in (d_depth4, d_xyz4, d_pose4):
            b.free()
    d_xyz64.free()
    # colour carried through the fused launch (+7 B/point)
    rgb = rng.integers(0, 256, size=(F, H, W, 3), dtype=np.uint8)
    d_rgb, d_rgba = ctx.alloc(rgb.nbytes).upload(rgb), ctx.alloc(n * 4)
    # this loop re-reads the same 196 MB of inputs: with the staging sweep off they are served by the Infinity Cache (the
    # round-1/2 figure); the library's default stages inputs of that size (they would not be cached in a real pass)
    ctx.set_tuning("fuse_prefetch", 1)
    ms = timed(ctx, lambda: r3d.fuse_frames_rgb_device(ctx, cam, d_depth.ptr, np.uint8, F, d_pose.ptr, d_rgb.ptr, d_xyz.ptr,
                                                       np.float32, d_rgba.ptr), 20 if short else 100)
    out["fuse_rgb_u8_f32"] = {"ms": ms, "GBps": n * 20 / ms / 1e6, "Gpts": n / ms / 1e6, "bound": "hbm", "bytes_per_point": 20,
                              "inputs": "cached (same buffers every launch), staging off"}
    ctx.set_tuning("fuse_prefetch", 0)
    if extra:
        ms = timed(ctx, lambda: r3d.fuse_frames_rgb_device(ctx, cam, d_depth.ptr, np.uint8, F, d_pose.ptr, d_rgb.ptr, d_xyz.ptr,
                                                           np.float32, d_rgba.ptr), 20 if short else 100)
        out["fuse_rgb_u8_f32_default"] = {"ms": ms, "GBps": n * 20 / ms / 1e6, "Gpts": n / ms / 1e6, "bound": "hbm", "bytes_per_point": 20,
                                      "inputs": "library default: 196 MB of inputs per launch are staged through the Infinity Cache"}
    d_rgb.free()
    d_rgba.free()
    r3d.fuse_frames_device(ctx, cam, d_depth.ptr, np.uint8, F, d_pose.ptr, d_xyz.ptr, np.float32)
    # apply-T on the fused cloud, in place semantics excluded: separate output
    d_xyz2 = ctx.alloc(n * 12)
    T = np.eye(4)
    T[:3, 3] = (1, 2, 3)
    ms = timed(ctx, lambda: r3d.apply_T_device(ctx, d_xyz.ptr, np.float32, n, T, d_xyz2.ptr, np.float32), 100)
    out["apply_T_f32"] = {"ms": ms, "GBps": n * 24 / ms / 1e6, "Gpts": n / ms / 1e6, "bound": "hbm", "bytes_per_point": 24}
    # voxel insert of the fused cloud (sparse synthetic worst case: nearly one voxel per point)
    vs = V.VoxelSet(0.1, 2 * n, ctx)

    def ins():
        vs.clear()
        vs.insert_device(d_xyz.ptr, n)
    ms_clear0 = timed(ctx, vs.clear, 10)
    per_path = {}
    for label, path in (("cas_lds_set", 1), ("sort_merge", 2)):
        ctx.set_tuning("voxel_path", path)
        per_path[label] = timed(ctx, ins, 10) - ms_clear0
    ctx.set_tuning("voxel_path", 0)
    ms_both = timed(ctx, ins, 10)
    per_path["auto_took_path"] = ctx.get_tuning("voxel_last_path")
    st = vs.stats()
    import time as _t
    codes = vs.codes()
    t0 = _t.perf_counter()
    for _ in range(3):
        codes = vs.codes()
    out["voxel_compact_sort_download"] = {"ms": (_t.perf_counter() - t0) / 3 * 1e3, "codes": int(codes.shape[0]),
                                          "what": "1.07 GB table -> compaction + 48-bit radix sort + D2H of the codes"}
    ms_clear = timed(ctx, vs.clear, 10)
    out["voxel_insert"] = {"ms": ms_both - ms_clear, "Gpts": n / (ms_both - ms_clear) / 1e6, "voxels": st["voxels"],
                           "paths_ms": per_path,
                           "bound": "sort-merge insert: HBM streams (20 + 48 + 8 + 8 B/point + 8..16 B/table slot); the CAS path: scattered "
                                    "64-bit atomics (19 G/s measured ceiling)",
                           "Gatomics_if_cas": st["voxels"] / per_path["cas_lds_set"] / 1e6}
    # the cloud and its voxels from one launch (r3d_fuse_frames_voxel) on the same frames: the worst case for it (random depth)
    def one_launch():
        vs.clear()
        r3d.fuse_frames_voxel_device(ctx, cam, d_depth.ptr, np.uint8, F, d_pose.ptr, None, d_xyz2.ptr, None, vs)
    ms_one = timed(ctx, one_launch, 10) - ms_clear
    out["fuse_voxel_one_launch"] = {"ms": ms_one, "Gpts": n / ms_one / 1e6, "voxels": vs.stats()["voxels"], "form_taken": ctx.get_tuning("voxel_last_path"),
                                    "two_calls_ms": out["fuse_u8_f32"]["ms"] + out["voxel_insert"]["ms"] if "fuse_u8_f32" in out else None}
    vs.close()
    d_xyz2.free()
    # C3: ICP on two 500k clouds
    m = 500000
    tgt = (rng.random((m, 3)) * 20).astype(np.float32)
    src = (tgt[rng.permutation(m)] * 1.01 + 0.02).astype(np.float32)
    dev = icp.IcpDevice(src, tgt, ctx, culled=False)
    if not short:
        ms = timed(ctx, dev.nn, 3)
        out["icp_nn_bruteforce_500k"] = {"ms": ms, "Tpairs": m * m / ms / 1e9, "TFLOPs_at_8_flop_per_pair": m * m * 8 / ms / 1e9,
                                         "bound": "fp32 VALU"}
    devc = icp.IcpDevice(src, tgt, ctx, culled=True)
    ctx.set_tuning("nn_warm", 1)     # the search from nothing (an ICP loop's first query)
    ms = timed(ctx, devc.nn, 10)
    swept = devc.nn(want_stats=True)
    out["icp_nn_culled_500k"] = {"ms": ms, "tile_sweeps_per_workgroup": swept / -(-m // 256), "tiles": -(-m // 1024),
                                 "note": "cold: same indices and distances as the brute-force sweep"}
    ctx.set_tuning("nn_warm", 3)     # every later query of a loop: bounds from the previous matches, wave-local kernel
    ms = timed(ctx, devc.nn, 10)
    ctx.set_tuning("nn_warm", 0)
    out["icp_nn_culled_500k_warm"] = {"ms": ms, "note": "warm (sources unmoved since the previous query: the bound is the answer); "
                                                          "bit-identical results"}
    ms = timed(ctx, lambda: devc.nn_sums(), 10)
    out["icp_nn_culled_fused_sums_500k"] = {"ms_incl_144B_D2H": ms}
    devc.state_reset()
    ms = timed(ctx, lambda: devc.iterate(6), 10) / 6
    out["icp_whole_iteration_500k"] = {"ms": ms, "what": "culled NN (warm from the 2nd of the six iterations of an enqueue) + fused 18 sums + device Umeyama solve + apply, no host sync"}
    devc.free()
    # index builds (bbox -> frame -> keys -> sort -> gather -> boxes), five more for the statistics
    import time
    d_t = ctx.alloc(tgt.nbytes).upload(tgt)
    t0 = time.perf_counter()
    for _ in range(5):
        ix = icp.NNIndex(ctx, d_t.ptr, m)
        ix.close()
    ctx.sync()
    out["nn_index_build_500k"] = {"ms_incl_alloc_free": (time.perf_counter() - t0) / 5 * 1e3}
    d_t.free()
    # the reference's own ICP case: two 480x640 single views, rigid point-to-plane
    S = importlib.import_module("3d_reconstruction_system_amd.synthetic")
    v2 = S.two_views(480, 640, yaw_deg=15.0, baseline=(0.35, 0.05, -0.2), depth_noise=0.001, seed=1)
    pa, pb = r3d.unproject(v2["depth_a"], v2["K"], ctx=ctx), r3d.unproject(v2["depth_b"], v2["K"], ctx=ctx)
    devp = icp.PlaneIcpDevice(pb, pa, (480, 640), ctx=ctx)
    devp.move_source(v2["T_ab"])
    devp.state_reset()
    ms = timed(ctx, lambda: devp.iterate(6), 5 if short else 20) / 6
    out["plane_icp_iteration_480x640"] = {"ms": ms, "what": "culled NN + residuals/classes + 24-class selection (8 launches) + 29 sums + "
                                                              "device 6x6 solve + move, one enqueue, no host sync"}
    d_n = ctx.alloc(pa.nbytes)
    ms = timed(ctx, lambda: L.check(ctx.lib.r3d_normals_organized(ctx.handle, devp.d_tgt.ptr, 1, 480, 640, 0.05, None, d_n.ptr)), 50)
    out["normals_480x640"] = {"ms": ms, "GBps": 480 * 640 * 24 / ms / 1e6}
    d_n.free()
    devp.free()
    t0 = time.perf_counter()
    for _ in range(20):
        dev.sums()
    out["icp_accumulate_500k"] = {"ms_incl_sync_and_D2H": (time.perf_counter() - t0) / 20 * 1e3, "bytes_per_pair": 28}
    dev.free()
    ctx.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
