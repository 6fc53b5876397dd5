#!/usr/bin/env python3
"""Interleaved A/B timing of kernel variants on one MI355X (device-resident data, HIP events
on the launch stream).  Usage: python tools/bench_variants.py [fuse|apply|nn|all] [--frames F]"""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
r3d = importlib.import_module("3d_reconstruction_system_amd")


def time_ms(ctx, fn, iters):
    fn()
    ctx.sync()
    ctx.timer_start()
    for _ in range(iters):
        fn()
    return ctx.timer_stop() / iters


# The fused / apply kernel variants left the library in round 2: their A/B harness is tools/ab_kernels.hip
# (make -C tools ab_kernels; history in profiles/variants_r01.md and profiles/r02_ab_kernels.log).


def bench_nn(ctx, n, m, rounds):
    L = importlib.import_module("3d_reconstruction_system_amd._lib")
    rng = np.random.default_rng(7)
    tgt = (rng.random(size=(m, 3)) * 20).astype(np.float32)
    src = (rng.random(size=(n, 3)) * 20).astype(np.float32)
    d_src = ctx.alloc(src.nbytes).upload(src)
    d_tgt = ctx.alloc(tgt.nbytes).upload(tgt)
    d_idx = ctx.alloc(n * 4)
    d_d2 = ctx.alloc(n * 4)
    print("nn %d x %d = %.3e pairs" % (n, m, n * m))
    for S in (1, 2, 4):
        ctx.set_tuning("nn_variant", S)
        ts = []
        for _ in range(rounds):
            ts.append(time_ms(ctx, lambda: L.check(ctx.lib.r3d_icp_nn(ctx.handle, d_src.ptr, n, d_tgt.ptr, m,
                                                                       d_idx.ptr, d_d2.ptr)), 1))
        med = np.median(ts)
        print("S=%d  med %.3f ms  %.2f Tpairs/s  (%.1f TFLOP/s at 8 flop/pair)"
              % (S, med, n * m / med / 1e9, n * m * 8 / med / 1e9))
    ctx.set_tuning("nn_variant", 0)
    icp = importlib.import_module("3d_reconstruction_system_amd.icp")
    import time
    for label, s_arr in (("uniform cube", src), ("ICP-like: target subset moved by s=1.01, 0.5 deg", None)):
        if s_arr is None:
            s_arr = (tgt[rng.permutation(m)[:n]] * 1.01 + 0.02).astype(np.float32)
        t0 = time.perf_counter()
        dev = icp.IcpDevice(s_arr, tgt, ctx, culled=True)
        ctx.sync()
        t_build = (time.perf_counter() - t0) * 1e3
        for S in (1, 2, 4):
            ctx.set_tuning("nn_variant", S)
            ts = [time_ms(ctx, dev.nn, 3) for _ in range(rounds)]
            swept = dev.nn(want_stats=True)
            blocks = -(-n // (256 * S))
            print("culled %-50s S=%d  med %.3f ms (= %.0f Tpairs/s brute-force equivalent), "
                  "%.1f of %d tiles swept per workgroup; upload+index build %.1f ms"
                  % (label, S, np.median(ts), n * m / np.median(ts) / 1e9, swept / blocks, -(-m // 1024), t_build))
        ctx.set_tuning("nn_variant", 0)
        dev.free()


def bench_voxel(ctx, F, H, W, rounds):
    V = importlib.import_module("3d_reconstruction_system_amd.voxelmap")
    L = importlib.import_module("3d_reconstruction_system_amd._lib")
    rng = np.random.default_rng(1234)
    n = F * H * W
    jj, ii = np.mgrid[0:H, 0:W]
    smooth = np.clip(40 + 30 * np.sin(ii / 97.0) * np.cos(jj / 61.0), 1, 255).astype(np.uint8)   # surfaces, like a real depth map
    for label, hi, scale in (("depth 1..255 random (sparse: most points their own voxel)", 256, 1.0),
                             ("depth 1..15 random (about 4 points per voxel)", 16, 1.0),
                             ("smooth surfaces, depth in 2 cm units (0.2-2 m: real indoor scale)", 0, 0.02)):
        depth = rng.integers(1, hi, size=(F, H, W), dtype=np.uint8) if hi else np.broadcast_to(smooth, (F, H, W)).copy()
        table = r3d.pose_table(rng.normal(size=(F, 4)), rng.normal(size=(F, 3)))
        d_depth = ctx.alloc(n).upload(depth)
        d_pose = ctx.alloc(table.nbytes).upload(table)
        d_out = ctx.alloc(n * 12)
        cam = ctx.camera(H, W, *r3d.REF_INTRINSICS)
        r3d.fuse_frames_device(ctx, cam, d_depth.ptr, np.uint8, F, d_pose.ptr, d_out.ptr, np.float32, depth_scale=scale)
        vs = V.VoxelSet(0.1, 2 * n, ctx)
        for dd in (1, 2):
            ctx.set_tuning("voxel_dedupe", dd)
            ts = []
            for _ in range(rounds):
                vs.clear()
                ctx.sync()
                ctx.timer_start()
                vs.insert_device(d_out.ptr, n)
                ts.append(ctx.timer_stop())
            st = vs.stats()
            med = np.median(ts)
            print("voxel insert [%s], %s: %.1f Mpts -> %d voxels, med %.3f ms, %.1f Gpts/s, %.1f GB/s at 12 B/pt"
                  % ("LDS dedupe" if dd == 2 else "direct    ", label, n / 1e6, st["voxels"], med, n / med / 1e6, n * 12 / med / 1e6))
        ctx.set_tuning("voxel_dedupe", 0)
        import time
        t0 = time.perf_counter()
        codes = vs.codes()
        t1 = time.perf_counter()
        data, nodes = V.format_bt(codes)
        t2 = time.perf_counter()
        print("   compact+D2H+sort %.1f ms, .bt build %.1f ms (%d nodes, %.1f MB)" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, nodes, len(data) / 1e6))
        vs.close()
        for b in (d_depth, d_pose, d_out):
            b.free()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--nn", type=int, default=100000)
    a = ap.parse_args()
    ctx = r3d.Context(0)
    if a.what in ("voxel", "all"):
        bench_voxel(ctx, a.frames, 384, 1280, a.rounds)
    if a.what in ("nn", "all"):
        bench_nn(ctx, a.nn, a.nn, 3)
    ctx.close()
