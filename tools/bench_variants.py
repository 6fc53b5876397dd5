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


def bench_fuse(ctx, F, H, W, rounds, iters, odt=np.float32):
    rng = np.random.default_rng(1234)
    n = F * H * W
    osz = np.dtype(odt).itemsize
    depth = rng.integers(1, 256, size=(F, H, W), dtype=np.uint8)
    table = r3d.pose_table(rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10)
    d_depth = ctx.alloc(n).upload(depth)
    d_pose = ctx.alloc(table.nbytes).upload(table)
    d_out = ctx.alloc(n * 3 * osz)
    cam = ctx.camera(H, W, *r3d.REF_INTRINSICS)
    configs = []
    for variant in (3, 5, 6, 7):
        for nt in (3,):
            for blocks in (2048, 4096):

                configs.append((variant, nt, blocks))
    results = {c: [] for c in configs}

    def run():
        r3d.fuse_frames_device(ctx, cam, d_depth.ptr, np.uint8, F, d_pose.ptr, d_out.ptr, odt)

    for _ in range(rounds):
        for c in configs:
            ctx.set_tuning("fuse_variant", c[0])
            ctx.set_tuning("nontemporal", c[1])
            ctx.set_tuning("fuse_blocks", c[2])
            results[c].append(time_ms(ctx, run, iters))
    bpp = 1 + 3 * osz
    print("fuse %dx%dx%d = %.1f Mpts, %d B/pt = %.1f MB per launch (%s out)" % (F, H, W, n / 1e6, bpp, n * bpp / 1e6, np.dtype(odt).name))
    print("variant nt blocks   med_ms   min_ms   GB/s(med)  Gpts/s")
    for c in configs:
        med, mn = np.median(results[c]), np.min(results[c])
        print("%7d %2d %6d %8.4f %8.4f %10.1f %7.1f" % (c[0], c[1], c[2], med, mn, n * bpp / med / 1e6, n / med / 1e6))
    for k in ("fuse_variant", "nontemporal", "fuse_blocks"):
        ctx.set_tuning(k, 0)


def bench_fuse_dtypes(ctx, rounds, iters):
    """Default kernel on every depth/output type and on the 1080p f32 shape of BASELINE config 5."""
    rng = np.random.default_rng(5)
    print("fuse, default variant: depth dtype x out dtype")
    for (F, H, W) in ((100, 384, 1280), (24, 1080, 1920)):
        n = F * H * W
        table = r3d.pose_table(rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10)
        d_pose = ctx.alloc(table.nbytes).upload(table)
        cam = ctx.camera(H, W, *r3d.REF_INTRINSICS)
        for ddt in (np.uint8, np.uint16, np.float32):
            if ddt == np.float32:
                depth = (rng.random((F, H, W)) * 99.5 + 0.5).astype(np.float32)
            else:
                depth = rng.integers(1, np.iinfo(ddt).max, size=(F, H, W), dtype=ddt)
            d_depth = ctx.alloc(depth.nbytes).upload(depth)
            for odt in (np.float32, np.float64):
                d_out = ctx.alloc(n * 3 * np.dtype(odt).itemsize)
                run = lambda: r3d.fuse_frames_device(ctx, cam, d_depth.ptr, ddt, F, d_pose.ptr, d_out.ptr, odt)
                ts = [time_ms(ctx, run, iters) for _ in range(rounds)]
                bpp = np.dtype(ddt).itemsize + 3 * np.dtype(odt).itemsize
                med = np.median(ts)
                print("%dx%dx%d %-7s -> %-7s %2d B/pt  med %.4f ms  %.1f GB/s  %.1f Gpts/s"
                      % (F, H, W, np.dtype(ddt).name, np.dtype(odt).name, bpp, med, n * bpp / med / 1e6, n / med / 1e6))
                d_out.free()
            d_depth.free()
        d_pose.free()


def bench_apply(ctx, n, rounds, iters):
    rng = np.random.default_rng(1)
    p = (rng.normal(size=(n, 3)) * 50).astype(np.float32)
    d_in = ctx.alloc(p.nbytes).upload(p)
    d_out = ctx.alloc(p.nbytes)
    T = np.eye(4)
    T[:3, 3] = (1, 2, 3)
    res = {}
    for variant in (0, 1):
        for blocks in (1024, 2048, 4096, 16384, 1000000):
            res[(variant, blocks)] = []
    for _ in range(rounds):
        for (variant, blocks) in res:
            ctx.set_tuning("apply_variant", variant)
            ctx.set_tuning("apply_blocks", blocks)
            res[(variant, blocks)].append(time_ms(ctx, lambda: r3d.apply_T_device(ctx, d_in.ptr, np.float32, n, T, d_out.ptr,
                                                                       np.float32), iters))
    ctx.set_tuning("apply_blocks", 0)
    ctx.set_tuning("apply_variant", 0)
    print("apply_T %.1f Mpts, 24 B/pt" % (n / 1e6))
    for (variant, blocks), v in res.items():
        med = np.median(v)
        print("variant %d blocks %7d  med %.4f ms  %.1f GB/s" % (variant, blocks, med, n * 24 / med / 1e6))


def bench_nn(ctx, n, m, rounds):
    import ctypes as C
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
        for blocks in (1, 2048, 4096, 8192):
            ctx.set_tuning("nn_variant", S)
            ctx.set_tuning("nn_blocks", blocks)
            ts = []
            for _ in range(rounds):
                ts.append(time_ms(ctx, lambda: L.check(ctx.lib.r3d_icp_nn(ctx.handle, d_src.ptr, n, d_tgt.ptr, m,
                                                                           d_idx.ptr, d_d2.ptr)), 1))
            med = np.median(ts)
            print("S=%d blocks>=%5d  med %.3f ms  %.2f Tpairs/s  (%.1f TFLOP/s at 8 flop/pair)"
                  % (S, blocks, med, n * m / med / 1e9, n * m * 8 / med / 1e9))
    ctx.set_tuning("nn_variant", 0)
    ctx.set_tuning("nn_blocks", 0)
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
    if a.what in ("fuse", "all"):
        bench_fuse(ctx, a.frames, 384, 1280, a.rounds, a.iters)
    if a.what in ("fuse64", "all"):
        bench_fuse(ctx, a.frames, 384, 1280, a.rounds, a.iters, np.float64)
    if a.what in ("dtypes", "all"):
        bench_fuse_dtypes(ctx, a.rounds, a.iters)
    if a.what in ("apply", "all"):
        bench_apply(ctx, 50_000_000, a.rounds, a.iters)
    if a.what in ("voxel", "all"):
        bench_voxel(ctx, a.frames, 384, 1280, a.rounds)
    if a.what in ("nn", "all"):
        bench_nn(ctx, a.nn, a.nn, 3)
    ctx.close()
