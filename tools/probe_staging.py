#!/usr/bin/env python3
"""Input staging through the Infinity Cache (fuse_prefetch / fuse_chunk_mb knobs): C2 geometry, big batches, every depth
type, f32 xyz (+ f64, + colour), prefetch off vs on at several chunk sizes.  Inputs filled on the device."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
r3d = importlib.import_module("3d_reconstruction_system_amd")
L = importlib.import_module("3d_reconstruction_system_amd._lib")
ctx = r3d.Context(0)
H, W = 384, 1280
rng = np.random.default_rng(0)
frames = [int(v) for v in sys.argv[1:]] or [1000]


def timed(launch, reps):
    for _ in range(reps):
        launch()
    ctx.sync()
    ts = []
    for _ in range(5):
        ctx.timer_start()
        for _ in range(reps):
            launch()
        ts.append(ctx.timer_stop() / reps)
    return sorted(ts)[2]


for F in frames:
    n = F * H * W
    cam = ctx.camera(H, W, *r3d.REF_INTRINSICS)
    tab = r3d.pose_table(rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10)
    d_pose = ctx.alloc(tab.nbytes).upload(tab)
    d_out = ctx.alloc(n * 24)
    d_rgb, d_rgba = ctx.alloc(n * 3), ctx.alloc(n * 4)
    L.check(ctx.lib.r3d_memset(ctx.handle, d_rgb.ptr, 0x5a, n * 3))
    reps = max(3, 3000 // F)
    for dt, db in ((np.uint8, 1), (np.uint16, 2), (np.float32, 4)):
        d_depth = ctx.alloc(n * db)
        L.check(ctx.lib.r3d_memset(ctx.handle, d_depth.ptr, 0x41, n * db))
        for what, bpp, fn in (
                ("f32 xyz", db + 12, lambda: r3d.fuse_frames_device(ctx, cam, d_depth.ptr, dt, F, d_pose.ptr, d_out.ptr, np.float32)),
                ("f64 xyz", db + 24, lambda: r3d.fuse_frames_device(ctx, cam, d_depth.ptr, dt, F, d_pose.ptr, d_out.ptr, np.float64)),
                ("f32 xyz + colour", db + 19, lambda: r3d.fuse_frames_rgb_device(ctx, cam, d_depth.ptr, dt, F, d_pose.ptr, d_rgb.ptr, d_out.ptr, np.float32, d_rgba.ptr))):
            row = "%4d frames %-7s %-16s" % (F, np.dtype(dt).name, what)
            ctx.set_tuning("fuse_prefetch", 1)
            ms = timed(fn, reps)
            row += " | off %7.3f ms %5.2f TB/s" % (ms, n * bpp / ms / 1e9)
            ctx.set_tuning("fuse_prefetch", 2)
            for mb in (64, 96, 128):
                ctx.set_tuning("fuse_chunk_mb", mb)
                ms = timed(fn, reps)
                row += " | %3d MB %5.2f" % (mb, n * bpp / ms / 1e9)
            ctx.set_tuning("fuse_prefetch", 0)
            ctx.set_tuning("fuse_chunk_mb", 0)
            ms = timed(fn, reps)
            row += " | AUTO %5.2f" % (n * bpp / ms / 1e9)
            print(row, flush=True)
        d_depth.free()
    for b in (d_pose, d_out, d_rgb, d_rgba):
        b.free()
ctx.close()
