#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (r3d_fuse_frames_host) on config C2."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
r3d = importlib.import_module("3d_reconstruction_system_amd")

F, H, W = 100, 384, 1280
rng = np.random.default_rng(1234)
depth = rng.integers(1, 256, size=(F, H, W), dtype=np.uint8)
q = rng.normal(size=(F, 4))
t = rng.normal(size=(F, 3)) * 10
ctx = r3d.Context(0)
t0 = time.perf_counter(); tab = r3d.pose_table(q, t); print("pose_table (host, %d frames): %.2f ms" % (F, (time.perf_counter() - t0) * 1e3))
for dt in (np.float32, np.float64):
    for rep in range(4):
        t0 = time.perf_counter()
        out = r3d.fuse_frames(depth, q, t, out_dtype=dt, ctx=ctx)
        dt_s = time.perf_counter() - t0
        print("%s rep %d: %.1f ms  %.2f Gpoints/s  (D2H-equivalent %.1f GB/s)" % (np.dtype(dt).name, rep, dt_s * 1e3,
              F * H * W / dt_s / 1e9, out.nbytes / dt_s / 1e9))
    pre = np.zeros((F * H * W, 3), dt)
    for rep in range(3):
        t0 = time.perf_counter()
        r3d.fuse_frames(depth, q, t, out_dtype=dt, ctx=ctx, out=pre)
        dt_s = time.perf_counter() - t0
        print("%s pageable pre-touched out rep %d: %.1f ms  %.2f Gpoints/s  (%.1f GB/s)" % (np.dtype(dt).name, rep,
              dt_s * 1e3, F * H * W / dt_s / 1e9, pre.nbytes / dt_s / 1e9))
    if hasattr(ctx, "pinned_empty"):
        pin = ctx.pinned_empty((F * H * W, 3), dt)
        for rep in range(3):
            t0 = time.perf_counter()
            r3d.fuse_frames(depth, q, t, out_dtype=dt, ctx=ctx, out=pin)
            dt_s = time.perf_counter() - t0
            print("%s pinned out rep %d: %.1f ms  %.2f Gpoints/s  (%.1f GB/s)" % (np.dtype(dt).name, rep, dt_s * 1e3,
                  F * H * W / dt_s / 1e9, pin.nbytes / dt_s / 1e9))

# apply-T through the host pipeline: 12 B/point each way, full duplex
ctx = r3d.Context(0)
pts = (rng.normal(size=(F * H * W, 3)) * 50).astype(np.float32)
outp = np.zeros_like(pts)
T = np.eye(4); T[:3, 3] = (1, 2, 3)
for rep in range(3):
    t0 = time.perf_counter()
    r3d.apply_T(pts, T, ctx=ctx) if rep == 0 else None
    dt_s = time.perf_counter() - t0
pin_i, pin_o = ctx.pinned_empty(pts.shape, np.float32), ctx.pinned_empty(pts.shape, np.float32)
pin_i[...] = pts
L = importlib.import_module("3d_reconstruction_system_amd._lib")
for label, a, b in (("pageable (pre-touched)", pts, outp), ("pinned", pin_i, pin_o)):
    for rep in range(3):
        t0 = time.perf_counter()
        L.check(ctx.lib.r3d_apply_T_host(ctx.handle, a.ctypes.data, 0, a.shape[0], np.ascontiguousarray(T).ctypes.data, b.ctypes.data, 0))
        dt_s = time.perf_counter() - t0
    print("apply_T host %s: %.1f ms  %.2f Gpoints/s  (%.1f GB/s each way)" % (label, dt_s * 1e3, a.shape[0] / dt_s / 1e9, a.nbytes / dt_s / 1e9))
# RGBD batch from host memory (config 5's flow on real data: decoded PNGs in, coloured cloud out)
F, H, W = 100, 384, 1280
rng = np.random.default_rng(3)
d = rng.integers(1, 256, size=(F, H, W), dtype=np.uint8)
rgb = rng.integers(0, 256, size=(F, H, W, 3), dtype=np.uint8)
q, t = rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10
tab = r3d.pose_table(q, t)
n = F * H * W
cam = ctx.camera(H, W, *r3d.REF_INTRINSICS)
for label, alloc in (("pageable (pre-touched)", lambda shape, dt: np.zeros(shape, dt)), ("pinned", ctx.pinned_empty)):
    xyz, rgba = alloc((n, 3), np.float32), alloc((n,), np.uint32)
    for rep in range(3):
        t0 = time.perf_counter()
        L.check(ctx.lib.r3d_fuse_frames_rgb_host(ctx.handle, cam.handle, d.ctypes.data, 0, F, 1.0, tab.ctypes.data, rgb.ctypes.data,
                                                 xyz.ctypes.data, 0, rgba.ctypes.data))
        dt_s = time.perf_counter() - t0
    print("fuse_frames_rgb host, outputs %s: %.1f ms  %.2f Gpoints/s  (%.1f GB/s out, %.1f GB/s in)"
          % (label, dt_s * 1e3, n / dt_s / 1e9, n * 16 / dt_s / 1e9, n * 4 / dt_s / 1e9))
ctx.close()
