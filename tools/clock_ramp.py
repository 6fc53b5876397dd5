#!/usr/bin/env python3
"""Does the fused launch get faster when the GPU has been busy for seconds rather than tens of milliseconds?  Prints the mean
launch time of consecutive groups of 500 launches over ~4 s, for the cached raster (bench loop) and for rotating rasters."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
r3d = importlib.import_module("3d_reconstruction_system_amd")
ctx = r3d.Context(0)
rng = np.random.default_rng(0)
F, H, W = 100, 384, 1280
n = F * H * W
cam = ctx.camera(H, W, *r3d.REF_INTRINSICS)
depth = rng.integers(1, 256, size=(F, H, W), dtype=np.uint8)
tab = r3d.pose_table(rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10)
d_pose, d_xyz = ctx.alloc(tab.nbytes).upload(tab), ctx.alloc(n * 12)
copies = [ctx.alloc(n).upload(depth) for _ in range(16)]
for label, knob, rot in (("cached raster, staging off", 1, False), ("cached raster, default", 0, False), ("rotating rasters, default", 0, True)):
    ctx.set_tuning("fuse_prefetch", knob)
    time.sleep(1.0)                       # start from an idle GPU
    out, k, t_begin = [], 0, time.perf_counter()
    while time.perf_counter() - t_begin < 4.0:
        ctx.timer_start()
        for _ in range(500):
            d = copies[k % 16] if rot else copies[0]
            k += 1
            r3d.fuse_frames_device(ctx, cam, d.ptr, np.uint8, F, d_pose.ptr, d_xyz.ptr, np.float32)
        out.append(ctx.timer_stop() / 500)
    sel = [0, 1, 2, 4, 8, 16, 32, len(out) // 2, len(out) - 1]
    print("%-28s groups of 500 launches: %s  (n=%d; TB/s of the last: %.2f)"
          % (label, " ".join("%d:%.1fus" % (i, out[i] * 1e3) for i in sel if i < len(out)), len(out), n * 13 / out[-1] / 1e9))
ctx.close()
