#!/usr/bin/env python3
"""How the fused kernels' rate depends on the batch size (frames per launch): C2 geometry 1280x384, 100 ... 2000 frames,
u8 / u16 / f32 depth, f32 / f64 xyz, with and without pose.  Inputs are filled on the device."""
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
frames = [int(v) for v in sys.argv[1:]] or [100, 250, 500, 1000, 2000]
for F in frames:
    n = F * H * W
    cam = ctx.camera(H, W, *r3d.REF_INTRINSICS)
    tab = r3d.pose_table(rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10)
    d_pose = ctx.alloc(tab.nbytes).upload(tab)
    d_out = ctx.alloc(n * 24)
    for dt, db in ((np.uint8, 1), (np.uint16, 2), (np.float32, 4)):
        d_depth = ctx.alloc(n * db)
        L.check(ctx.lib.r3d_memset(ctx.handle, d_depth.ptr, 0x41, n * db))
        for loads in (0,):
          row = "%4d frames %-7s" % (F, np.dtype(dt).name)
          for odt, ob in ((np.float32, 12), (np.float64, 24)):
            def launch():
                r3d.fuse_frames_device(ctx, cam, d_depth.ptr, dt, F, d_pose.ptr, d_out.ptr, odt)
            reps = max(4, 4000 // F)
            for _ in range(reps):
                launch()
            ctx.sync()
            ts = []
            for _ in range(5):
                ctx.timer_start()
                for _ in range(reps):
                    launch()
                ts.append(ctx.timer_stop() / reps)
            ms = sorted(ts)[2]
            row += " | %s xyz %8.3f ms %5.2f TB/s" % (np.dtype(odt).name, ms, n * (db + ob) / ms / 1e9)
          print(row, flush=True)
        d_depth.free()
    d_pose.free()
    d_out.free()
ctx.close()
