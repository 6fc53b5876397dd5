#!/usr/bin/env python3
"""Does the 256 MiB Infinity Cache flatter a benchmark that overwrites the SAME output buffer every launch?
C2 launches (100 frames, 639 MB of traffic) into 1 buffer vs rotating over K buffers (K x 590 MB >> 256 MiB), inputs
likewise; plus hipMemsetAsync over 0.59 GB (same buffer) and over 9.4 GB."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
r3d = importlib.import_module("3d_reconstruction_system_amd")
L = importlib.import_module("3d_reconstruction_system_amd._lib")
ctx = r3d.Context(0)
H, W, F = 384, 1280, 100
n = F * H * W
rng = np.random.default_rng(0)
cam = ctx.camera(H, W, *r3d.REF_INTRINSICS)
tab = r3d.pose_table(rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10)
d_pose = ctx.alloc(tab.nbytes).upload(tab)
K = 16
outs = [ctx.alloc(n * 12) for _ in range(K)]
ins = [ctx.alloc(n) for _ in range(K)]
for b in ins:
    L.check(ctx.lib.r3d_memset(ctx.handle, b.ptr, 0x41, n))


def run(k_out, k_in, reps=160):
    def launch(i):
        r3d.fuse_frames_device(ctx, cam, ins[i % k_in].ptr, np.uint8, F, d_pose.ptr, outs[i % k_out].ptr, np.float32)
    for i in range(64):
        launch(i)
    ctx.sync()
    ts = []
    for g in range(5):
        ctx.timer_start()
        for i in range(reps):
            launch(i)
        ts.append(ctx.timer_stop() / reps)
    return sorted(ts)[2]


for k_out, k_in in ((1, 1), (16, 1), (1, 16), (16, 16)):
    ms = run(k_out, k_in)
    print("fused u8->f32, C2 launch, %2d output buffer(s), %2d input buffer(s): %.4f ms = %.2f TB/s"
          % (k_out, k_in, ms, n * 13 / ms / 1e9), flush=True)

big = ctx.alloc(16 * n * 12)
for nbytes, label in ((n * 12, "0.59 GB, same buffer"), (16 * n * 12, "9.4 GB")):
    for _ in range(3):
        L.check(ctx.lib.r3d_memset(ctx.handle, big.ptr, 1, nbytes))
    ctx.sync()
    ts = []
    reps = 40 if nbytes < 1e9 else 4
    for g in range(5):
        ctx.timer_start()
        for _ in range(reps):
            L.check(ctx.lib.r3d_memset(ctx.handle, big.ptr, 1, nbytes))
        ts.append(ctx.timer_stop() / reps)
    ms = sorted(ts)[2]
    print("hipMemsetAsync %s: %.4f ms = %.2f TB/s" % (label, ms, nbytes / ms / 1e9), flush=True)
ctx.close()
