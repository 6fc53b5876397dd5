#!/usr/bin/env python3
"""Sort-merge insert, same process, same cloud: the segmented first pass (voxel_path 2) against the dense one with its histogram
in front (voxel_path 3, round 5's first form), on BASELINE C2's worst-case cloud; then a cloud of surfaces forced down path 2
(its segments overflow: the fallback must give the CAS path's set).  usage: voxel_front_ab.py [reps]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
r3d = importlib.import_module("3d_reconstruction_system_amd")
V = importlib.import_module("3d_reconstruction_system_amd.voxelmap")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7
ctx = r3d.Context(0)
F, H, W = 100, 384, 1280
n = F * H * W
rng = np.random.default_rng(1234)
cam = ctx.camera(H, W, *r3d.REF_INTRINSICS)
tab = r3d.pose_table(rng.normal(size=(F, 4)), rng.normal(size=(F, 3)))
d_pose, d_xyz, d_depth = ctx.alloc(tab.nbytes).upload(tab), ctx.alloc(n * 12), ctx.alloc(n)


def cloud(depth):
    d_depth.upload(depth)
    r3d.fuse_frames_device(ctx, cam, d_depth.ptr, np.uint8, F, d_pose.ptr, d_xyz.ptr, np.float32)


def run(path, log2cap, label):
    vs = V.VoxelSet(0.1, 1 << log2cap, ctx)
    ctx.set_tuning("voxel_path", path)
    times = []
    for _ in range(reps + 1):
        vs.clear()
        ctx.sync()
        ctx.timer_start()
        vs.insert_device(d_xyz.ptr, n)
        times.append(ctx.timer_stop())
    st = vs.stats()
    codes = vs.codes()
    print("%-28s path %d: median %.3f ms (%s), %d voxels, ignored %d, fallbacks %d" % (
        label, path, float(np.median(times[1:])), " ".join("%.3f" % t for t in times[1:]), st["voxels"], st["ignored_points"],
        vs.sort_fallbacks()), flush=True)
    vs.close() if hasattr(vs, "close") else None
    return codes


cloud(rng.integers(1, 256, size=(F, H, W), dtype=np.uint8))
ref = None
for rnd in range(2):
    for path in (3, 2):
        c = run(path, 27, "C2 random depth")
        if ref is None:
            ref = c
        assert np.array_equal(c, ref), "paths disagree"
c1 = run(1, 27, "C2 random depth")
assert np.array_equal(c1, ref)
# surfaces: a slanted plane per frame -> tens of points per voxel
yy, xx = np.mgrid[0:H, 0:W]
plane = np.stack([np.clip(40 + (xx // 8 + yy // 6 + 3 * f) % 200, 1, 255) for f in range(F)]).astype(np.uint8)
cloud(plane)
a = run(1, 27, "planes")
b = run(2, 27, "planes (forced sort)")
c = run(3, 27, "planes (forced sort, dense)")
assert np.array_equal(a, b) and np.array_equal(a, c), "fallback disagrees"
print("OK")
