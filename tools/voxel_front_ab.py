#!/usr/bin/env python3
"""The sort-merge insert (voxel_path 2) against the compare-and-swap path (1), same process, same clouds: BASELINE C2's worst
case (random depth), planes under random poses, and a cloud whose keys crowd into a handful of voxels, forced down path 2 (a
first-pass segment is full: what it cannot take goes in through the deferred list).  usage: voxel_front_ab.py [reps]"""
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
    print("%-28s path %d: median %.3f ms (%s), %d voxels, ignored %d" % (
        label, path, float(np.median(times[1:])), " ".join("%.3f" % t for t in times[1:]), st["voxels"], st["ignored_points"]), flush=True)
    vs.close() if hasattr(vs, "close") else None
    return codes


cloud(rng.integers(1, 256, size=(F, H, W), dtype=np.uint8))
ref = run(2, 27, "C2 random depth")
assert np.array_equal(run(2, 27, "C2 random depth"), ref)
assert np.array_equal(run(1, 27, "C2 random depth"), ref)
yy, xx = np.mgrid[0:H, 0:W]
plane = np.stack([np.clip(40 + (xx // 8 + yy // 6 + 3 * f) % 200, 1, 255) for f in range(F)]).astype(np.uint8)
cloud(plane)
assert np.array_equal(run(1, 27, "planes"), run(2, 27, "planes (forced sort)"))
# eight voxels, alternating from point to point (the neighbour-lane test removes nothing): every key in eight segments' worth
pts = np.zeros((n, 3), np.float32)
pts[:, 0] = (np.arange(n) % 8) * 0.1 + 0.05
d_xyz.upload(pts)
a = run(1, 27, "eight voxels")
b = run(2, 27, "eight voxels (forced sort)")
assert np.array_equal(a, b) and a.shape[0] == 8
# C2's cloud with a fifth of the pixels at one point per frame (no depth), scattered
depth = rng.integers(1, 256, size=(F, H, W), dtype=np.uint8)
depth[rng.random((F, H, W)) < 0.2] = 0
cloud(depth)
ctx.set_tuning("voxel_path", 0)
a = run(0, 27, "C2, 20 % without depth")
assert ctx.get_tuning("voxel_last_path") == 2
assert np.array_equal(a, run(1, 27, "C2, 20 % without depth"))
print("OK")
