#!/usr/bin/env python3
"""The worst case of the voxel insert (BASELINE C2's cloud from random depth: 49.2 M points -> ~48.3 M voxels), REPS times
through one path -- the program to put behind `rocprofv3 --kernel-trace --stats --` (or `--pmc`) when looking at the
sort-merge insert's stages.  usage: voxel_sort_once.py [path: 1 cas | 2 sort-merge] [reps] [log2 of the table's slots]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
r3d = importlib.import_module("3d_reconstruction_system_amd")
V = importlib.import_module("3d_reconstruction_system_amd.voxelmap")
path = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = r3d.Context(0)
F, H, W = 100, 384, 1280
n = F * H * W
log2cap = int(sys.argv[3]) if len(sys.argv) > 3 else int(np.ceil(np.log2(2 * n)))
rng = np.random.default_rng(1234)
depth = rng.integers(1, 256, size=(F, H, W), dtype=np.uint8)
tab = r3d.pose_table(rng.normal(size=(F, 4)), rng.normal(size=(F, 3)))
d_depth, d_pose, d_xyz = ctx.alloc(n).upload(depth), ctx.alloc(tab.nbytes).upload(tab), ctx.alloc(n * 12)
cam = ctx.camera(H, W, *r3d.REF_INTRINSICS)
r3d.fuse_frames_device(ctx, cam, d_depth.ptr, np.uint8, F, d_pose.ptr, d_xyz.ptr, np.float32)
vs = V.VoxelSet(0.1, 1 << log2cap, ctx)
ctx.set_tuning("voxel_path", path)
times = []
for _ in range(reps + 1):
    vs.clear()
    ctx.sync()
    ctx.timer_start()
    vs.insert_device(d_xyz.ptr, n)
    times.append(ctx.timer_stop())
print("path %d, table 2^%d: %s ms, %d voxels" % (path, log2cap, " ".join("%.3f" % t for t in times[1:]), vs.stats()["voxels"]))
