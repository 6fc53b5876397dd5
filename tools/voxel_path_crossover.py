#!/usr/bin/env python3
"""Where does the sort-merge insert stop paying?  A cloud of surfaces (100 frames of slanted planes, 49 M points) voxelised at
growing voxel sizes -- i.e. growing numbers of points per voxel, neighbours in memory sharing voxels as in a scan -- through the
CAS path (1), the sort-merge path (2) and the library's own choice (0); with `room`, 26 views of the inside of a box room
(synthetic.room_views, 20 M points) at shrinking voxel sizes instead.  usage: voxel_path_crossover.py [reps] [room]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
r3d = importlib.import_module("3d_reconstruction_system_amd")
V = importlib.import_module("3d_reconstruction_system_amd.voxelmap")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ctx = r3d.Context(0)
room = len(sys.argv) > 2 and sys.argv[2] == "room"
rng = np.random.default_rng(7)
if room:
    F, H, W = 26, 768, 1024
    depth, q, t, K = importlib.import_module("3d_reconstruction_system_amd.synthetic").room_views(F, H, W, 3)
    tab, sizes = r3d.pose_table(q, t), (0.1, 0.05, 0.03, 0.02, 0.01, 0.005)
else:
    F, H, W = 100, 384, 1280
    K = r3d.REF_INTRINSICS
    tab = r3d.pose_table(rng.normal(size=(F, 4)), rng.normal(size=(F, 3)))
    yy, xx = np.mgrid[0:H, 0:W]
    depth = np.stack([np.clip(40 + (xx // 8 + yy // 6 + 3 * f) % 200, 1, 255) for f in range(F)]).astype(np.uint8)
    sizes = (0.1, 0.2, 0.3, 0.5, 0.8, 1.2, 2.0)
n = F * H * W
cam = ctx.camera(H, W, *K)
d_pose, d_xyz, d_depth = ctx.alloc(tab.nbytes).upload(tab), ctx.alloc(n * 12), ctx.alloc(depth.nbytes).upload(depth)
r3d.fuse_frames_device(ctx, cam, d_depth.ptr, depth.dtype.type, F, d_pose.ptr, d_xyz.ptr, np.float32)
log2cap = int(np.ceil(np.log2(2 * n)))
for res in sizes:
    out = []
    for path in (1, 2, 0):
        vs = V.VoxelSet(res, 1 << log2cap, ctx)
        ctx.set_tuning("voxel_path", path)
        ts = []
        for _ in range(reps + 1):
            vs.clear()
            ctx.sync()
            ctx.timer_start()
            vs.insert_device(d_xyz.ptr, n)
            ts.append(ctx.timer_stop())
        out.append((float(np.median(ts[1:])), ctx.get_tuning("voxel_last_path"), vs.stats()["voxels"]))
        vs.close()
    print("res %.3f: %5.1f points per voxel | CAS %.3f ms | sort-merge %.3f ms | auto %.3f ms (took path %d)" % (
        res, n / max(out[0][2], 1), out[0][0], out[1][0], out[2][0], out[2][1]), flush=True)
ctx.set_tuning("voxel_path", 0)
