#!/usr/bin/env python3
"""r3d_fuse_frames_voxel on a duplicate-heavy batch (300 frames of 1080p f32 depth + RGB, fronto-parallel planes = 622 M points):
the program to put behind `rocprofv3 --pmc ... --` when looking for what bounds fuse_voxel_kernel.  Prints the one-launch time
next to the two calls'.  argv[1] = fuse_blocks (0 = library default)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
r3d = importlib.import_module("3d_reconstruction_system_amd")
L = importlib.import_module("3d_reconstruction_system_amd._lib")
V = importlib.import_module("3d_reconstruction_system_amd.voxelmap")
ctx = r3d.Context(0)
F, H, W = 300, 1080, 1920
n = F * H * W
d_depth, d_rgb = ctx.alloc(n * 4), ctx.alloc(n * 3)
L.check(ctx.lib.r3d_memset(ctx.handle, d_depth.ptr, 0x41, n * 4))
L.check(ctx.lib.r3d_memset(ctx.handle, d_rgb.ptr, 0x5a, n * 3))
rng = np.random.default_rng(5)
tab = r3d.pose_table(rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10)
d_pose = ctx.alloc(tab.nbytes).upload(tab)
cam = ctx.camera(H, W, 960.0, 960.0, 959.5, 539.5)
d_xyz, d_rgba = ctx.alloc(n * 12), ctx.alloc(n * 4)
vs = V.VoxelSet(0.1, 1 << 26, ctx)
if len(sys.argv) > 1:
    ctx.set_tuning("fuse_blocks", int(sys.argv[1]))
one = []
for _ in range(3):
    vs.clear()
    ctx.sync()
    ctx.timer_start()
    r3d.fuse_frames_voxel_device(ctx, cam, d_depth.ptr, np.float32, F, d_pose.ptr, d_rgb.ptr, d_xyz.ptr, d_rgba.ptr, vs)
    one.append(ctx.timer_stop())
st1 = vs.stats()
ctx.set_tuning("fuse_blocks", 0)
two = []
for _ in range(3):
    vs.clear()
    ctx.sync()
    ctx.timer_start()
    r3d.fuse_frames_rgb_device(ctx, cam, d_depth.ptr, np.float32, F, d_pose.ptr, d_rgb.ptr, d_xyz.ptr, np.float32, d_rgba.ptr)
    vs.insert_device(d_xyz.ptr, n)
    two.append(ctx.timer_stop())
st2 = vs.stats()
print("one launch %.2f ms = %.1f Gpoints/s; two calls %.2f ms; %d voxels%s"
      % (min(one), n / min(one) / 1e6, min(two), st1["voxels"], "" if st1 == st2 else "  MISMATCH " + str(st2)))
