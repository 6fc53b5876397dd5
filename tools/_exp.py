import importlib, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
r3d = importlib.import_module("3d_reconstruction_system_amd")
V = importlib.import_module("3d_reconstruction_system_amd.voxelmap")
ctx = r3d.Context(0)
F, H, W = 100, 384, 1280
n = F * H * W
rng = np.random.default_rng(1234)
depth = rng.integers(1, 256, size=(F, H, W), dtype=np.uint8)
tab = r3d.pose_table(rng.normal(size=(F, 4)), rng.normal(size=(F, 3)))
d_depth, d_pose, d_xyz = ctx.alloc(n).upload(depth), ctx.alloc(tab.nbytes).upload(tab), ctx.alloc(n * 12)
cam = ctx.camera(H, W, *r3d.REF_INTRINSICS)
r3d.fuse_frames_device(ctx, cam, d_depth.ptr, np.uint8, F, d_pose.ptr, d_xyz.ptr, np.float32)
vs = V.VoxelSet(0.1, 1 << 27, ctx)
ctx.set_tuning("voxel_path", 2)
for exp in (0, 6, 1, 0):
    ctx.set_tuning("voxel_dedupe", 10 + exp if exp else 0)
    for _ in range(6):
        vs.clear(); ctx.sync(); vs.insert_device(d_xyz.ptr, n); ctx.sync()
