"""The f2 export chain on the GPU box: insert -> ascending Morton codes in host memory -> OctoMap .bt, timed stage by stage.
usage: python tools/voxel_export_once.py [frames]   (C2-like random cloud: ~1 voxel per point, the worst case for every stage)"""
import importlib
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
R = importlib.import_module("3d_reconstruction_system_amd")
V = importlib.import_module("3d_reconstruction_system_amd.voxelmap")
import ctypes as C
L = importlib.import_module("3d_reconstruction_system_amd._lib")

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 100
H, W = 384, 1280
ctx = R.Context(0)
rng = np.random.default_rng(1234)
depth = rng.integers(1, 256, (frames, H, W), dtype=np.uint8)
q = rng.normal(size=(frames, 4))
t = rng.normal(size=(frames, 3)) * 10
world = R.fuse_frames(depth, q, t, out_dtype=np.float32, ctx=ctx)
n = world.shape[0]
for rep in range(3):
    vs = V.VoxelSet(0.1, 1 << 27, ctx)
    t0 = time.perf_counter(); vs.insert(world); st = vs.stats(); t1 = time.perf_counter()
    codes = vs.codes(); t2 = time.perf_counter()
    vs.close()
    td = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    p = os.path.join(td, "m.bt")
    nodes = C.c_int64()
    t3 = time.perf_counter()
    L.check(ctx.lib.r3d_octree_write_bt(os.fsencode(p), codes.ctypes.data, codes.shape[0], C.c_double(0.1), C.byref(nodes)))
    t4 = time.perf_counter()
    size = os.path.getsize(p)
    os.remove(p); os.rmdir(td)
    print("%d points -> %d voxels: insert from host memory (H2D incl.) %.1f ms | codes() to host, sorted %.1f ms | .bt (%d nodes, %.1f MB) %.0f ms"
          % (n, codes.shape[0], (t1 - t0) * 1e3, (t2 - t1) * 1e3, nodes.value, size / 1e6, (t4 - t3) * 1e3))
