#!/usr/bin/env python3
"""Where the wall time of the camera_to_world drop-in goes (C2: 100 frames of 1280x384 -> 1.26 GB ASCII PLY)."""
import importlib
import os
import sys
import tempfile
import time

t_start = time.perf_counter()
import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t0 = time.perf_counter()
r3d = importlib.import_module("3d_reconstruction_system_amd")
t_import = time.perf_counter() - t0
F, H, W = 100, 384, 1280
rng = np.random.default_rng(0)
td = tempfile.mkdtemp(dir="/dev/shm")
from PIL import Image  # noqa: E402
paths = []
for k in range(F):
    p = os.path.join(td, "%03d.png" % k)
    Image.fromarray(rng.integers(1, 256, (H, W), dtype=np.uint8), "L").save(p, compress_level=1)
    paths.append(p)
q, t = rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10
t0 = time.perf_counter()
ctx = r3d.Context(0)
t_ctx = time.perf_counter() - t0
t0 = time.perf_counter()
depths = r3d.cloud_io.read_depth_batch(paths)
t_png = time.perf_counter() - t0
t0 = time.perf_counter()
world = r3d.fuse_frames(depths, q, t, out_dtype=np.float64, ctx=ctx)
t_fuse = time.perf_counter() - t0
t0 = time.perf_counter()
world2 = r3d.fuse_frames(depths, q, t, out_dtype=np.float64, ctx=ctx, out=world)
t_fuse2 = time.perf_counter() - t0
out = os.path.join(td, "o.ply")
t0 = time.perf_counter()
r3d.cloud_io.write_ply(out, world)
t_ply = time.perf_counter() - t0
lib = r3d.load_library()
import ctypes as C  # noqa: E402
nb = C.c_size_t()
t0 = time.perf_counter()
r3d.cloud_io.format_ply(world)
t_fmt = time.perf_counter() - t0
print("numpy import %.3f | package import + dlopen %.3f | HIP context %.3f | PNG decode %.3f | fuse f64 host->host fresh array %.3f "
      "(reused array %.3f) | write_ply %.3f (format only, to memory: %.3f) | PLY %.2f GB"
      % (t0 - t0 + (t_start and 0) + 0, t_import, t_ctx, t_png, t_fuse, t_fuse2, t_ply, t_fmt, os.path.getsize(out) / 1e9))
import shutil  # noqa: E402
shutil.rmtree(td)
