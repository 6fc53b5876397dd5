#!/usr/bin/env python3
"""End to end for an RGBD sequence (config 5's flow on files): depth PNGs + colour PNGs + pose file in, one coloured fused
ASCII PLY (genply_noRGB's row layout, pixel_to_camera.py:71-87) out.  Stages timed separately: native threaded PNG decode
of both image sets, the pipelined host -> GPU -> host fuse carrying the colour, the native coloured-PLY writer."""
import importlib
import os
import shutil
import sys
import tempfile
import time

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
r3d = importlib.import_module("3d_reconstruction_system_amd")

F, H, W = int(os.environ.get("FRAMES", "50")), 384, 1280
td = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
rng = np.random.default_rng(1234)
dp, cp = [], []
for k in range(F):
    base = np.add.outer(np.arange(H), np.arange(W)) / 37.0 + k
    depth = np.clip(40 + 30 * np.sin(base) + rng.integers(0, 6, (H, W)), 1, 255).astype(np.uint8)
    col = np.stack([np.clip(128 + 100 * np.sin(base + c) + rng.integers(0, 9, (H, W)), 0, 255) for c in range(3)], -1).astype(np.uint8)
    dp.append(os.path.join(td, "d%04d.png" % k))
    cp.append(os.path.join(td, "c%04d.png" % k))
    Image.fromarray(depth, "L").save(dp[-1])
    Image.fromarray(col, "RGB").save(cp[-1])
q, t = rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10
ctx = r3d.Context(0)
r3d.fuse_frames_rgb(np.ones((1, 8, 8), np.uint8), np.ones((1, 8, 8, 3), np.uint8), q[:1], t[:1], ctx=ctx)   # warm-up
n = F * H * W
t0 = time.perf_counter()
depth = r3d.cloud_io.read_depth_batch(dp)
rgb = r3d.cloud_io.read_rgb_batch(cp)
t1 = time.perf_counter()
xyz, rgba = r3d.fuse_frames_rgb(depth, rgb, q, t, out_dtype=np.float64, ctx=ctx)
t2 = time.perf_counter()
out = os.path.join(td, "fused_rgb.ply")
r3d.cloud_io.write_ply_rgb(out, xyz, rgba)
t3 = time.perf_counter()
print("%d RGBD frames of %dx%d = %.1f Mpoints: PNG decode (depth + colour) %.0f ms | fuse with colour, host to host, f64 xyz %.0f ms | "
      "coloured PLY (%.2f GB) %.0f ms | total %.2f s = %.1f Mpoints/s"
      % (F, W, H, n / 1e6, (t1 - t0) * 1e3, (t2 - t1) * 1e3, os.path.getsize(out) / 1e9, (t3 - t2) * 1e3, t3 - t0, n / (t3 - t0) / 1e6))
head = open(out, "rb").read(400).decode()
assert "property uchar alpha" in head and head.startswith("ply\n    format ascii 1.0")
ctx.close()
shutil.rmtree(td)
