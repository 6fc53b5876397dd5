"""Config 5's ingestion on this machine: N 1080p JPEG files (AirSim's format) -> grey rasters / colour planes through the
native batch decoders, beside a PIL loop.  usage: python tools/bench_jpeg_decode.py [n_files]"""
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
R = importlib.import_module("3d_reconstruction_system_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H, W = 1080, 1920
td = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:H, 0:W]
paths = []
for k in range(8):                                   # eight distinct images, reused round robin
    img = (np.stack([128 + 100 * np.sin((xx + 31 * k) / 47.0 + yy / 29.0), 128 + 90 * np.cos(xx / 13.0 + k), 100 + (yy + 7 * k) % 97], 2)
           + rng.normal(0, 4, (H, W, 3))).clip(0, 255).astype(np.uint8)
    p = os.path.join(td, "src%d.jpg" % k)
    Image.fromarray(img, "RGB").save(p, quality=90)
    paths.append(p)
files = [paths[k % 8] for k in range(n)]
print("%d files of %dx%d, %.0f KB each" % (n, W, H, os.path.getsize(paths[0]) / 1e3))
grey = np.empty((n, H, W), np.uint8)
rgb = np.empty((n, H, W, 3), np.uint8)
grey[:] = 0
rgb[:] = 0
for rep in range(2):
    t = time.perf_counter(); R.cloud_io.read_depth_batch(files, out=grey); t1 = time.perf_counter() - t
    t = time.perf_counter(); R.cloud_io.read_rgb_batch(files, out=rgb); t2 = time.perf_counter() - t
    print("native: grey %.0f ms (%.2f ms per file), colour %.0f ms (%.2f ms per file)" % (t1 * 1e3, t1 / n * 1e3, t2 * 1e3, t2 / n * 1e3))
t = time.perf_counter()
for p in files[:16]:
    np.asarray(Image.open(p).convert("RGB"))
t3 = (time.perf_counter() - t) / 16
print("PIL loop (libjpeg-turbo, one thread): %.1f ms per file" % (t3 * 1e3))
shutil.rmtree(td, ignore_errors=True)
