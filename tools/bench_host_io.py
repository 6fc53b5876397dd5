#!/usr/bin/env python3
"""Host-side I/O rates of the library on this machine: PNG batch decode, PLY / txt formatting."""
import importlib
import os
import sys
import tempfile
import time

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
R = importlib.import_module("3d_reconstruction_system_amd")

rng = np.random.default_rng(0)
td = tempfile.mkdtemp()
paths = []
for k in range(100):
    base = (np.add.outer(np.arange(384), np.arange(1280)) // 3 + rng.integers(0, 40, (384, 1280))).astype(np.uint8)
    p = os.path.join(td, "%03d.png" % k)
    Image.fromarray(base, "L").save(p)
    paths.append(p)
R.cloud_io.read_depth_batch(paths[:4])
t = time.perf_counter(); a = R.cloud_io.read_depth_batch(paths); t1 = time.perf_counter() - t
t = time.perf_counter(); b = np.stack([np.array(Image.open(p)) for p in paths]); t2 = time.perf_counter() - t
print("100 PNGs 1280x384: native batch %.1f ms, PIL loop %.1f ms, equal=%s" % (t1 * 1e3, t2 * 1e3, np.array_equal(a, b)))
pts = (rng.normal(size=(10_000_000, 3)) * 50).astype(np.float32)
t = time.perf_counter(); s = R.cloud_io.format_ply(pts); dt = time.perf_counter() - t
print("format_ply 10M points (two-call protocol, formats twice): %.1f ms -> %.0f Mpoints/s per pass" % (dt * 1e3, 20 / dt))
t = time.perf_counter(); s2 = R.cloud_io.format_xyz_txt(pts.astype(np.float64)); dt = time.perf_counter() - t
print("format_xyz_txt 10M points: %.1f ms -> %.0f Mpoints/s per pass" % (dt * 1e3, 20 / dt))
