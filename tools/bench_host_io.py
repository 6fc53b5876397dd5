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
# files, the way the drop-in writes them (tmpfs when there is one: the formatter, not a disk, is what is timed)
out = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
H, W, F = 384, 1280, 100
fx, fy, cx, cy = R.REF_INTRINSICS
depth = rng.integers(1, 256, (F, H, W), dtype=np.uint8)
Z = depth.astype(np.float64)
cam = np.stack([((np.arange(W)[None, None, :] - cx) / fx) * Z, ((np.arange(H)[None, :, None] - cy) / fy) * Z, Z], -1).reshape(-1, 3)
world = cam @ np.array([[0.36, 0.48, -0.8], [-0.8, 0.6, 0.0], [0.48, 0.64, 0.6]]) + np.array([1.5, -2.25, 10.125])
for rep in range(2):
    t = time.perf_counter(); R.cloud_io.write_xyz_txt_batch([os.path.join(out, "%d.txt" % k) for k in range(F)], cam, z_raw=depth); t1 = time.perf_counter() - t
    t = time.perf_counter(); R.cloud_io.write_ply(os.path.join(out, "w.ply"), world); t2 = time.perf_counter() - t
    t = time.perf_counter(); R.cloud_io.write_xyz_txt(os.path.join(out, "w.txt"), world[:H * W]); t3 = time.perf_counter() - t
    nb = sum(os.path.getsize(os.path.join(out, "%d.txt" % k)) for k in range(F))
    print("%d camera txt files (%.2f GB) %.0f ms = %.0f Mpoints/s | fused PLY (%.2f GB) %.0f ms = %.0f Mpoints/s | one world txt %.1f ms"
          % (F, nb / 1e9, t1 * 1e3, F * H * W / t1 / 1e6, os.path.getsize(os.path.join(out, "w.ply")) / 1e9, t2 * 1e3, F * H * W / t2 / 1e6, t3 * 1e3))
os.environ["R3D_HOST_THREADS"] = "1"
import subprocess
code = ("import sys,time,importlib,numpy as np;sys.path.insert(0,%r);R=importlib.import_module('3d_reconstruction_system_amd');"
        "rng=np.random.default_rng(0);a=rng.normal(size=(1000000,3))*50;"
        "t=time.perf_counter();R.cloud_io.format_xyz_txt(a);t1=time.perf_counter()-t;"
        "t=time.perf_counter();R.cloud_io.format_ply(a);t2=time.perf_counter()-t;"
        "print('one thread: repr() txt %%.0f ns per number, %%%%.4f PLY %%.0f ns per number' %% (t1/6e6*1e9, t2/6e6*1e9))" % ROOT)
print(subprocess.run([sys.executable, "-c", code], capture_output=True, text=True).stdout.strip())
import shutil
shutil.rmtree(out, ignore_errors=True)
shutil.rmtree(td, ignore_errors=True)
