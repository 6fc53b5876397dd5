#!/usr/bin/env python3
"""End-to-end wall time of the camera_to_world drop-in on a synthetic 100-frame 1280x384 scene (PNG files in,
fused ASCII PLY out), i.e. what a user of the reference's script sees.  The reference needs ~12 s per frame."""
import importlib
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

F, H, W = int(os.environ.get("FRAMES", "100")), 384, 1280
td = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
for d in ("depth", "camera_pose", "point", "point_world", "ply"):
    os.makedirs(os.path.join(td, d))
rng = np.random.default_rng(1234)
lines = ["id,tx,ty,tz,qx,qy,qz,qw,name,tail\n"]
for k in range(F):
    depth = np.clip(40 + 30 * np.sin(np.add.outer(np.arange(H), np.arange(W)) / 37.0 + k) + rng.integers(0, 6, (H, W)), 1, 255)
    Image.fromarray(depth.astype(np.uint8), "L").save(os.path.join(td, "depth", "%04d.png" % k))
    q, t = rng.normal(size=4), rng.normal(size=3) * 10
    lines.append("%d,%r,%r,%r,%r,%r,%r,%r,%04d.png,x\n" % ((k,) + tuple(map(float, t)) + tuple(map(float, q)) + (k,)))
open(os.path.join(td, "camera_pose", "image_colmap_simi_2.txt"), "w").writelines(lines)
script = os.path.join(ROOT, "3d_reconstruction_system_amd", "transfer", "camera_to_world.py")
for label, env in (("fused PLY only (R3D_SKIP_INTERMEDIATE=1)", {"R3D_SKIP_INTERMEDIATE": "1"}),
                   ("all reference files (per-frame camera txt + world txt + PLY)", {})):
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, script], cwd=td, env=dict(os.environ, **env), capture_output=True, text=True)
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr[-2000:]
    ply = os.path.getsize(os.path.join(td, "ply", "small_035_p8.ply"))
    print("%-62s %6.2f s wall incl. interpreter start  (%d frames, %.1f Mpoints, PLY %.2f GB) -> %.1f Mpoints/s"
          % (label, dt, F, F * H * W / 1e6, ply / 1e9, F * H * W / dt / 1e6))
import shutil
shutil.rmtree(td)
