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

# transfer_T_icp.py --estimate at BASELINE config 3: two 500k-point camera txts -> T_data.txt (GPU ICP from no guess) ->
# merged world txt + PLY.  File parsing and formatting are the library's host code.
r3d = importlib.import_module("3d_reconstruction_system_amd")
td = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
for d in ("point", "point_world", os.path.join("ply", "icp")):
    os.makedirs(os.path.join(td, d))
n = int(os.environ.get("ICP_POINTS", "500000"))
tgt = rng.random((n, 3)) * 20.0
ax = rng.normal(size=3)
ax /= np.linalg.norm(ax)
a = np.deg2rad(10.0)
K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
Rm = np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K
tv = rng.normal(size=3)
tv *= 0.5 / np.linalg.norm(tv)
src = (tgt[rng.permutation(n)] - tv) @ np.linalg.inv(1.7 * Rm).T
r3d.cloud_io.write_xyz_txt(os.path.join(td, "point", "0.txt"), (tgt + rng.normal(size=tgt.shape) * 0.01).astype(np.float32))
r3d.cloud_io.write_xyz_txt(os.path.join(td, "point", "24.txt"), src.astype(np.float32))
script = os.path.join(ROOT, "3d_reconstruction_system_amd", "other_tools", "transfer_T_icp.py")
t0 = time.perf_counter()
r = subprocess.run([sys.executable, script, "--estimate"], cwd=td, capture_output=True, text=True)
dt = time.perf_counter() - t0
assert r.returncode == 0, r.stderr[-2000:]
T = r3d.get_T(os.path.join(td, "T_data.txt"))
T_true = np.eye(4)
T_true[:3, :3], T_true[:3, 3] = 1.7 * Rm, tv
print("transfer_T_icp.py --estimate, two %d-point txt clouds (s=1.7, 10 deg, |t|=0.5): %.2f s wall incl. interpreter start; "
      "|T - T_true|max = %.1e; merged PLY %.0f MB" % (n, dt, np.abs(T - T_true).max(),
                                                      os.path.getsize(os.path.join(td, "ply", "icp", "024.ply")) / 1e6))
shutil.rmtree(td)
