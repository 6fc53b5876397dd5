#!/usr/bin/env python3
"""A/B of the camera_to_world.py drop-in's text path on the bench's 100-frame scene: text formatted on the GPU (default)
vs on the host (R3D_HOST_TEXT=1), wall time of the child process and the script's own stage times (R3D_TIMING=1)."""
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F, H, W = int(os.environ.get("FRAMES", "100")), 384, 1280
td = tempfile.mkdtemp(prefix="r3d_ab_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    for d in ("depth", "camera_pose", "point", "point_world", "ply"):
        os.makedirs(os.path.join(td, d))
    rng = np.random.default_rng(1234)
    base = 40 + 30 * np.sin(np.add.outer(np.arange(H), np.arange(W + 7 * F)) / 37.0)
    lines = ["id,tx,ty,tz,qx,qy,qz,qw,name,tail\n"]
    for k in range(F):
        depth = np.clip(base[:, 7 * k:7 * k + W] + rng.integers(0, 6, (H, W)), 1, 255).astype(np.uint8)
        Image.fromarray(depth, "L").save(os.path.join(td, "depth", "%04d.png" % k), compress_level=1)
        q, t = rng.normal(size=4), rng.normal(size=3) * 10
        lines.append("%d,%r,%r,%r,%r,%r,%r,%r,%04d.png,x\n" % ((k,) + tuple(map(float, t)) + tuple(map(float, q)) + (k,)))
    open(os.path.join(td, "camera_pose", "image_colmap_simi_2.txt"), "w").writelines(lines)
    script = os.path.join(ROOT, "3d_reconstruction_system_amd", "transfer", "camera_to_world.py")
    digests = {}
    for rep in range(int(os.environ.get("REPS", "3"))):
        for label, env in (("gpu_text", {}), ("host_text", {"R3D_HOST_TEXT": "1"})):
            t0 = time.perf_counter()
            r = subprocess.run([sys.executable, script], cwd=td, env=dict(os.environ, R3D_TIMING="1", **env), capture_output=True, text=True)
            dt = time.perf_counter() - t0
            assert r.returncode == 0, r.stderr[-2000:]
            print("%-10s rep %d: %.3f s wall" % (label, rep, dt), flush=True)
            if rep == 0:
                print(r.stderr.strip())
                import hashlib
                h = hashlib.sha256()
                for rel in ["ply/small_035_p8.ply", "point_world/small_worldpoint_5_23_5.txt", "point/0000.txt", "point/%04d.txt" % (F - 1)]:
                    h.update(open(os.path.join(td, rel), "rb").read())
                digests[label] = h.hexdigest()
    print("same bytes either way:", digests["gpu_text"] == digests["host_text"])
finally:
    shutil.rmtree(td, ignore_errors=True)
