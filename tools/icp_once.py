#!/usr/bin/env python3
"""One C3-recipe ICP estimate (after a small warm-up) -- the program to put behind `rocprofv3 --kernel-trace --stats --`."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
r3d = importlib.import_module("3d_reconstruction_system_amd")
icp = importlib.import_module("3d_reconstruction_system_amd.icp")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
rng = np.random.default_rng(7)
tgt = rng.random((n, 3)) * 20.0
axis = rng.normal(size=3)
axis /= np.linalg.norm(axis)
a = np.deg2rad(10.0)
K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
R = np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K
t = rng.normal(size=3)
t *= 0.5 / np.linalg.norm(t)
src = ((tgt[rng.permutation(n)] - t) @ np.linalg.inv(1.7 * R).T).astype(np.float32)
tgt = (tgt + np.random.default_rng(8).normal(size=tgt.shape) * 0.01).astype(np.float32)
ctx = r3d.Context(0)
T, info = icp.icp_similarity(src, tgt, ctx=ctx, profile=True)
T, info = icp.icp_similarity(src, tgt, ctx=ctx, profile=True)
T_true = np.eye(4)
T_true[:3, :3], T_true[:3, 3] = 1.7 * R, t
print("coarse %d fine %d  |T-T_true| %.2e  stages %s" % (info["coarse_iterations"], info["iterations"],
                                                          np.abs(T - T_true).max(),
                                                          {k: round(v, 2) for k, v in info["timings_ms"].items()}))
print("trace:", info.get("trace_ms"))
ctx.close()
