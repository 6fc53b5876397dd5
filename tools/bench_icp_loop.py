#!/usr/bin/env python3
"""Wall time of the whole ICP similarity loop on two 500k-point clouds (BASELINE config C3)."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
r3d = importlib.import_module("3d_reconstruction_system_amd")
icp = importlib.import_module("3d_reconstruction_system_amd.icp")
from oracle import icp_ref as OI  # noqa: E402  (synthetic data generator only)

ctx = r3d.Context(0)
src, tgt, T_true, _ = OI.synthetic_pair(n_tgt=500000, n_src=500000, s=1.005, angle_deg=0.2, t_norm=0.02, noise=0.002, seed=7)
for culled in (True, False):
    icp.icp_similarity(src[:1000], tgt[:1000], max_iter=2, ctx=ctx, culled=culled)          # warm-up
    t0 = time.perf_counter()
    T, info = icp.icp_similarity(src, tgt, max_iter=20, tol=0.0, ctx=ctx, culled=culled)
    dt = time.perf_counter() - t0
    print("%-12s %d iterations in %.1f ms (%.2f ms/iteration incl. upload, index build, host SVD); |T - T_true|max = %.2e, rms %.3e"
          % ("culled NN" if culled else "brute force", info["iterations"], dt * 1e3, dt * 1e3 / info["iterations"],
             np.abs(T - T_true).max(), info["rms_history"][-1]))
ctx.close()
