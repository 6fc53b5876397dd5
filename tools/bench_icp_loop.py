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


def synthetic_pair(n, s, angle_deg, t_norm, noise, seed):
    """tgt uniform in a 20 m cube; src = the inverse similarity of a shuffled, slightly noisy copy."""
    rng = np.random.default_rng(seed)
    tgt = rng.random((n, 3)) * 20.0
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    a = np.deg2rad(angle_deg)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    R = np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K
    t = rng.normal(size=3)
    t *= t_norm / np.linalg.norm(t)
    T = np.eye(4)
    T[:3, :3] = s * R
    T[:3, 3] = t
    q = tgt[rng.permutation(n)] + rng.normal(size=(n, 3)) * noise
    src = (q - t) @ np.linalg.inv(s * R).T
    return src.astype(np.float32), tgt.astype(np.float32), T


ctx = r3d.Context(0)
# SURVEY.md 8(d) C3 recipe: target = 500k points uniform in a 20 m cube + N(0, 0.01); source = the inverse similarity
# (s=1.7, 10 degrees, |t|=0.5) of a permutation of the noise-free target.  No initial guess is given.
src, tgt, T_true = synthetic_pair(500000, s=1.7, angle_deg=10.0, t_norm=0.5, noise=0.0, seed=7)
tgt = (tgt.astype(np.float64) + np.random.default_rng(8).normal(size=tgt.shape) * 0.01).astype(np.float32)
icp.icp_similarity(src[:2000], tgt[:2000], max_iter=2, ctx=ctx)                                # warm-up
for check_every in (4, 1):
    t0 = time.perf_counter()
    T, info = icp.icp_similarity(src, tgt, ctx=ctx, check_every=check_every)
    dt = time.perf_counter() - t0
    print("auto init, check_every=%d: %d coarse + %d fine iterations in %.1f ms wall incl. upload, both index builds, spacing "
          "probe; |T - T_true|max = %.2e, final rms %.3e, dead zone %.3f"
          % (check_every, info["coarse_iterations"], info["iterations"], dt * 1e3, np.abs(T - T_true).max(),
             info["rms_history"][-1], info["dead_zone"]))
T, info = icp.icp_similarity(src, tgt, ctx=ctx, profile=True)
print("stages (ms, synced):", {k: round(v, 2) for k, v in info["timings_ms"].items()})
# the fine loop alone, from a near-aligned start, fused+device-solve vs stepped from the host, culled vs brute force
near = (src.astype(np.float64) @ (T_true[:3, :3] * 1.003).T + T_true[:3, 3] + 0.01).astype(np.float32)
for culled in (True, False):
    for fused in (True, False):
        dev = icp.IcpDevice(near, tgt, ctx, culled)
        dev.state_reset()
        ctx.sync()
        n_it = 8 if culled else 3
        t0 = time.perf_counter()
        if fused:
            dev.iterate(n_it)
            st = dev.state()
        else:
            for _ in range(n_it):
                dev.nn()
                dev.move_source(icp.umeyama_from_sums(dev.sums()))
            ctx.sync()
        dt = time.perf_counter() - t0
        print("%-11s %-44s %.3f ms/iteration (%d iterations)"
              % ("culled NN" if culled else "brute force",
                 "one enqueue, fused sums, device solve" if fused else "host-stepped: nn, sums + D2H, host SVD, apply",
                 dt * 1e3 / n_it, n_it))
        dev.free()
ctx.close()
