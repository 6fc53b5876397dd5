#!/usr/bin/env python3
"""What the culled nearest-neighbour search does per workgroup on the two-view (surface) pair and on C3's volumetric pair:
time, tile sweeps per workgroup, cold and warm (the sources do not move between the timed queries: the warm bound is the exact
answer -- the best case for the culling, what is left is the walk itself)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
r3d = importlib.import_module("3d_reconstruction_system_amd")
icp = importlib.import_module("3d_reconstruction_system_amd.icp")
S = importlib.import_module("3d_reconstruction_system_amd.synthetic")
ctx = r3d.Context(0)


def probe(name, dev, n, m):
    for mode, knob in (("cold", 1), ("warm bound, LDS kernel", 2), ("warm, wave-local kernel", 3)):
        ctx.set_tuning("nn_warm", knob)
        dev.nn()
        ctx.sync()
        ts = []
        for _ in range(20):
            ctx.timer_start()
            dev.nn()
            ts.append(ctx.timer_stop())
        swept = dev.index.query(dev.d_src.ptr, dev.n, dev.d_idx.ptr, dev.d_d2.ptr, want_stats=True, presorted=True)
        wgs = -(-n // 256)
        print("%-28s %-24s: %.1f us (min %.1f), %.2f tiles opened per 256 sources (%d x 256 sources, %d tiles)"
              % (name, mode, sorted(ts)[10] * 1e3, min(ts) * 1e3, swept / wgs, wgs, -(-m // 1024)), flush=True)
    ctx.set_tuning("nn_warm", 0)


v = S.two_views(480, 640, yaw_deg=15.0, baseline=(0.35, 0.05, -0.2), depth_noise=0.001)
pa, pb = r3d.unproject(v["depth_a"], v["K"], ctx=ctx), r3d.unproject(v["depth_b"], v["K"], ctx=ctx)
dev = icp.PlaneIcpDevice(pb, pa, tgt_shape=(480, 640), ctx=ctx, init=v["T_ab"])
probe("two views 480x640 (surfaces)", dev, dev.n, dev.m)
rng = np.random.default_rng(0)
m = 500000
tgt = (rng.random((m, 3)) * 20).astype(np.float32)
src = (tgt[rng.permutation(m)] * 1.01 + 0.02).astype(np.float32)
devc = icp.IcpDevice(src, tgt, ctx, culled=True)
devc.sort_source()
probe("C3 500k x 500k (volume)", devc, m, m)
