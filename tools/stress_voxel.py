"""Random clouds through every way into the occupied-voxel set -- CAS with and without the LDS set, sort-merge, the library's
own choice, two inserts into one set by different paths, ready-made codes -- against the oracle's set of the same f32 cloud.
Cloud kinds: uniform boxes of every scale (one voxel per point ... thousands of points per voxel), planes, points on voxel
faces, clouds with NaN / inf / out-of-range rows; table sizes from nearly full to mostly empty (where region spills differ).
usage: python tools/stress_voxel.py [seconds] [seed]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
R = importlib.import_module("3d_reconstruction_system_amd")
V = importlib.import_module("3d_reconstruction_system_amd.voxelmap")
OM = importlib.import_module("oracle.octomap_ref")

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
ctx = R.Context(0)
t0 = time.time()
case = 0
while time.time() - t0 < budget:
    n = int(10 ** rng.uniform(2.0, 6.6))
    res = float(rng.choice([0.1, 0.05, 0.02, 0.25, 1.0 / 3.0, 0.007]))
    kind = int(rng.integers(0, 5))
    if kind == 0:
        cloud = rng.uniform(-1, 1, (n, 3)) * 10 ** rng.uniform(-1.5, 2.5)
    elif kind == 1:                                         # a few planes, noisy
        cloud = rng.uniform(-20, 20, (n, 3))
        cloud[:, int(rng.integers(0, 3))] = rng.choice([-3.0, 0.0, 2.5], n) + rng.normal(0, 0.003, n)
    elif kind == 2:                                         # lattice points: exactly on voxel faces and next to them
        cloud = rng.integers(-300, 300, (n, 3)) * res + rng.choice([0.0, 1e-7, -1e-7], (n, 3))
    elif kind == 3:                                         # heavy duplication in runs (a scan)
        base = rng.uniform(-30, 30, (max(1, n // 64), 3))
        cloud = np.repeat(base, 64, axis=0)[:n] + rng.normal(0, res / 8, (min(n, base.shape[0] * 64), 3))
        n = cloud.shape[0]
    else:                                                   # one voxel per point, far apart
        cloud = rng.uniform(-3000, 3000, (n, 3))
    cloud = cloud.astype(np.float32)
    if rng.integers(0, 3) == 0 and n > 10:
        bad = rng.integers(0, n, max(1, n // 50))
        cloud[bad, rng.integers(0, 3, bad.size)] = rng.choice([np.nan, np.inf, -np.inf, 1e9, -4000.0], bad.size)
    want, dropped = OM.occupied_set(cloud, res)
    n_vox = max(1, len(want))
    log2cap = int(np.ceil(np.log2(n_vox / rng.uniform(0.08, 0.62))))
    log2cap = max(10, min(28, log2cap))
    capacity = 1 << log2cap
    d = ctx.alloc(cloud.nbytes or 16).upload(cloud)
    what = "seed %d case %d: n=%d res=%g kind=%d voxels=%d cap=2^%d" % (seed, case, n, res, kind, len(want), log2cap)
    try:
        for path, dedupe in ((1, 0), (1, 1), (2, 0), (0, 0)):
            ctx.set_tuning("voxel_path", path)
            ctx.set_tuning("voxel_dedupe", dedupe)
            vs = V.VoxelSet(res, capacity, ctx)
            half = n // 2
            if rng.integers(0, 2) and half:                 # two inserts, the second through the OTHER path where there is one
                vs.insert_device(d.ptr, half)
                ctx.set_tuning("voxel_path", {1: 2, 2: 1, 0: 0}[path])
                vs.insert_device(d.ptr + half * 12, n - half)
            else:
                vs.insert_device(d.ptr, n)
            st = vs.stats()
            assert st == {"voxels": len(want), "ignored_points": dropped, "overflow": 0}, (what, path, dedupe, st, dropped)
            got = vs.codes()
            assert np.array_equal(got, want), (what, path, dedupe)
            vs.close()
    finally:
        ctx.set_tuning("voxel_dedupe", 0)
        ctx.set_tuning("voxel_path", 0)
        d.free()
    case += 1
    if case % 25 == 0:
        print("%d cases ok (%.0f s); last: %s" % (case, time.time() - t0, what), flush=True)
print("stress OK: %d cases" % case)
