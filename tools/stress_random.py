"""The random-shape sweep of tests/test_gpu_fusion.py (every fuse / unproject / colour kernel against the oracle) under fresh
seeds, for a given number of seconds.  usage: python tools/stress_random.py [seconds] [first_seed]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
T = importlib.import_module("test_gpu_fusion")
R = importlib.import_module("3d_reconstruction_system_amd")

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
ctx = R.Context(0)
t0 = time.time()
n = 0
O = importlib.import_module("oracle.fusion_ref")
import numpy as np


def apply_sweep(seed, cases):
    """apply-T / SE(3) on clouds of random length (tile tails, one point, odd counts) and type, any grid, against the oracle
    (tolerance of SURVEY 8d); the two output widths against each other bit for bit."""
    rng = np.random.default_rng(seed)
    for case in range(cases):
        n_pts = int(rng.choice([1, 2, 255, 256, 257, 1023, 1024, 1025, 4097])) if case % 4 == 0 else int(10 ** rng.uniform(0, 5.5))
        idt = [np.float32, np.float64][int(rng.integers(0, 2))]
        odt = [np.float32, np.float64][int(rng.integers(0, 2))]
        p = (rng.normal(size=(n_pts, 3)) * 10 ** rng.uniform(-2, 3)).astype(idt)
        Tm = np.eye(4)
        Tm[:3, :3] = rng.uniform(0.2, 3) * np.asarray(R.scipy_transfer(rng.normal(size=4)))
        Tm[:3, 3] = rng.normal(size=3) * 5
        rinv = np.asarray(R.scipy_transfer(rng.normal(size=4)))
        t = rng.normal(size=3) * 10
        ctx.set_tuning("apply_blocks", int(rng.choice([0, 0, 1, 3, 17])))
        try:
            got = R.apply_T(p, Tm, out_dtype=odt, ctx=ctx)
            got2 = R.se3_apply(p, rinv, t, out_dtype=odt, ctx=ctx)
        finally:
            ctx.set_tuning("apply_blocks", 0)
        T.check(got, O.apply_T(p, Tm), odt)
        T.check(got2, O.se3_apply(p, rinv, t), odt)
        if odt == np.float64:   # the f32-output kernel = the f64-output kernel rounded once (the oracle's matmul has its own summation order)
            assert np.array_equal(R.apply_T(p, Tm, out_dtype=np.float32, ctx=ctx), got.astype(np.float32)), (seed, case, n_pts, idt)
            assert np.array_equal(R.se3_apply(p, rinv, t, out_dtype=np.float32, ctx=ctx), got2.astype(np.float32)), (seed, case, n_pts, idt)


while time.time() - t0 < budget:
    T.random_shape_sweep(R, ctx, seed + n, 60)
    apply_sweep(seed + n, 40)
    n += 1
    print("seed %d ok (%d sweeps of 60 cases, %.0f s)" % (seed + n - 1, n, time.time() - t0), flush=True)
print("stress OK: %d fuse cases, %d apply cases" % (60 * n, 40 * n))
