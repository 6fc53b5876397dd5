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
while time.time() - t0 < budget:
    T.random_shape_sweep(R, ctx, seed + n, 60)
    n += 1
    print("seed %d ok (%d sweeps of 60 cases, %.0f s)" % (seed + n - 1, n, time.time() - t0), flush=True)
print("stress OK: %d cases" % (60 * n))
