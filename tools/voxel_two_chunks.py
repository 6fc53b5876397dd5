"""The sort-merge insert beyond one chunk: 2^27 + 5 M points (it works through the cloud 2^27 points at a time) against the CAS
path on the same cloud -- same count, same sorted codes -- and a second insert of the same points (nothing new).
usage: python tools/voxel_two_chunks.py"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
R = importlib.import_module("3d_reconstruction_system_amd")
V = importlib.import_module("3d_reconstruction_system_amd.voxelmap")

ctx = R.Context(0)
n = (1 << 27) + 5_000_000
rng = np.random.default_rng(1)
print("making %d points" % n, flush=True)
cloud = np.empty((n, 3), np.float32)
for lo in range(0, n, 1 << 24):
    hi = min(n, lo + (1 << 24))
    cloud[lo:hi] = rng.uniform(-400, 400, (hi - lo, 3))
d = ctx.alloc(cloud.nbytes).upload(cloud)
del cloud
got = {}
for path in (1, 2):
    ctx.set_tuning("voxel_path", path)
    vs = V.VoxelSet(0.1, 1 << 29, ctx)
    t = time.perf_counter()
    vs.insert_device(d.ptr, n)
    st = vs.stats()
    dt = time.perf_counter() - t
    assert ctx.get_tuning("voxel_last_path") == path
    vs.insert_device(d.ptr, n)
    assert vs.stats() == st, (vs.stats(), st)
    got[path] = (st, vs.codes())
    print("path %d: %s in %.1f ms, %d codes" % (path, st, dt * 1e3, got[path][1].shape[0]), flush=True)
    vs.close()
ctx.set_tuning("voxel_path", 0)
assert got[1][0] == got[2][0]
assert np.array_equal(got[1][1], got[2][1])
print("two-chunk sort-merge insert == CAS insert: %d voxels" % got[1][1].shape[0])
