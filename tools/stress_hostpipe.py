"""The host pipeline (pageable NumPy in and out, staged through pinned chunks by the copy crew) against the same call on pinned
arrays (no staging copies), random batch shapes whose chunks have every kind of size, bit for bit; and r3d_download of random
byte counts.  usage: python tools/stress_hostpipe.py [seconds] [seed]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
R = importlib.import_module("3d_reconstruction_system_amd")

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
ctx = R.Context(0)
t0 = time.time()
n = 0
while time.time() - t0 < budget:
    px = int(rng.integers(1, 2_500_000))
    f = int(rng.integers(1, max(2, min(40, 60_000_000 // px))))
    h = int(rng.choice([1, 1, 2, 3, 7, 16]))
    w = max(1, px // h)
    odtype = [np.float32, np.float64][int(rng.integers(0, 2))]
    ddtype = [np.uint8, np.uint16, np.float32][int(rng.integers(0, 3))]
    d = rng.integers(1, 200, (f, h, w)).astype(ddtype)
    q = rng.normal(size=(f, 4))
    t = rng.normal(size=(f, 3)) * 10
    pin_in = ctx.pinned_empty(d.shape, ddtype)
    pin_in[...] = d
    pin_out = ctx.pinned_empty((f * h * w, 3), odtype)
    R.fuse_frames(pin_in, q, t, out_dtype=odtype, ctx=ctx, out=pin_out)
    got = R.fuse_frames(d, q, t, out_dtype=odtype, ctx=ctx)
    if not np.array_equal(got, pin_out, equal_nan=True):
        bad = np.flatnonzero((got != pin_out).any(axis=1))
        raise SystemExit("MISMATCH seed %d case %d shape %s %s->%s: %d rows differ, first %d, last %d of %d"
                         % (seed, n, d.shape, ddtype.__name__, odtype.__name__, bad.size, bad[0], bad[-1], got.shape[0]))
    nbytes = int(rng.integers(1, 90_000_000))
    if rng.integers(0, 2):
        nbytes = int(rng.integers(64, 1300)) * 65536 + int(rng.integers(0, 64))
    src = rng.integers(0, 256, nbytes, dtype=np.uint8)
    buf = ctx.alloc(nbytes).upload(src)
    back = buf.download(np.uint8, nbytes)
    buf.free()
    if not np.array_equal(back, src):
        raise SystemExit("DOWNLOAD MISMATCH seed %d case %d: %d bytes" % (seed, n, nbytes))
    del pin_in, pin_out
    n += 1
    if n % 20 == 0:
        print("%d cases ok (%.0f s)" % (n, time.time() - t0), flush=True)
print("stress OK: %d cases" % n)
