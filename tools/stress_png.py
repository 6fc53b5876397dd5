"""The native PNG decoders on random files: sizes 1..300 a side, 8/16-bit grey, grey+alpha, RGB, RGBA (8 and 16 bit), every
zlib level, PIL's adaptive filters.  Grey path against the IMREAD_GRAYSCALE rules computed from PIL's samples, colour path
against PIL's RGB bytes.  CPU only.   usage: python tools/stress_png.py [seconds] [seed]"""
import importlib
import os
import sys
import tempfile
import time

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
R = importlib.import_module("3d_reconstruction_system_amd")

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
p = os.path.join(tempfile.mkdtemp(), "t.png")
t0 = time.time()
n = 0
while time.time() - t0 < budget:
    h, w = int(rng.integers(1, 300)), int(rng.integers(1, 300))
    mode = ["L", "I;16", "LA", "RGB", "RGBA"][int(rng.integers(0, 5))]
    ch = {"L": 1, "I;16": 1, "LA": 2, "RGB": 3, "RGBA": 4}[mode]
    yy, xx = np.mgrid[0:h, 0:w]
    smooth = rng.integers(0, 2)
    top = 65535 if mode == "I;16" else 255
    if smooth:
        a = np.stack([(np.sin(xx / rng.uniform(2, 30) + k) + np.cos(yy / rng.uniform(2, 30)) + 2) / 4 * top for k in range(ch)], 2)
        a = a + rng.integers(0, 4, a.shape)
    else:
        a = rng.integers(0, top + 1, (h, w, ch))
    if mode in ("RGB", "RGBA") and rng.integers(0, 3) == 0:
        a[..., 1] = a[..., 0]
        a[..., 2] = a[..., 0]                                       # R = G = B: kept as is by libpng's rule
    a = np.clip(a, 0, top).astype(np.uint16 if mode == "I;16" else np.uint8)
    Image.fromarray(a[..., 0] if ch == 1 else a, mode).save(p, compress_level=int(rng.integers(0, 10)))
    got = R.cloud_io.read_depth_gray(p)
    if mode == "L":
        want = a[..., 0]
    elif mode == "I;16":
        want = (a[..., 0] >> 8).astype(np.uint8)
    elif mode == "LA":
        want = a[..., 0]
    else:
        want = R.cloud_io.rgb_to_gray(a[..., :3], "opencv_png")
    assert np.array_equal(got, want), ("grey", seed, n, mode, h, w)
    if mode != "I;16":
        rgb = R.cloud_io.read_rgb_batch([p])[0]
        assert np.array_equal(rgb, np.asarray(Image.open(p).convert("RGB"))), ("rgb", seed, n, mode, h, w)
    n += 1
print("stress OK: %d files" % n)
