"""The native JPEG decoders against libjpeg-turbo (through PIL) on random files: sizes 1..400 a side, noise / smooth / flat /
saturated content, qualities 1..100, 4:4:4 / 4:2:2 / 4:2:0, grey, optimised tables, restart intervals.  Grey output against
PIL's draft('L') decode, colour output against convert('RGB'), byte for byte.  CPU only.
usage: python tools/stress_jpeg.py [seconds] [seed]"""
import importlib
import io
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
td = tempfile.mkdtemp()
p = os.path.join(td, "t.jpg")
t0 = time.time()
n = skipped = 0
while time.time() - t0 < budget:
    h, w = int(rng.integers(1, 400)), int(rng.integers(1, 400))
    kind = int(rng.integers(0, 5))
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == 0:
        img = rng.integers(0, 256, (h, w, 3))
    elif kind == 1:
        img = np.stack([128 + 100 * np.sin(xx / rng.uniform(2, 40) + yy / rng.uniform(2, 40)), 128 + 90 * np.cos(xx / rng.uniform(1, 9)),
                        (yy * rng.uniform(0.2, 3)) % 256], 2) + rng.normal(0, rng.uniform(0, 30), (h, w, 3))
    elif kind == 2:
        img = np.full((h, w, 3), rng.integers(0, 256, 3))
    elif kind == 3:
        img = rng.choice([0, 255], (h, w, 3))                      # saturated: the range-limit table's ends
    else:
        img = (rng.integers(0, 2, (h // 8 + 1, w // 8 + 1, 3)) * 255).repeat(8, 0).repeat(8, 1)[:h, :w]   # hard block edges
    img = np.clip(img, 0, 255).astype(np.uint8)
    kw = dict(quality=int(rng.integers(1, 101)))
    grey = rng.integers(0, 5) == 0
    if not grey:
        kw["subsampling"] = int(rng.integers(0, 3))
    if rng.integers(0, 3) == 0:
        kw["optimize"] = True
    r = int(rng.integers(0, 4))
    if r == 1:
        kw["restart_marker_blocks"] = int(rng.integers(1, 20))
    elif r == 2:
        kw["restart_marker_rows"] = int(rng.integers(1, 4))
    try:
        Image.fromarray(img[..., 0] if grey else img, "L" if grey else "RGB").save(p, **kw)
    except (OSError, TypeError, ValueError):
        skipped += 1
        continue
    im = Image.open(p)
    im.draft("L", im.size)
    want_l = np.asarray(im)
    want_rgb = np.asarray(Image.open(p).convert("RGB"))
    got_l = R.cloud_io.read_depth_gray(p)
    got_rgb = R.cloud_io.read_rgb_batch([p])[0]
    if not (np.array_equal(got_l, want_l) and np.array_equal(got_rgb, want_rgb)):
        keep = os.path.join(tempfile.gettempdir(), "r3d_stress_jpeg_fail_%d_%d.jpg" % (seed, n))
        os.replace(p, keep)
        raise SystemExit("MISMATCH seed %d case %d: %dx%d kind %d %s grey=%s -> %s" % (seed, n, w, h, kind, kw, grey, keep))
    n += 1
print("stress OK: %d files (%d the encoder refused)" % (n, skipped))
