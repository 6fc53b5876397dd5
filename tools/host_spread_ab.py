"""A/B of the host pools' placement hint (R3D_HOST_SPREAD, csrc/r3d_hostpool.h) on one box: the drop-in leg of bench.py's
end_to_end and the batch decoders, hint on / off alternately.  usage: python tools/host_spread_ab.py [rounds]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, json, tempfile, importlib
import numpy as np
sys.path.insert(0, %r)
import bench
from PIL import Image
R = importlib.import_module("3d_reconstruction_system_amd")
out = {"dropin_wall_s": bench.e2e_dropin().get("wall_s")}
d = tempfile.mkdtemp(dir="/dev/shm")
rng = np.random.default_rng(0)
H, W = 384, 1280
yy, xx = np.mgrid[0:H, 0:W]
img = (np.stack([128 + 100 * np.sin(xx / 17 + yy / 9), 128 + 90 * np.cos(xx / 5.0), 100 + yy %% 97], 2) + rng.normal(0, 10, (H, W, 3))).clip(0, 255).astype(np.uint8)
png, jpg = [], []
for k in range(100):
    p = os.path.join(d, "%%d.png" %% k); Image.fromarray(img[..., 0] + np.uint8(k), "L").save(p); png.append(p)
    p = os.path.join(d, "%%d.jpg" %% k); Image.fromarray(img, "RGB").save(p, quality=90); jpg.append(p)
def med(f, n=5):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return round(sorted(ts)[n // 2] * 1e3, 2), round(ts[0] * 1e3, 2)
for name, f in (("png_grey_100", lambda: R.cloud_io.read_depth_batch(png)), ("jpeg_grey_100", lambda: R.cloud_io.read_depth_batch(jpg)),
                ("jpeg_rgb_100", lambda: R.cloud_io.read_rgb_batch(jpg)), ("png_grey_8", lambda: R.cloud_io.read_depth_batch(png[:8])),
                ("jpeg_rgb_8", lambda: R.cloud_io.read_rgb_batch(jpg[:8]))):
    out[name + "_ms_median_first"] = med(f)
xyz = rng.normal(0, 10, (4_000_000, 3)).astype(np.float32)
out["format_ply_4M_ms_median_first"] = med(lambda: R.cloud_io.format_ply(xyz), 3)
import shutil; shutil.rmtree(d)
print(json.dumps(out))
''' % ROOT


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    for r in range(rounds):
        for v in ("1", "0"):
            p = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, R3D_HOST_SPREAD=v))
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            print("spread=%s" % v, line[-1] if line else ("FAILED " + p.stderr[-400:]), flush=True)


if __name__ == "__main__":
    main()
