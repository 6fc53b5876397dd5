"""How long the drop-in child spends outside get_file_name(): interpreter start + imports before, teardown after.
usage: python tools/dropin_exit_probe.py   (builds bench.py's 100-frame scene, runs the script with timestamps around main)"""
import os
import subprocess
import sys
import tempfile
import time
import shutil
import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W, frames = 384, 1280, 100
td = tempfile.mkdtemp(prefix="r3d_exit_", dir="/dev/shm")
try:
    for d in ("depth", "camera_pose", "point", "point_world", "ply"):
        os.makedirs(os.path.join(td, d))
    rng = np.random.default_rng(1234)
    base = 40 + 30 * np.sin(np.add.outer(np.arange(H), np.arange(W + 7 * frames)) / 37.0)
    lines = ["id,tx,ty,tz,qx,qy,qz,qw,name,tail\n"]
    for k in range(frames):
        depth = np.clip(base[:, 7 * k:7 * k + W] + rng.integers(0, 6, (H, W)), 1, 255).astype(np.uint8)
        Image.fromarray(depth, "L").save(os.path.join(td, "depth", "%04d.png" % k), compress_level=1)
        q, t = rng.normal(size=4), rng.normal(size=3) * 10
        lines.append("%d,%r,%r,%r,%r,%r,%r,%r,%04d.png,x\n" % ((k,) + tuple(map(float, t)) + tuple(map(float, q)) + (k,)))
    with open(os.path.join(td, "camera_pose", "image_colmap_simi_2.txt"), "w") as f:
        f.writelines(lines)
    script = os.path.join(ROOT, "3d_reconstruction_system_amd", "transfer", "camera_to_world.py")
    code = ("import time,sys,runpy;t0=time.time();sys.argv=[%r];m=runpy.run_path(%r,run_name='not_main');t1=time.time();"
            "m['main']();t2=time.time();print('STAMPS',t0,t1,t2,flush=True)" % (script, script))
    if os.environ.get("PROBE_FAST_EXIT"):
        code += ";import os;sys.stdout.flush();sys.stderr.flush();os._exit(0)"
    for rep in range(3):
        ts = time.time()
        r = subprocess.run([sys.executable, "-c", code], cwd=td, capture_output=True, text=True, env=dict(os.environ, R3D_TIMING="1"))
        te = time.time()
        st = [ln for ln in r.stdout.splitlines() if ln.startswith("STAMPS")]
        if not st:
            print("failed", r.stderr[-400:])
            break
        t0, t1, t2 = map(float, st[0].split()[1:])
        sys.stdout.write("".join(ln + "\n" for ln in r.stderr.splitlines() if ln.startswith("[r3d timing]")))
        print("spawn->python %.0f ms | imports %.0f ms | main() %.0f ms | exit/teardown %.0f ms | wall %.0f ms"
              % ((t0 - ts) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (te - t2) * 1e3, (te - ts) * 1e3))
finally:
    shutil.rmtree(td, ignore_errors=True)
