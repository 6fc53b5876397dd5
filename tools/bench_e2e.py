"""bench.py's end_to_end object: host buffers over PCIe (child process, --workload e2e) and the camera_to_world.py drop-in."""
import importlib
import json
import os
import sys
import tempfile
import time

import numpy as np

from bench_common import BYTES_PER_POINT, FRAMES_PER_GPU, H, HBM_COPY_GBS, HBM_PEAK_GBS, ROOT, W, cpu_model, run_child  # noqa: F401


def e2e_host(a):
    """end_to_end (i), one JSON line: config 2's batch from PINNED HOST memory to PINNED HOST memory through the C ABI's host
    entry point (r3d_fuse_frames_host: chunks over PCIe both ways at once, kernels in between) -- what a caller that keeps its
    rasters and wants its cloud in host memory gets, and the figure north_star's ">= 2 Gpoints/s per GPU" floor is about.
    Pageable NumPy arrays (staged through the library's pinned ring by host threads) beside it."""
    r3d = importlib.import_module("3d_reconstruction_system_amd")
    L = importlib.import_module("3d_reconstruction_system_amd._lib")
    ctx = r3d.Context(0)
    F = FRAMES_PER_GPU
    n = F * H * W
    rng = np.random.default_rng(1234)
    cam = ctx.camera(H, W, *r3d.REF_INTRINSICS)
    raster = rng.integers(1, 256, size=(F, H, W), dtype=np.uint8)
    tab = np.ascontiguousarray(r3d.pose_table(rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10))
    out = {}
    for label, alloc in (("pinned", ctx.pinned_empty), ("pageable", lambda shape, dt: np.zeros(shape, dt))):
        src, dst = alloc((F, H, W), np.uint8), alloc((n, 3), np.float32)
        src[...] = raster
        dst[...] = 0                                    # touched: page faults are not PCIe
        times = []
        for rep in range(7):
            t0 = time.perf_counter()
            L.check(ctx.lib.r3d_fuse_frames_host(ctx.handle, cam.handle, src.ctypes.data, 0, F, 1.0, tab.ctypes.data,
                                                 dst.ctypes.data, 0))          # returns when the cloud is in `dst`
            times.append(time.perf_counter() - t0)
        sec = sorted(times[2:])[2]
        out[label] = {"ms": round(sec * 1e3, 3), "Gpoints_s": round(n / sec / 1e9, 3),
                      "pcie_GBps_h2d": round(n / sec / 1e9, 2), "pcie_GBps_d2h": round(n * 12 / sec / 1e9, 2)}
        if label == "pinned":     # the cloud that came back is the device-resident launch's, bit for bit (sampled rows)
            d_depth, d_pose, d_xyz = ctx.alloc(n).upload(raster), ctx.alloc(tab.nbytes).upload(tab), ctx.alloc(n * 12)
            r3d.fuse_frames_device(ctx, cam, d_depth.ptr, np.uint8, F, d_pose.ptr, d_xyz.ptr, np.float32)
            want = d_xyz.download(np.float32, n * 3).reshape(-1, 3)
            out["identical_to_device_resident_launch"] = bool(np.array_equal(want[::257], dst[::257]))
            for b in (d_depth, d_pose, d_xyz):
                b.free()
        del src, dst
    out["what"] = ("C2's batch (100 x 1280x384 u8, %d points) host memory -> r3d_fuse_frames_host -> f32 xyz in host memory; median "
                   "of 5 calls after 2; 1 B/point up and 12 B/point down the PCIe link at the same time" % n)
    out["floor_Gpoints_s"] = 2.0
    out["meets_floor"] = bool(out["pinned"]["Gpoints_s"] >= 2.0)
    print(json.dumps(out), flush=True)
    ctx.close()


def e2e_dropin(frames=100):
    """end_to_end (ii): `python camera_to_world.py` -- the drop-in with the reference's name and defaults -- run from a
    directory with `frames` synthetic 1280x384 depth PNGs and a pose file, writing EVERY file the reference writes (one
    camera txt per frame, the world txt, the fused ASCII PLY).  Wall seconds of the child process, interpreter start included.
    CPU-only here: the scene is made before anything touches the GPU, the script is its own process."""
    import shutil
    import subprocess
    from PIL import Image
    td = tempfile.mkdtemp(prefix="r3d_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        for d in ("depth", "camera_pose", "point", "point_world", "ply"):
            os.makedirs(os.path.join(td, d))
        rng = np.random.default_rng(1234)
        base = 40 + 30 * np.sin(np.add.outer(np.arange(H), np.arange(W + 7 * frames)) / 37.0)
        lines = ["id,tx,ty,tz,qx,qy,qz,qw,name,tail\n"]
        t0 = time.perf_counter()
        for k in range(frames):
            depth = np.clip(base[:, 7 * k:7 * k + W] + rng.integers(0, 6, (H, W)), 1, 255).astype(np.uint8)
            Image.fromarray(depth, "L").save(os.path.join(td, "depth", "%04d.png" % k), compress_level=1)
            q, t = rng.normal(size=4), rng.normal(size=3) * 10
            lines.append("%d,%r,%r,%r,%r,%r,%r,%r,%04d.png,x\n" % ((k,) + tuple(map(float, t)) + tuple(map(float, q)) + (k,)))
        with open(os.path.join(td, "camera_pose", "image_colmap_simi_2.txt"), "w") as f:
            f.writelines(lines)
        prep = time.perf_counter() - t0
        script = os.path.join(ROOT, "3d_reconstruction_system_amd", "transfer", "camera_to_world.py")
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, script], cwd=td, capture_output=True, text=True, timeout=300)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            return {"failed": "camera_to_world.py exited with %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:])}
        written = 0
        for d in ("point", "point_world", "ply"):
            for name in os.listdir(os.path.join(td, d)):
                written += os.path.getsize(os.path.join(td, d, name))
        pts = frames * H * W
        return {"frames": frames, "points": pts, "wall_s": round(wall, 3), "Mpoints_s": round(pts / wall / 1e6, 1),
                "s_per_frame": round(wall / frames, 5), "bytes_written": written, "scene_prep_s_not_counted": round(prep, 2),
                "what": "python camera_to_world.py (drop-in, reference defaults) on %d synthetic 1280x384 PNGs in %s: PNG decode, "
                        "one fused launch, %d camera txt files + world txt + fused ASCII PLY; wall clock of the child process"
                        % (frames, "/dev/shm" if td.startswith("/dev/shm") else "a temp dir", frames)}
    except Exception as e:  # pragma: no cover
        return {"failed": "%s: %s" % (type(e).__name__, str(e)[:200])}
    finally:
        shutil.rmtree(td, ignore_errors=True)



def end_to_end_children():
    """Both end_to_end legs, run BEFORE the parent touches the GPU (same rule as the regimes child)."""
    return {"host_buffers": run_child("e2e"), "dropin_camera_to_world": e2e_dropin()}
