#!/usr/bin/env python3
"""Headline benchmark: fused depth -> world point-cloud throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]            (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], "C2"): KITTI-sized 1280x384 uint8 depth, 100 frames per GPU,
synthetic (seed 1234), device-resident; one STEP = one pass of the hot path over the batch =
ONE fused unproject + SE(3) launch over all 100 frames (49,152,000 points -> f32 xyz), and for
N > 1 additionally the collective that assembles the fused world cloud on every rank (north_star):
by default (`--assemble auto`) the fastest measured strategy -- normally an all-gather of the INPUTS (depth + poses,
1 B/point) followed by a local fuse of every rank's frames, which is bit-identical to and several times faster than
all-gathering the xyz OUTPUTS (12 B/point) on xGMI.  Weak scaling: every rank owns 100 frames.
`value` is whole-job Mpoints/s = UNIQUE fused points of all ranks / max-over-ranks time.  For N > 1 EVERY assembly
strategy (none / outputs / inputs, ncclAllGather and direct send/recv) is timed before the headline region and printed
under "assemble" with its achieved xGMI GB/s per link; the exchange runs through the library's own RCCL communicator
(r3d_comm_*, C ABI) when it comes up, torch.distributed otherwise ("transport").

Extra objects on the JSON line:
  roofline     -- the fused kernel against the HBM roof: algorithmic bytes (13 B/point) per launch
                  / average launch duration measured with HIP events on the launch stream over the timed region
                  (roofline.sustained: the same launch over >= 4000 launches before it; roofline.cold_inputs: the same
                  launch on rasters that are NOT in the Infinity Cache, measured in this run by a child process).
  cpu_baseline -- the loop-faithful CPU restatement of the reference path (oracle/, test
                  infrastructure; rank 0, N=1 only) timed on a bounded sample, 1 core; with and without the PLY writer.
  end_to_end   -- N=1 only, measured by child processes before this one touches the GPU: (i) pinned host rasters ->
                  r3d_fuse_frames_host -> pinned host xyz (PCIe both ways: north_star's 2 Gpoints/s floor), (ii) the
                  camera_to_world.py drop-in on 100 synthetic 1280x384 PNG files -> every file the reference writes.
"""
import argparse
import importlib
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W, FRAMES_PER_GPU = 384, 1280, 100
BYTES_PER_POINT = 13          # 1 B u8 depth read + 12 B f32 xyz written
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0         # ... and the float4 copy it measures (read + write mixed): the practical ceiling of a 1:1 stream
# N>1 assembly survey: seconds without progress before the pre-measured shards-stay-resident line goes out instead
WATCHDOG_S = int(os.environ.get("R3D_BENCH_WATCHDOG_S", "240"))
XGMI_LINK_GBS = 153.0         # one xGMI link, per direction (7 links per GPU, full mesh of 8)
OVERLAP_CHUNKS = 4            # slices of the pipelined 'inputs' assembly


def cpu_baseline(sample_frames=1):
    """Loop-faithful restatement of camera_to_world.py:67-105 (per-point Python loops + text round
    trip), 1 core, on `sample_frames` frames of the same workload.  Reported, not optimised against."""
    from oracle import fusion_ref as O
    rng = np.random.default_rng(1234)
    depth = rng.integers(1, 256, size=(sample_frames, H, W), dtype=np.uint8)
    q = rng.normal(size=(sample_frames, 4))
    t = rng.normal(size=(sample_frames, 3)) * 10
    with tempfile.TemporaryDirectory() as td:
        t0 = time.perf_counter()
        O.fuse_frames_loop(depth, q, t, td)
        dt = time.perf_counter() - t0
        # ... and the reference's last step, the ASCII PLY of the fused cloud (c2w:112-134), on ONE frame's points
        one = O.fuse_frames(depth[:1], q[:1], t[:1])
        cols = [one[:, 0].tolist(), one[:, 1].tolist(), one[:, 2].tolist()]
        t0 = time.perf_counter()
        O.genply_loop(cols, os.path.join(td, "one.ply"))
        dt_ply = time.perf_counter() - t0
    t0 = time.perf_counter()
    O.fuse_frames(depth, q, t)
    dt_vec = time.perf_counter() - t0
    pts = sample_frames * H * W
    per_frame, per_frame_ply = dt / sample_frames, dt / sample_frames + dt_ply
    return {"value": round(pts / dt / 1e6, 5), "unit": "Mpoints/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
            "sample": "%d frame(s) of 1280x384 u8 (%d points), per-point Python loops + txt round trip as "
                      "camera_to_world.py:67-105, no PLY; %.1f s" % (sample_frames, pts, dt),
            "files_to_files": {"s_per_frame": round(per_frame_ply, 3), "Mpoints_s": round(H * W / per_frame_ply / 1e6, 5),
                               "what": "the same loops PLUS the ASCII PLY writer (genply, camera_to_world.py:112-134; %.2f s for "
                                       "one frame's %d points): the reference's whole per-frame path, files to files"
                                       % (dt_ply, H * W)},
            "vectorised_numpy_fp64_Mpoints_s": round(pts / dt_vec / 1e6, 3),
            "host_cpus": os.cpu_count()}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def pmc_traffic(frames, out_dtype, depth="u8"):
    """HBM bytes per launch from the committed rocprofv3 PMC summary (separate --pmc passes over this same command).
    Counters cannot be read from inside the process, so this is a RECORDED figure: it is emitted only when this run's
    launch is the one that was profiled (same frames, depth and output types), otherwise null."""
    p = os.path.join(ROOT, "profiles", "pmc_fuse_latest.json")
    try:
        with open(p) as f:
            rec = json.load(f)
    except Exception:
        return None, None
    cfg = rec.get("config", {"frames": 100, "out_dtype": "float32", "depth": "u8"})   # r01 file: the C2 default launch
    if (cfg.get("frames"), cfg.get("out_dtype"), cfg.get("depth")) != (frames, out_dtype, depth):
        return None, None
    import hashlib
    with open(p, "rb") as f:
        sha = hashlib.sha256(f.read()).hexdigest()[:12]
    return rec.get("hbm_bytes_per_launch"), "recorded: profiles/pmc_fuse_latest.json @%s (%s)" % (sha, rec.get("collected", "rocprofv3 --pmc passes "
                                                                                                 "over this command"))


def kernel_duration_ms(torch, stream, launch, min_launches=1000, min_ms=100.0, warm=50):
    """Median duration of one launch of the dominant kernel, independent of --steps: after `warm` untimed launches,
    at least `min_launches` launches (and at least `min_ms` of GPU time) timed with HIP events on the launch stream.  Returns (median_ms, mean_ms, n)."""
    for _ in range(warm):
        launch()
    torch.cuda.synchronize()
    per = 100                     # launches between two events, back to back exactly as in the timed region
    durations, total = [], 0.0
    while len(durations) * per < min_launches or total < min_ms:
        n = 5
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        evs[0].record(stream)
        for i in range(n):
            for _ in range(per):
                launch()
            evs[i + 1].record(stream)
        evs[n].synchronize()
        block = [evs[i].elapsed_time(evs[i + 1]) / per for i in range(n)]
        durations += block
        total += sum(block) * per
        if len(durations) * per >= 8000:
            break
    durations.sort()
    return durations[len(durations) // 2], sum(durations) / len(durations), len(durations) * per


def regimes(a):
    """The headline launch OUTSIDE the bench loop's comfortable regime, measured live on this box; one JSON line.
    The headline loop re-reads ONE 49 MB raster, which therefore sits in the 256 MiB Infinity Cache from the second launch
    on; a real pass touches fresh frames.  Here: (1) 16 rotating copies of the raster (786 MB: none of them cached), plain and
    with the library's input staging forced on; (2) the launch right after an H2D upload of fresh frames from pinned host
    memory (where does DMA leave the data?), with staging off / auto / on; (3) BASELINE config 4's whole input -- 1000 frames,
    491.5 M points, 6.4 GB of traffic -- as ONE launch on one GPU.  Run by the N=1 headline as a CHILD process before the
    parent touches the GPU: the same kernel symbol at other regimes must not mix into the rocprofv3 statistics of the parent's
    launches (the committed kernel-trace summary has to describe the launches roofline.kernel_ms describes)."""
    r3d = importlib.import_module("3d_reconstruction_system_amd")
    L = importlib.import_module("3d_reconstruction_system_amd._lib")
    ctx = r3d.Context(0)
    rng = np.random.default_rng(1234)
    F = FRAMES_PER_GPU
    n = F * H * W
    bytes_per_launch = n * BYTES_PER_POINT
    cam = ctx.camera(H, W, *r3d.REF_INTRINSICS)
    raster = rng.integers(1, 256, size=(F, H, W), dtype=np.uint8)
    tab = r3d.pose_table(rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10)
    d_pose, d_xyz = ctx.alloc(tab.nbytes).upload(tab), ctx.alloc(n * 12)
    copies = [ctx.alloc(n).upload(raster) for _ in range(16)]

    def frac(ms):
        return round(bytes_per_launch / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)

    def median_ms(launch, groups=8, per=16, warm=32):
        for _ in range(warm):
            launch()
        ctx.sync()
        t = []
        for _ in range(groups):
            ctx.timer_start()
            for _ in range(per):
                launch()
            t.append(ctx.timer_stop() / per)
        return sorted(t)[len(t) // 2]

    out = {"raster_copies": len(copies)}
    state = {"i": 0}

    def fuse_rotating():
        d = copies[state["i"] % len(copies)]
        state["i"] += 1
        r3d.fuse_frames_device(ctx, cam, d.ptr, np.uint8, F, d_pose.ptr, d_xyz.ptr, np.float32)
    # warm the clocks on the cached launch first (an idle GPU boosts, dips for ~20 ms, then settles)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        r3d.fuse_frames_device(ctx, cam, copies[0].ptr, np.uint8, F, d_pose.ptr, d_xyz.ptr, np.float32)
        ctx.sync()
    for key, knob in (("plain", 1), ("staged", 2), ("auto", 0)):
        ctx.set_tuning("fuse_prefetch", knob)
        s0 = ctx.get_tuning("fuse_sweeps")
        ms = median_ms(fuse_rotating)
        out[key + "_ms"], out[key + "_frac"] = round(ms, 5), frac(ms)
        out[key + "_sweeps_per_launch"] = round((ctx.get_tuning("fuse_sweeps") - s0) / (32 + 8 * 16), 3)
    # for the record, what the library's default costs where it is NOT needed: ONE raster re-read every launch (the parent's
    # bench loop), staging off = the fused kernel alone on cached inputs (rounds 1-2 measured this) / library default
    def fuse_same():
        r3d.fuse_frames_device(ctx, cam, copies[0].ptr, np.uint8, F, d_pose.ptr, d_xyz.ptr, np.float32)
    same = {}
    for key, knob in (("staging_off", 1), ("auto", 0), ("staging_forced", 2)):
        ctx.set_tuning("fuse_prefetch", knob)
        s0 = ctx.get_tuning("fuse_sweeps")
        ms = median_ms(fuse_same, groups=8, per=50, warm=100)
        same[key + "_ms"], same[key + "_frac"] = round(ms, 5), frac(ms)
        same[key + "_sweeps"] = ctx.get_tuning("fuse_sweeps") - s0              # of 500 launches
    out["same_raster_every_launch"] = same
    # (2) fuse right after an H2D upload of fresh frames (pinned host memory -> the same device raster every time)
    host = ctx.pinned_empty((F, H, W), np.uint8)
    host[...] = raster
    h2d = {}
    for key, knob in (("plain", 1), ("staged", 2), ("auto", 0)):
        ctx.set_tuning("fuse_prefetch", knob)
        t = []
        for k in range(24):
            host[0, 0, :16] = k                                           # "fresh": never the bytes that were there before
            # evict: the 15 other copies (737 MB) stream through the cache before the upload lands
            for c in copies[1:]:
                L.check(ctx.lib.r3d_cache_prefetch(ctx.handle, c.ptr, n))
            L.check(ctx.lib.r3d_memcpy_h2d(ctx.handle, copies[0].ptr, host.ctypes.data, n))
            s0 = ctx.get_tuning("fuse_sweeps")
            ctx.timer_start()
            r3d.fuse_frames_device(ctx, cam, copies[0].ptr, np.uint8, F, d_pose.ptr, d_xyz.ptr, np.float32)
            t.append(ctx.timer_stop())
        ms = sorted(t[4:])[10]
        h2d[key + "_ms"], h2d[key + "_frac"] = round(ms, 5), frac(ms)
        h2d[key + "_sweeps_last_launch"] = ctx.get_tuning("fuse_sweeps") - s0
    h2d["note"] = ("single launches, each right after a 49 MB H2D copy from pinned host memory into the raster it reads (the "
                   "other 15 rasters are swept through the cache before the copy); staging off / on / library default")
    out["after_h2d_upload"] = h2d
    ctx.set_tuning("fuse_prefetch", 0)
    out["staging_policy"] = ("by provenance: auto stages a launch whose inputs exceed %d MB unless those bytes are presumed cached "
                             "(read by a launch on this device, fewer than %d MB of other inputs since, not rewritten through the "
                             "library); foreign producers say r3d_ctx_set_tuning('fuse_inputs_fresh', 1)"
                             % (ctx.get_tuning("fuse_stage_auto_mb"), ctx.get_tuning("fuse_resident_mb")))
    # what one sweep costs when it is needed: the raster alone, cold (the other copies went through the cache in between)
    ts = []
    for k in range(12):
        for c in copies[1:9]:
            L.check(ctx.lib.r3d_cache_prefetch(ctx.handle, c.ptr, n))
        ctx.timer_start()
        L.check(ctx.lib.r3d_cache_prefetch(ctx.handle, copies[0].ptr, n))
        ts.append(ctx.timer_stop())
    out["sweep_alone_cold_ms"] = round(sorted(ts[2:])[5], 5)
    out["note"] = ("each launch reads a different copy of the raster (first touch of fresh frames); 'staged' = a read-only sweep "
                   "puts the launch's inputs into the Infinity Cache first; 'auto' = the library's default policy")
    for c in copies[1:]:
        c.free()
    # (3) C4's whole input on ONE GPU: 1000 frames in one call (inputs staged chunk by chunk by default)
    try:
        F4 = 1000
        n4 = F4 * H * W
        d_depth4, d_xyz4 = ctx.alloc(n4), ctx.alloc(n4 * 12)
        L.check(ctx.lib.r3d_memset(ctx.handle, d_depth4.ptr, 0x41, n4))
        tab4 = r3d.pose_table(rng.normal(size=(F4, 4)), rng.normal(size=(F4, 3)) * 10)
        d_pose4 = ctx.alloc(tab4.nbytes).upload(tab4)
        ms = median_ms(lambda: r3d.fuse_frames_device(ctx, cam, d_depth4.ptr, np.uint8, F4, d_pose4.ptr, d_xyz4.ptr, np.float32),
                       groups=5, per=4, warm=12)
        out["c4_1000_frames_one_gpu"] = {"ms": round(ms, 4), "Mpoints_s": round(n4 / ms / 1e3, 1),
                                         "frac": round(n4 * BYTES_PER_POINT / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                         "points": n4, "what": "BASELINE config 4's 1000 frames fused by one call on one GPU "
                                                               "(6.4 GB of traffic, inputs staged through the Infinity Cache)"}
    except Exception as e:  # pragma: no cover
        out["c4_1000_frames_one_gpu"] = {"failed": "%s: %s" % (type(e).__name__, str(e)[:100])}
    print(json.dumps(out), flush=True)
    ctx.close()


def regimes_in_child():
    """Run `bench.py --workload regimes` as a child process (see regimes()); returns its dict, or {"failed": ...}."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", "regimes"], capture_output=True, text=True,
                           timeout=300, cwd=ROOT)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"failed": "child exited with %d: %s" % (r.returncode, (r.stderr or r.stdout)[-200:])}
        return json.loads(lines[-1])
    except Exception as e:  # pragma: no cover
        return {"failed": "%s: %s" % (type(e).__name__, str(e)[:160])}


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
    import subprocess
    out = {}
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", "e2e"], capture_output=True, text=True,
                           timeout=300, cwd=ROOT)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        out["host_buffers"] = json.loads(lines[-1]) if r.returncode == 0 and lines else \
            {"failed": "child exited with %d: %s" % (r.returncode, (r.stderr or r.stdout)[-200:])}
    except Exception as e:  # pragma: no cover
        out["host_buffers"] = {"failed": "%s: %s" % (type(e).__name__, str(e)[:160])}
    out["dropin_camera_to_world"] = e2e_dropin()
    return out


def apply_cpu_baseline(sample_points=200000):
    """Loop-faithful restatement of transfer_T_icp.py:71-97 (local_world with flag=True: per-line parse, np.dot(T, p), three
    list appends, one text line out), 1 core, on a bounded sample.  Reported, not optimised against."""
    from oracle import fusion_ref as O
    rng = np.random.default_rng(1234)
    pts = rng.normal(size=(sample_points, 3)) * 50
    T = np.eye(4)
    T[:3, :3] *= 1.7
    T[:3, 3] = (1, 2, 3)
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "24.txt")
        with open(src, "w") as f:
            for x, y, z in pts.tolist():
                f.write("%r,%r,%r\n" % (x, y, z))
        xs, ys, zs = [], [], []
        t0 = time.perf_counter()
        with open(os.path.join(td, "world.txt"), "w") as fout:
            O.local_world_loop(src, fout, T, xs, ys, zs, True)
        dt = time.perf_counter() - t0
    return {"value": round(sample_points / dt / 1e6, 5), "unit": "Mpoints/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
            "sample": "%d points through local_world's per-line loop (parse, 4x4 dot, text out) as transfer_T_icp.py:71-97; "
                      "%.1f s" % (sample_points, dt), "host_cpus": os.cpu_count()}


def secondary(a):
    """One JSON line for a secondary kernel (single GPU, HIP-event stopwatch of the library on its own stream)."""
    r3d = importlib.import_module("3d_reconstruction_system_amd")
    ctx = r3d.Context(0)
    rng = np.random.default_rng(1234)

    def timed(fn, iters):
        """Median of 5 groups of `iters`/5 launches, after a warm-up of >= 5 launches and >= 60 ms (the clocks of an
        idle GPU boost for the first ~2 ms and then dip for ~20 ms: profiles/r02_*: neither belongs in a rate)."""
        t0 = time.perf_counter()
        k = 0
        while k < 5 or time.perf_counter() - t0 < 0.06:
            fn()
            k += 1
            if k % 5 == 0:
                ctx.sync()
        ctx.sync()
        per = max(iters // 5, 1)
        groups = []
        for _ in range(5):
            ctx.timer_start()
            for _ in range(per):
                fn()
            groups.append(ctx.timer_stop() / per)
        return sorted(groups)[2]

    if a.workload == "apply":
        n = FRAMES_PER_GPU * H * W
        d_in = ctx.alloc(n * 12).upload((rng.normal(size=(n, 3)) * 50).astype(np.float32))
        d_out = ctx.alloc(n * 12)
        T = np.eye(4)
        T[:3, :3] *= 1.7
        T[:3, 3] = (1, 2, 3)
        ms = timed(lambda: r3d.apply_T_device(ctx, d_in.ptr, np.float32, n, T, d_out.ptr, np.float32), max(a.steps // 10, 20))
        gbs = n * 24 / ms / 1e6
        line = {"metric": "Mpoints/s apply-T (4x4 on a 49.2 Mpoint f32 cloud)", "value": round(n / ms / 1e3, 1), "unit": "Mpoints/s",
                "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None, "kernel": "apply_lane_kernel<f32,affine>",
                             "frac_of_measured_copy": round(gbs / HBM_COPY_GBS, 4),
                             "kernel_ms": round(ms, 5), "algorithmic_bytes_per_launch": n * 24}}
        if not a.no_cpu_baseline:
            line["cpu_baseline"] = apply_cpu_baseline()
    elif a.workload == "icp":
        icp = importlib.import_module("3d_reconstruction_system_amd.icp")
        m = 500000
        # SURVEY.md 8(d) C3 recipe: target uniform in a 20 m cube + N(0, 0.01); source = inverse similarity
        # (s=1.7, 10 degrees, |t|=0.5) of a permutation of the noise-free target; no initial guess
        tgt0 = rng.random((m, 3)) * 20
        ax = rng.normal(size=3)
        ax /= np.linalg.norm(ax)
        ang = np.deg2rad(10.0)
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        Rm = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
        tv = rng.normal(size=3)
        tv *= 0.5 / np.linalg.norm(tv)
        T_true = np.eye(4)
        T_true[:3, :3], T_true[:3, 3] = 1.7 * Rm, tv
        src = ((tgt0[rng.permutation(m)] - tv) @ np.linalg.inv(1.7 * Rm).T).astype(np.float32)
        tgt = (tgt0 + rng.normal(size=tgt0.shape) * 0.01).astype(np.float32)
        icp.icp_similarity(src[:3000], tgt[:3000], max_iter=2, ctx=ctx)                      # warm-up
        walls = []
        for _ in range(4):   # the first full-size call also grows the library's scratch buffers (hipMalloc): reported apart
            t0 = time.perf_counter()
            T, info = icp.icp_similarity(src, tgt, ctx=ctx)
            walls.append((time.perf_counter() - t0) * 1e3)
        first_call_ms, wall_ms = walls[0], sorted(walls[1:])[1]
        near = (src.astype(np.float64) @ (T_true[:3, :3] * 1.002).T + T_true[:3, 3]).astype(np.float32)
        dev_b = icp.IcpDevice(near, tgt, ctx, culled=False)
        ms_b = timed(dev_b.nn, 3)
        dev_b.free()
        dev_c = icp.IcpDevice(near, tgt, ctx, culled=True)
        ms_c = timed(dev_c.nn, 20)
        dev_c.state_reset()
        ms_it = timed(lambda: dev_c.iterate(6), 10) / 6     # as the estimator runs them: six per enqueue (the later five start warm)
        dev_c.free()
        tf = m * m * 8 / ms_b / 1e9
        # the reference's own case (readme.md:25): two partially overlapping 480x640 single views, rigid point-to-plane ICP
        Sy = importlib.import_module("3d_reconstruction_system_amd.synthetic")
        v2 = Sy.two_views(480, 640, yaw_deg=15.0, baseline=(0.35, 0.05, -0.2), depth_noise=0.001, seed=1)
        pa = r3d.unproject(v2["depth_a"], v2["K"], ctx=ctx)
        pb = r3d.unproject(v2["depth_b"], v2["K"], ctx=ctx)
        ca, sa = np.cos(np.deg2rad(5.0)), np.sin(np.deg2rad(5.0))
        E = np.eye(4)
        E[:3, :3] = [[ca, 0, sa], [0, 1, 0], [-sa, 0, ca]]
        E[:3, 3] = (0.06, -0.05, 0.06)
        T0 = E @ v2["T_ab"]
        icp.icp_point_to_plane(pb[:60000], pa, tgt_shape=(480, 640), init=T0, max_iter=2, ctx=ctx)      # warm-up
        plane_walls = []
        for _ in range(4):   # (first full-size call apart, as above)
            t0 = time.perf_counter()
            Tp, infop = icp.icp_point_to_plane(pb, pa, tgt_shape=(480, 640), init=T0, ctx=ctx)
            plane_walls.append((time.perf_counter() - t0) * 1e3)
        plane_ms = sorted(plane_walls[1:])[1]
        devp = icp.PlaneIcpDevice(pb, pa, (480, 640), ctx=ctx)
        devp.move_source(Tp)
        devp.state_reset()
        ms_pit = timed(lambda: devp.iterate(6), 10) / 6
        devp.free()
        plane = {"what": "two 480x640 single views of a room, 15 deg apart, 67 % overlap, depth noise 0.1 %, start 5 deg / 10 cm off",
                 "wall_ms": round(plane_ms, 2), "first_call_ms": round(plane_walls[0], 2), "iterations": infop["iterations"], "iteration_ms": round(ms_pit, 4),
                 "T_error_max_abs": float(np.abs(Tp - v2["T_ab"]).max()), "pairs": infop["pairs"]}
        line = {"metric": "ICP similarity estimation, two 500k-point clouds (C3: s=1.7, 10 deg, |t|=0.5, no initial guess)",
                "value": round(wall_ms, 2), "unit": "ms wall (upload, index builds, coarse + fine stages; median of 3 calls after the first)",
                "first_call_ms": round(first_call_ms, 2), "higher_is_better": False,
                "T_error_max_abs": float(np.abs(T - T_true).max()), "coarse_iterations": info["coarse_iterations"],
                "fine_iterations": info["iterations"], "final_rms": info["rms_history"][-1],
                "fine_iteration_ms": round(ms_it, 4), "culled_nn_ms": round(ms_c, 4), "bruteforce_nn_ms": round(ms_b, 3),
                "point_to_plane_two_views": plane,
                "roofline": {"bound": "valu", "achieved": round(tf, 1), "peak": 157.3, "unit": "TFLOP/s", "frac": round(tf / 157.3, 4),
                             "traffic": None, "kernel": "nn_kernel<4> (brute force, 8 flop/pair)", "kernel_ms": round(ms_b, 3)}}
    elif a.workload == "c5":
        # BASELINE config 5 geometry per GPU: AirSim 1920x1080 f32 depth + RGB, fused cloud carrying colour + voxel insert
        V = importlib.import_module("3d_reconstruction_system_amd.voxelmap")
        F, h5, w5 = max(1, min(a.frames, 100)) if a.frames != FRAMES_PER_GPU else 50, 1080, 1920
        n = F * h5 * w5
        # (poses first, then depth, then colour: a checker can regenerate the first k frames without drawing all F)
        tab = r3d.pose_table(rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10)
        depth = (rng.random((F, h5, w5), dtype=np.float32) * 99.5 + 0.5)
        rgb = rng.integers(0, 256, size=(F, h5, w5, 3), dtype=np.uint8)
        d_depth, d_rgb, d_pose = ctx.alloc(depth.nbytes).upload(depth), ctx.alloc(rgb.nbytes).upload(rgb), ctx.alloc(tab.nbytes).upload(tab)
        del depth, rgb
        d_xyz, d_rgba = ctx.alloc(n * 12), ctx.alloc(n * 4)
        cam = ctx.camera(h5, w5, 960.0, 960.0, 959.5, 539.5)
        ms = timed(lambda: r3d.fuse_frames_rgb_device(ctx, cam, d_depth.ptr, np.float32, F, d_pose.ptr, d_rgb.ptr, d_xyz.ptr,
                                                      np.float32, d_rgba.ptr), max(a.steps // 20, 20))
        vs = V.VoxelSet(0.1, 2 * n, ctx)

        def both():
            vs.clear()
            vs.insert_device(d_xyz.ptr, n)
        ms_clear = timed(vs.clear, 5)
        ms_v = timed(both, 5) - ms_clear
        both()
        st_all = vs.stats()

        def one_launch():   # the cloud and the map from one kernel (r3d_fuse_frames_voxel): the cloud is not read back
            vs.clear()
            r3d.fuse_frames_voxel_device(ctx, cam, d_depth.ptr, np.float32, F, d_pose.ptr, d_rgb.ptr, d_xyz.ptr, d_rgba.ptr, vs)
        ms_one = timed(one_launch, 5) - ms_clear
        one_launch()
        st_one = vs.stats()
        # a checkable digest of the map's voxel half: the occupied set of the first k frames' cloud (the test suite forms the
        # same set with the oracle and compares count, ignored points and two order-independent digests of the codes)
        k_chk = min(F, 3)
        vs.clear()
        vs.insert_device(d_xyz.ptr, k_chk * h5 * w5)
        st_k = vs.stats()
        codes = vs.codes()
        voxel_check = {"frames": k_chk, "points": k_chk * h5 * w5, "voxels": int(codes.shape[0]), "ignored_points": st_k["ignored_points"],
                       "overflow": st_k["overflow"], "codes_xor": int(np.bitwise_xor.reduce(codes)) if codes.size else 0,
                       "codes_sum_mod_2_64": int(np.sum(codes, dtype=np.uint64)) if codes.size else 0,
                       "seed": 1234, "resolution": 0.1}
        bpp = 16 + 7
        gbs = n * bpp / ms / 1e6
        line = {"metric": "Mpoints/s fused RGBD (1920x1080 f32 depth + RGB -> f32 xyz + rgba), %d frames" % F,
                "value": round(n / ms / 1e3, 1), "unit": "Mpoints/s", "voxel_insert_ms": round(ms_v, 3),
                "fuse_plus_voxel_Mpoints_s": round(n / (ms + ms_v) / 1e3, 1), "voxels": st_all["voxels"],
                "one_launch_cloud_and_voxels": {"ms": round(ms_one, 3), "Mpoints_s": round(n / ms_one / 1e3, 1),
                                                "same_counters_as_two_calls": st_one == st_all},
                "voxel_check": voxel_check,
                "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None, "kernel": "fuse_rgb_kernel<f32,pose>",
                             "frac_of_measured_copy": round(gbs / HBM_COPY_GBS, 4),
                             "kernel_ms": round(ms, 5), "algorithmic_bytes_per_launch": n * bpp,
                             "bytes_per_point": "16 (f32 depth in, f32 xyz out) + 7 (rgb in, rgba out)"}}
    else:
        V = importlib.import_module("3d_reconstruction_system_amd.voxelmap")
        F = FRAMES_PER_GPU
        n = F * H * W
        depth = rng.integers(1, 256, size=(F, H, W), dtype=np.uint8)
        tab = r3d.pose_table(rng.normal(size=(F, 4)), rng.normal(size=(F, 3)))
        d_depth, d_pose, d_xyz = ctx.alloc(n).upload(depth), ctx.alloc(tab.nbytes).upload(tab), ctx.alloc(n * 12)
        cam = ctx.camera(H, W, *r3d.REF_INTRINSICS)
        r3d.fuse_frames_device(ctx, cam, d_depth.ptr, np.uint8, F, d_pose.ptr, d_xyz.ptr, np.float32)
        vs = V.VoxelSet(0.1, 2 * n, ctx)

        def both():
            vs.clear()
            vs.insert_device(d_xyz.ptr, n)
        ms_clear = timed(vs.clear, 10)
        per_path = {}
        for label, path in (("cas_lds_set", 1), ("sort_merge", 2), ("auto", 0)):
            ctx.set_tuning("voxel_path", path)
            per_path[label] = {"ms": round(timed(both, 10) - ms_clear, 4), "path_taken": ctx.get_tuning("voxel_last_path")}
        ms = per_path["auto"]["ms"]
        both()
        st = vs.stats()
        # algorithmic bytes of a set insert: every point read once (12 B), every distinct voxel written once (8 B)
        alg = n * 12 + st["voxels"] * 8
        gbs = alg / ms / 1e6
        line = {"metric": "Mpoints/s voxel insert (C2 cloud, 0.1 m, worst case ~1 voxel per point)", "value": round(n / ms / 1e3, 1),
                "unit": "Mpoints/s", "voxels": st["voxels"], "kernel_ms": round(ms, 4), "paths": per_path,
                "table_slots": 1 << int(np.ceil(np.log2(2 * n))),
                "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None, "algorithmic_bytes_per_launch": alg,
                             "kernel": "voxel_keys_kernel + 2 radix passes + voxel_bounds_kernel + voxel_merge_kernel (sort-merge insert: "
                                       "streams 20 + 48 + 8 + 8 B/point + 16 B/table slot; the CAS path is bound by scattered 64-bit "
                                       "atomics at ~19 G/s instead)"}}
    line.setdefault("higher_is_better", True)
    line.update({"n_gpus": 1, "data": "synthetic", "dtype": "f64" if a.workload in ("apply", "c5") else "f32",
                 "config": {"workload": a.workload}})
    print(json.dumps(line), flush=True)
    ctx.close()


def c5_sharded(a):
    """BASELINE config 5's shape over the GPUs of a node, torch-free: `torch.distributed.run --nproc-per-node N bench.py
    --workload c5 --gpus N`.  Every rank owns F frames of 1920x1080 f32 depth + RGB; a step = fuse them with colour (one
    launch), voxelise the rank's shard into its own HBM hash set, unite the sets through the C ABI (r3d_voxelset_union:
    all-gather of the DISTINCT codes only, 8 B/voxel; the 16 B/point of the coloured cloud never leave their GPU).
    Synthetic depth is random, i.e. the worst case of ~1 voxel per point."""
    r3d = importlib.import_module("3d_reconstruction_system_amd")
    V = importlib.import_module("3d_reconstruction_system_amd.voxelmap")
    CM = importlib.import_module("3d_reconstruction_system_amd.comm")
    rank, world = CM.env_rank_world()
    if world != a.gpus:
        sys.exit("WORLD_SIZE=%d does not match --gpus %d" % (world, a.gpus))
    ctx = r3d.Context(CM.env_local_device())
    comm = CM.Comm.from_env(ctx)
    F = max(1, min(a.frames, 250)) if a.frames != FRAMES_PER_GPU else 50
    h5, w5 = 1080, 1920
    n = F * h5 * w5
    rng = np.random.default_rng(5 + rank)
    depth = rng.random((F, h5, w5), dtype=np.float32) * 99.5 + 0.5
    rgb = rng.integers(0, 256, size=(F, h5, w5, 3), dtype=np.uint8)
    tab = r3d.pose_table(rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10)
    d_depth, d_rgb, d_pose = ctx.alloc(depth.nbytes).upload(depth), ctx.alloc(rgb.nbytes).upload(rgb), ctx.alloc(tab.nbytes).upload(tab)
    del depth, rgb
    d_xyz, d_rgba = ctx.alloc(n * 12), ctx.alloc(n * 4)
    cam = ctx.camera(h5, w5, 960.0, 960.0, 959.5, 539.5)
    vs = V.VoxelSet(0.1, 2 * n * world, ctx)
    d_t = ctx.alloc(8 * (world + 1))

    def step():
        r3d.fuse_frames_rgb_device(ctx, cam, d_depth.ptr, np.float32, F, d_pose.ptr, d_rgb.ptr, d_xyz.ptr, np.float32, d_rgba.ptr)
        vs.clear()
        vs.insert_device(d_xyz.ptr, n)
        vs.union_across(comm)

    def max_over_ranks(seconds):
        mine = np.array([seconds])
        ctx.lib.r3d_memcpy_h2d(ctx.handle, d_t.ptr + 8 * world, mine.ctypes.data, 8)
        comm.allgather(d_t.ptr + 8 * world, [8] * world, d_t.ptr)
        return float(d_t.download(np.float64, world).max())

    steps, warm = max(1, min(a.steps, 50)), max(1, min(a.warmup, 5))
    for _ in range(warm):
        step()
    comm.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    comm.barrier()
    sec = max_over_ranks(time.perf_counter() - t0)
    st = vs.stats()
    ms_fuse = []
    for _ in range(5):
        ctx.timer_start()
        r3d.fuse_frames_rgb_device(ctx, cam, d_depth.ptr, np.float32, F, d_pose.ptr, d_rgb.ptr, d_xyz.ptr, np.float32, d_rgba.ptr)
        ms_fuse.append(ctx.timer_stop())
    ms = sorted(ms_fuse)[2]
    if rank == 0:
        gbs = n * 23 / ms / 1e6
        print(json.dumps({
            "metric": "Mpoints/s fused RGBD + voxel map (1920x1080 f32 depth + RGB, %d frames per GPU, one map)" % F,
            "value": round(world * n * steps / sec / 1e6, 1), "unit": "Mpoints/s", "n_gpus": world, "steps": steps, "warmup": warm,
            "ms_per_step": round(sec / steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic (random depth: ~1 voxel per point, the worst case for the map)",
            "config": {"workload": "C5: fuse with colour + voxel insert + union of the ranks' sets", "frames_per_gpu": F,
                       "points_per_step": world * n, "parallelism": "frames sharded, %d rank(s), one process per GPU, "
                                                                    "r3d_comm (%s)" % (world, comm.rccl_origin())},
            "union_voxels": st["voxels"], "union_overflow": st["overflow"],
            "fabric_bytes_in_per_gpu": 8 * st["voxels"] * (world - 1) // max(world, 1),
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None, "kernel": "fuse_rgb_kernel<f32,pose>",
                         "kernel_ms": round(ms, 5), "algorithmic_bytes_per_launch": n * 23}}), flush=True)
    comm.barrier()
    comm.close()
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--frames", type=int, default=FRAMES_PER_GPU, help="frames per GPU (100 = config C2)")
    ap.add_argument("--assemble", default="auto",
                    choices=["auto", "inputs", "inputs_direct", "inputs_overlap", "outputs", "outputs_direct", "none"],
                    help="N>1: how the headline step assembles the fused world cloud on every rank.  Every strategy is "
                         "timed before the headline region and printed under 'assemble'; 'auto' (default) then runs the "
                         "fastest one that leaves the whole cloud on every rank.  'outputs' = fuse own frames, all-gather "
                         "xyz (12 B/point over xGMI, north_star's wording); 'inputs' = all-gather depth+poses (1 B/point) "
                         "then fuse all frames locally (same bits); '*_direct' = grouped send/recv per peer instead of "
                         "ncclAllGather; 'none' = shards stay resident")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-regimes", action="store_true",
                    help="N=1: skip the child process that measures the headline launch outside the bench loop's regime (rotating "
                         "rasters = inputs not in the Infinity Cache, plain / staged; right after an H2D upload; config 4's 1000 "
                         "frames as one launch).  It runs by default and lands in roofline.cold_inputs; rocprofv3 --pmc passes use "
                         "this flag (a profiled process must not start another program)")
    ap.add_argument("--no-end-to-end", action="store_true",
                    help="N=1: skip the end_to_end object (two more child processes: host buffers over PCIe through "
                         "r3d_fuse_frames_host, and the camera_to_world.py drop-in on 100 PNG files)")
    ap.add_argument("--out-dtype", default="float32", choices=["float32", "float64"])
    ap.add_argument("--workload", default="fuse", choices=["fuse", "apply", "icp", "voxel", "c5", "regimes", "e2e"],
                    help="fuse (default, the headline C2 line); the others print one JSON line for a secondary kernel on "
                         "one GPU: apply = 4x4 apply on the C2 cloud, icp = C3 (two 500k clouds, SURVEY recipe), voxel = occupancy insert, "
                         "c5 = config 5 geometry (1080p f32 RGBD, colour carried, + voxel insert)")
    a = ap.parse_args()
    if a.workload == "c5" and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        return c5_sharded(a)
    if a.workload == "regimes":
        return regimes(a)
    if a.workload == "e2e":
        return e2e_host(a)
    if a.workload != "fuse":
        return secondary(a)

    # N = 1: the other regimes of the headline launch, measured by a child process BEFORE this one touches the GPU
    cold = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and a.gpus == 1 and not a.no_regimes and a.out_dtype == "float32" \
            and a.frames == FRAMES_PER_GPU:
        cold = regimes_in_child()
    e2e = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and a.gpus == 1 and not a.no_end_to_end and a.out_dtype == "float32" \
            and a.frames == FRAMES_PER_GPU:
        e2e = end_to_end_children()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (a.gpus, a.gpus))
        sys.exit("WORLD_SIZE=%d does not match --gpus %d" % (world, a.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no GPU is visible and there is no CPU fallback")
    # one rank per GPU; R3D_DIST_BACKEND=gloo lets several ranks share one GPU for a functional rehearsal
    backend = os.environ.get("R3D_DIST_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # R3D_BENCH_FORCE_COLLECTIVES=1 runs the N>1 code path (process group, assembly step) even with one rank:
    # a rehearsal of the RCCL calls on a single-GPU box
    use_dist = world > 1 or os.environ.get("R3D_BENCH_FORCE_COLLECTIVES", "0") not in ("", "0")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    r3d = importlib.import_module("3d_reconstruction_system_amd")
    D = importlib.import_module("3d_reconstruction_system_amd.dist")
    stream = torch.cuda.current_stream(dev)
    ctx = r3d.Context(dev_index, stream=stream.cuda_stream)
    cam = ctx.camera(H, W, *r3d.REF_INTRINSICS)

    # the exchange step's transports are created further down, AFTER the shards-stay-resident job has been measured and
    # with the watchdog armed: a communicator bring-up that stalls must not cost the line either
    transport, transport_note = None, ""
    side = side_transport = ctx2 = None

    # synthetic job: rank r owns frames [r*F, (r+1)*F) of a world*F-frame sequence
    F = a.frames
    rng = np.random.default_rng(1234 + rank)
    depth = torch.from_numpy(rng.integers(1, 256, size=(F, H, W), dtype=np.uint8)).to(dev)
    table = torch.from_numpy(r3d.pose_table(rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10)).to(dev)
    out_np = np.float32 if a.out_dtype == "float32" else np.float64
    out_t = torch.float32 if a.out_dtype == "float32" else torch.float64
    xyz_bytes = 12 if a.out_dtype == "float32" else 24
    n_local = F * H * W
    full = depth_all = pose_all = None
    if use_dist:
        full = torch.empty((world * n_local, 3), dtype=out_t, device=dev)
        shard = full[rank * n_local:(rank + 1) * n_local]       # fuse straight into this rank's slot of the world cloud
        depth_all = torch.empty((world * F, H, W), dtype=torch.uint8, device=dev)
        pose_all = torch.empty((world * F, 12), dtype=torch.float64, device=dev)
    else:
        shard = torch.empty((n_local, 3), dtype=out_t, device=dev)
    frames_pr, points_pr = [F] * world, [n_local] * world

    def fuse():                                                  # this rank's frames only
        r3d.fuse_frames_device(ctx, cam, depth.data_ptr(), np.uint8, F, table.data_ptr(), shard.data_ptr(), out_np)

    def fuse_all():                                              # every rank's frames, from the gathered inputs
        r3d.fuse_frames_device(ctx, cam, depth_all.data_ptr(), np.uint8, world * F, pose_all.data_ptr(), full.data_ptr(),
                               out_np)

    def with_algo(algo):
        if isinstance(transport, D.R3dTransport):
            transport.algo = algo

    def make_step(m):
        """none: shards stay resident.  outputs: fuse own frames, all-gather xyz (north_star, 12 B/point over xGMI).
        inputs: all-gather rasters + poses (1 B/point), fuse every frame locally.  *_direct: the same exchange as one
        grouped send/recv per peer instead of ncclAllGather (r3d_comm only)."""
        algo = 2 if m.endswith("_direct") else 0
        if m == "none":
            return fuse
        if m.startswith("outputs"):
            def step_outputs():
                fuse()
                with_algo(algo)
                transport.allgather_rows(shard, points_pr, out=full)
            return step_outputs

        if m == "inputs_overlap":
            # the 'inputs' strategy as a pipeline: the rasters travel in OVERLAP_CHUNKS slices on a side stream while the
            # main stream fuses the slices that have landed (one launch per rank's slice, straight into its place in the
            # rank-major world cloud) -- same bits, step time ~ max(exchange, fuse) instead of their sum
            C_, fc = OVERLAP_CHUNKS, F // OVERLAP_CHUNKS
            per = H * W
            d_chunks = [torch.empty((world * fc, H, W), dtype=torch.uint8, device=dev) for _ in range(C_)]
            p_chunks = [torch.empty((world * fc, 12), dtype=torch.float64, device=dev) for _ in range(C_)]
            events = [torch.cuda.Event() for _ in range(C_)]

            def step_overlap():
                side.wait_stream(stream)                      # inputs are ready / last step's fuses have read the chunks
                with torch.cuda.stream(side):
                    for c in range(C_):
                        side_transport.allgather_rows(depth[c * fc:(c + 1) * fc], [fc] * world, out=d_chunks[c])
                        side_transport.allgather_rows(table[c * fc:(c + 1) * fc], [fc] * world, out=p_chunks[c])
                        events[c].record(side)
                for c in range(C_):
                    stream.wait_event(events[c])
                    if not isinstance(side_transport, D.R3dTransport):
                        ctx.inputs_fresh()
                    for r in range(world):
                        r3d.fuse_frames_device(ctx, cam, d_chunks[c][r * fc:].data_ptr(), np.uint8, fc,
                                               p_chunks[c][r * fc:].data_ptr(),
                                               full[(r * F + c * fc) * per:].data_ptr(), out_np)
            return step_overlap

        def step_inputs():
            with_algo(algo)
            transport.allgather_rows(depth, frames_pr, out=depth_all)
            transport.allgather_rows(table, frames_pr, out=pose_all)
            if not isinstance(transport, D.R3dTransport):
                ctx.inputs_fresh()        # torch's collective wrote the rasters: a foreign producer (r3d_comm tracks its own)
            fuse_all()
        return step_inputs

    def fence():
        if use_dist:
            if backend == "nccl":
                dist.barrier(device_ids=[dev_index])
            else:
                dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(seconds):
        if not use_dist:
            return seconds
        tm = torch.tensor([seconds], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        return float(tm.item())

    # kernel-only duration of the dominant kernel (this rank's fused launch), measured BEFORE the timed region and
    # independently of --steps: it doubles as the clock ramp, so that a short --steps run sees a warm GPU
    bytes_per_launch = n_local * (1 + 3 * (4 if a.out_dtype == "float32" else 8))
    kernel_ms, kernel_mean_ms, kernel_n = kernel_duration_ms(torch, stream, fuse, min_launches=4000)
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9

    # which kernel the library dispatched for this launch (r3d_fuse.hip picks by output type)
    kernel_label = "fuse_lane_kernel<u8,f32,pose>" if a.out_dtype == "float32" else "fuse_pair_kernel<u8,f64,pose>"

    def make_line(mode, elapsed, gpu_ms_per_step, kernel_region_ms):
        """kernel_region_ms: duration of ONE fused launch of this rank inside the timed region -- HIP events on the launch
        stream around the K steps / K for the steps that consist of that launch alone; for the assembling strategies (whose
        step holds collectives too) the sustained figure measured before the region."""
        total_pts = world * n_local * a.steps
        k_ms = kernel_region_ms if kernel_region_ms else kernel_ms
        ach = bytes_per_launch / (k_ms * 1e-3) / 1e9
        traffic, traffic_source = pmc_traffic(F, a.out_dtype)
        sweeps = region.get("sweeps")            # staging sweeps the library enqueued inside the timed region (counted, not guessed)
        fused_only_ms = k_ms if not sweeps else kernel_ms
        kernels = [{"name": kernel_label, "ms": round(fused_only_ms, 5), "launches_per_step": 1}]
        if sweeps:
            kernels.insert(0, {"name": "cache_touch_kernel", "ms": round(max(k_ms - fused_only_ms, 0.0), 5),
                               "launches_per_step": round(sweeps / max(a.steps, 1), 3)})
        line = {
            "metric": "Mpoints/s fused (1280x384 depth, N frames)",
            "value": round(total_pts / elapsed / 1e6, 2),
            "unit": "Mpoints/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "C2: 1280x384 u8 depth x %d frames per GPU -> %s xyz, fused unproject+SE(3); every step "
                                   "re-reads the same 49 MB raster, which is therefore served by the 256 MiB Infinity Cache "
                                   "from the 2nd launch on while the xyz stream goes to HBM (fresh-raster regimes: "
                                   "roofline.cold_inputs)" % (F, a.out_dtype),
                       "frames_per_gpu": F, "points_per_step": world * n_local,
                       "step": {"none": "1 fused launch over this rank's frames (shards stay resident)",
                                "outputs": "1 fused launch + all-gather of xyz shards (12 B/point over xGMI)",
                                "inputs": "all-gather of depth+poses (1 B/point over xGMI) + 1 fused launch over "
                                          "all ranks' frames on every rank (replicated compute: each GPU writes the "
                                          "whole cloud into its own HBM)"}[mode.replace("_direct", "").replace("_overlap", "")]
                               + (" [grouped send/recv per peer]" if mode.endswith("_direct") else "")
                               + (" [pipelined: %d slices gathered on a side stream while the landed ones are fused]"
                                  % OVERLAP_CHUNKS if mode.endswith("_overlap") else ""),
                       "assemble": mode,
                       "assemble_choice": a.assemble,
                       "parallelism": "frames sharded, %d rank(s), one process per GPU" % world},
            "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "frac_of_measured_copy": round(ach / HBM_COPY_GBS, 4),
                         "kernel": kernel_label,
                         "kernel_ms": round(k_ms, 5),
                         # every kernel the library launched per step of the timed region, each with its own duration (they
                         # sum to kernel_ms): since round 4 a raster the previous launch has just read is not staged again, so
                         # the steady-state step is the fused kernel alone (round 3: + a 7.8 us sweep on every launch)
                         "kernels": kernels,
                         "staging_sweeps_in_timed_region": sweeps,
                         "frac_kernel_only": round(bytes_per_launch / (fused_only_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "timing": ("HIP events on the launch stream around the %d launches of the timed region / %d "
                                    "(same region as ms_per_step, which is the wall clock between the two fences)"
                                    % (a.steps, a.steps)) if kernel_region_ms else
                                   "sustained median (below): this strategy's step also holds collectives",
                         # the same launch sustained: median over >= 4000 launches in groups of 100 between HIP events,
                         # after 50 untimed launches, before the timed region (also the clock ramp for a short --steps run)
                         "sustained": {"kernel_ms": round(kernel_ms, 5), "kernel_mean_ms": round(kernel_mean_ms, 5),
                                       "launches": kernel_n, "frac": round(achieved / HBM_PEAK_GBS, 4)},
                         "kernel_ms_over_ms_per_step": round(k_ms / (elapsed / a.steps * 1e3), 4),
                         "inputs": "raster in the Infinity Cache (see config.workload); cold_inputs = the same launch on rasters "
                                   "that are not, measured in this run by a child process",
                         "cold_inputs": cold},
            "gpu_ms_per_step": round(gpu_ms_per_step, 5),
            "kernel_only_Mpoints_s_per_gpu": round(n_local / k_ms / 1e3, 1),
        }
        if use_dist:
            line["transport"] = transport_note
            line["assemble"] = assemble
            # the quantity that scales with N: every rank fuses its own frames and keeps its shard (for the voxel / ICP stages)
            line["value_shards_resident"] = assemble.get("none", {}).get("Mpoints_s")
            line["scaling_note"] = ("weak: every rank owns %d frames.  A strategy that leaves the WHOLE cloud on EVERY rank "
                                    "makes each GPU write world x %.0f MB into its own HBM, so its whole-job rate (`value`) cannot "
                                    "exceed one GPU's kernel rate; `value_shards_resident` (assemble.none: shards stay resident "
                                    "for the voxel / ICP stages) is the rate that scales" % (F, n_local * xyz_bytes / 1e6))
        else:
            line["value_shards_resident"] = line["value"]
        return line

    def headline(step):
        """--warmup untimed steps, then EXACTLY --steps timed ones between two fences; seconds = max over ranks."""
        for _ in range(a.warmup):
            step()
        fence()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0 = ctx.get_tuning("fuse_sweeps")
        t0 = time.perf_counter()
        ev0.record(stream)
        for _ in range(a.steps):
            step()
        ev1.record(stream)
        fence()
        sec = time.perf_counter() - t0
        region["sweeps"] = ctx.get_tuning("fuse_sweeps") - s0
        return max_over_ranks(sec), ev0.elapsed_time(ev1) / max(a.steps, 1)   # this rank's stream: first start to last end

    # N > 1: every assembly strategy, timed briefly BEFORE the headline region (all of them go on the line); the
    # headline step is --assemble, by default the fastest strategy that leaves the whole cloud on every rank.
    # The exchange code has only ever met RCCL with one rank (no multi-GPU node was available to the build), so the
    # survey runs under a watchdog: the shards-stay-resident job is measured FIRST by the full contract, and if a
    # strategy then stalls for WATCHDOG_S seconds every rank prints nothing more / rank 0 prints THAT line, flagged.
    assemble = {}
    mode = "none"
    fallback = {}
    region = {}
    beat = {"t": time.monotonic(), "what": "start", "armed": False}

    def watchdog():
        while beat["armed"]:
            time.sleep(1.0)
            if beat["armed"] and time.monotonic() - beat["t"] > WATCHDOG_S:
                if rank == 0 and fallback:
                    line = make_line("none", fallback["elapsed"], fallback["gpu"], fallback["gpu"])
                    line["watchdog"] = "assembly strategy '%s' made no progress for %d s; this line is the " \
                                       "shards-stay-resident job measured before the survey" % (beat["what"], WATCHDOG_S)
                    print(json.dumps(line), flush=True)
                os._exit(4 if fallback else 3)      # a wedged exchange is a failed multi-GPU run even though a line went out

    if use_dist:
        elapsed_none, gpu_none = headline(make_step("none"))
        fallback["elapsed"], fallback["gpu"] = elapsed_none, gpu_none
        assemble["none"] = {"ms_per_step": round(elapsed_none / a.steps * 1e3, 4),
                            "Mpoints_s": round(world * n_local * a.steps / elapsed_none / 1e6, 1), "fabric_bytes_in_per_gpu": 0}
        import threading
        beat.update(t=time.monotonic(), armed=True, what="communicator set-up")
        threading.Thread(target=watchdog, daemon=True).start()
        # the exchange step: the library's own RCCL communicator (C ABI, r3d_comm_*) when it comes up on every rank,
        # torch.distributed otherwise (always for gloo rehearsals)
        if use_dist:
            want = os.environ.get("R3D_BENCH_TRANSPORT", "r3d" if backend == "nccl" else "torch")
            if want == "r3d":
                try:
                    CM = importlib.import_module("3d_reconstruction_system_amd.comm")
                    box = [CM.Comm.unique_id() if rank == 0 else None]
                    dist.broadcast_object_list(box, src=0)
                    transport = D.R3dTransport(CM.Comm(ctx, box[0], rank, world))
                    transport_note = "r3d_comm over RCCL (%s)" % transport.comm.rccl_origin()
                except Exception as e:     # e.g. no librccl to dlopen: every rank takes the same way out
                    transport, transport_note = None, "r3d_comm unavailable (%s: %s); " % (type(e).__name__, str(e)[:120])
            if transport is None:
                transport = D.TorchTransport()
                transport_note += "torch.distributed (%s)" % backend
        # a second exchange channel on a side stream, for the pipelined strategy (gather chunk c+1 while chunk c is fused)
        if use_dist and a.frames % OVERLAP_CHUNKS == 0:
            try:
                side = torch.cuda.Stream(dev)
                if isinstance(transport, D.R3dTransport):
                    CM = importlib.import_module("3d_reconstruction_system_amd.comm")
                    ctx2 = r3d.Context(dev_index, stream=side.cuda_stream)
                    box = [CM.Comm.unique_id() if rank == 0 else None]
                    dist.broadcast_object_list(box, src=0)
                    side_transport = D.R3dTransport(CM.Comm(ctx2, box[0], rank, world))
                else:
                    side_transport = transport        # torch collectives follow torch's current stream
            except Exception:
                side = side_transport = None

        modes = ["outputs", "inputs"]
        if isinstance(transport, D.R3dTransport):
            modes += ["outputs_direct", "inputs_direct"]
        if side_transport is not None:
            modes.append("inputs_overlap")
        for m in modes:
            beat.update(t=time.monotonic(), what=m)
            try:   # a side measurement must never cost the headline line
                st = make_step(m)
                for _ in range(3):
                    st()
                fence()
                beat["t"] = time.monotonic()
                t1 = time.perf_counter()
                for _ in range(10):
                    st()
                fence()
                sec = max_over_ranks((time.perf_counter() - t1) / 10)
                beat["t"] = time.monotonic()
                fabric_in = (world - 1) * (n_local * xyz_bytes if m.startswith("outputs") else F * (H * W + 96))
                if m == "inputs_overlap":    # same bits as 'inputs': checked here once, cheaply, on a strided sample
                    probe = full[::997].clone()
                    make_step("inputs")()
                    if not torch.equal(probe, full[::997]):
                        raise RuntimeError("pipelined assembly differs from the plain one")
                entry = {"ms_per_step": round(sec * 1e3, 4), "Mpoints_s": round(world * n_local / sec / 1e6, 1),
                         "fabric_bytes_in_per_gpu": fabric_in}
                if fabric_in and world > 1:
                    gbs = fabric_in / sec / 1e9          # whole step time, compute included: a lower bound on the links
                    entry["xgmi_GBps_in_per_gpu"] = round(gbs, 1)
                    entry["xgmi_GBps_per_link"] = round(gbs / (world - 1), 1)
                    entry["frac_of_link_peak"] = round(gbs / (world - 1) / XGMI_LINK_GBS, 4)
                assemble[m] = entry
            except Exception as e:  # pragma: no cover
                assemble[m] = {"failed": "%s: %s" % (type(e).__name__, str(e)[:100])}
        ok = {m: v["ms_per_step"] for m, v in assemble.items() if "ms_per_step" in v and m != "none"}
        if a.assemble == "auto":
            mode = min(ok, key=ok.get) if ok else "none"
        else:
            mode = a.assemble if (a.assemble in ok or a.assemble == "none") else "none"
        beat.update(t=time.monotonic(), what="headline (%s)" % mode)
    if use_dist and mode == "none":
        elapsed, gpu_ms_per_step = elapsed_none, gpu_none        # already measured by the full contract
    else:
        elapsed, gpu_ms_per_step = headline(make_step(mode))
    beat["armed"] = False

    if rank == 0:
        line = make_line(mode, elapsed, gpu_ms_per_step, gpu_ms_per_step if mode == "none" else None)
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(3)
        if e2e is not None:
            line["end_to_end"] = e2e
            ref = line.get("cpu_baseline", {}).get("files_to_files")
            drop = e2e.get("dropin_camera_to_world", {})
            if ref and "s_per_frame" in drop:      # the reference's per-frame path (1 core, loops + PLY) beside the drop-in's
                drop["cpu_reference_s_per_frame"] = ref["s_per_frame"]
                drop["cpu_reference_kind"] = "port: oracle/fusion_ref.py loops incl. genply, 1 core, extrapolated per frame"
        print(json.dumps(line), flush=True)
    if use_dist:
        fence()
        dist.destroy_process_group()
    if ctx2 is not None:
        ctx2.close()
    ctx.close()


if __name__ == "__main__":
    main()
