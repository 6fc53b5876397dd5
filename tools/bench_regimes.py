"""bench.py --workload regimes: the headline launch in the regimes the headline step is NOT in (run as a child process)."""
import importlib
import json
import os
import sys
import tempfile
import time

import numpy as np

from bench_common import BYTES_PER_POINT, FRAMES_PER_GPU, H, HBM_COPY_GBS, HBM_PEAK_GBS, ROOT, W, cpu_model, run_child  # noqa: F401


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
        out[key + "_ms"], out[key + "_frac"], out[key + "_Mpoints_s"] = round(ms, 5), frac(ms), round(n / ms / 1e3, 1)
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

