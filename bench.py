#!/usr/bin/env python3
"""Headline benchmark: fused depth -> world point-cloud throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]            (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], "C2"): KITTI-sized 1280x384 uint8 depth, 100 frames per GPU, synthetic (seed 1234),
resident in HBM.  One STEP = one pass of the hot path over one batch of FRESH frames = the library's fused unproject + SE(3)
call over 100 frames (49,152,000 points -> f32 xyz).  Every frame of a pipeline is read once (camera_to_world.py:149-172), so
the step rotates through 16 resident rasters (786 MB, three Infinity Caches' worth): no launch finds its raster in the
256 MiB Infinity Cache, and the library stages it there with a read-only sweep before the fused kernel (both are in the step
and in every duration below).  The regime rounds 1-4 quoted -- ONE raster re-read every step, i.e. inputs served by the
Infinity Cache -- is still measured and reported beside it (`value_cached_inputs`, `roofline.frac_cached_inputs`).

For N > 1 the step additionally assembles the fused world cloud on every rank (tools/bench_assemble.py: every strategy is
timed, checked bit for bit and printed under "assemble"; the headline step is the fastest).  Weak scaling: every rank owns
100 frames; `value` = UNIQUE fused points of all ranks / max-over-ranks time.

Extra objects on the JSON line:
  roofline     -- the step against the HBM roof.  `frac` = algorithmic bytes (13 B/point) / sustained median step duration
                  (HIP events on the launch stream, >= 4000 launches, fresh rasters, sweep + fused kernel) / 8 TB/s;
                  frac_timed_region = the same over the K timed steps; frac_kernel_only = the fused kernel alone (events
                  between sweep and kernel); frac_cached_inputs / frac_after_h2d / frac_hbm_side: see make_line.
  cpu_baseline -- the loop-faithful CPU restatement of the reference path (oracle/, test infrastructure; rank 0, N=1
                  only) timed on a bounded sample, 1 core; with and without the PLY writer.
  end_to_end   -- N=1 only, child processes (tools/bench_e2e.py): pinned host rasters -> r3d_fuse_frames_host -> pinned
                  host xyz (PCIe both ways), and the camera_to_world.py drop-in on 100 PNG files.
Secondary workloads (`--workload apply|icp|voxel|c5|regimes|e2e`) live in tools/bench_*.py.
"""
import argparse
import importlib
import json
import os
import sys
import tempfile
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_common import (BYTES_PER_POINT, FRAMES_PER_GPU, H, HBM_COPY_GBS, HBM_PEAK_GBS, RASTER_COPIES, W, cpu_model,  # noqa: E402
                          run_child)


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
    per_frame_ply = dt / sample_frames + dt_ply
    return {"value": round(pts / dt / 1e6, 5), "unit": "Mpoints/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
            "sample": "%d frame(s) of 1280x384 u8 (%d points), per-point Python loops + txt round trip as "
                      "camera_to_world.py:67-105, no PLY; %.1f s" % (sample_frames, pts, dt),
            "files_to_files": {"s_per_frame": round(per_frame_ply, 3), "Mpoints_s": round(H * W / per_frame_ply / 1e6, 5),
                               "what": "the same loops PLUS the ASCII PLY writer (genply, camera_to_world.py:112-134; %.2f s for "
                                       "one frame's %d points): the reference's whole per-frame path, files to files"
                                       % (dt_ply, H * W)},
            "vectorised_numpy_fp64_Mpoints_s": round(pts / dt_vec / 1e6, 3),
            "host_cpus": os.cpu_count()}


def pmc_traffic(frames, out_dtype, depth="u8", regime="fresh"):
    """HBM bytes per step from the committed rocprofv3 PMC summary (separate --pmc passes over this same command).
    Counters cannot be read from inside the process, so this is a RECORDED figure: it is emitted only when this run's
    step is the one that was profiled (same frames, types and input regime), otherwise null."""
    p = os.path.join(ROOT, "profiles", "pmc_fuse_latest.json")
    try:
        with open(p) as f:
            rec = json.load(f)
    except Exception:
        return None, None
    cfg = rec.get("config", {})
    if (cfg.get("frames"), cfg.get("out_dtype"), cfg.get("depth"), cfg.get("inputs", "cached")) != (frames, out_dtype, depth, regime):
        return None, None
    import hashlib
    with open(p, "rb") as f:
        sha = hashlib.sha256(f.read()).hexdigest()[:12]
    return rec.get("hbm_bytes_per_launch"), "recorded: profiles/pmc_fuse_latest.json @%s (%s)" % (sha, rec.get("collected", "rocprofv3 --pmc "
                                                                                                 "passes over this command"))


def sustained_ms(torch, stream, launch, min_launches=4000, min_ms=100.0, warm=50):
    """Duration of one step of the dominant kernel(s), independent of --steps: after `warm` untimed launches, at least
    `min_launches` launches (and `min_ms` of GPU time) in groups of 100 between HIP events on the launch stream, back to back
    exactly as in the timed region.  Returns (median_ms, mean_ms, n).  Doubles as the clock ramp for a short --steps run."""
    for _ in range(warm):
        launch()
    torch.cuda.synchronize()
    per, durations, total = 100, [], 0.0
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


def kernel_only_ms(torch, stream, ctx, sweep, fuse_unstaged, groups=5, per=60):
    """The fused kernel ALONE in the fresh-input regime: sweep (the library's own staging kernel, called explicitly), event,
    fused launch with the automatic staging off, event -- `per` such pairs enqueued back to back, one sync per group, the
    elapsed times of the pairs averaged; median over the groups.  The events sit between the two kernels of every step, so
    this carries their overhead: an upper bound on the kernel's duration (rocprofv3's per-kernel average is the tighter one)."""
    ctx.set_tuning("fuse_prefetch", 1)
    try:
        means = []
        for g in range(groups + 1):
            evs = []
            for _ in range(per):
                sweep()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                fuse_unstaged()
                e1.record(stream)
                evs.append((e0, e1))
            evs[-1][1].synchronize()
            if g:
                means.append(sum(e0.elapsed_time(e1) for e0, e1 in evs) / per)
        return sorted(means)[len(means) // 2]
    finally:
        ctx.set_tuning("fuse_prefetch", 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--frames", type=int, default=FRAMES_PER_GPU, help="frames per GPU (100 = config C2)")
    ap.add_argument("--assemble", default="auto",
                    choices=["auto", "inputs", "inputs_direct", "inputs_overlap", "outputs", "outputs_direct", "none"],
                    help="N>1: how the headline step assembles the fused world cloud on every rank (tools/bench_assemble.py); "
                         "'auto' (default) runs the fastest strategy that leaves the whole cloud on every rank")
    ap.add_argument("--inputs", default="fresh", choices=["fresh", "cached"],
                    help="fresh (default): the step rotates through %d resident rasters, none of which is in the Infinity Cache when "
                         "its launch starts; cached: ONE raster re-read every step (rounds 1-4's headline regime)" % RASTER_COPIES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-regimes", action="store_true",
                    help="N=1: skip the child process that measures the launch in the other input regimes (tools/bench_regimes.py); "
                         "rocprofv3 --pmc passes use this flag (a profiled process must not start another program)")
    ap.add_argument("--no-end-to-end", action="store_true", help="N=1: skip the end_to_end object (tools/bench_e2e.py)")
    ap.add_argument("--out-dtype", default="float32", choices=["float32", "float64"])
    ap.add_argument("--workload", default="fuse", choices=["fuse", "apply", "icp", "voxel", "c5", "regimes", "e2e"],
                    help="fuse (default, the headline C2 line); the others print one JSON line for a secondary kernel on one GPU "
                         "(tools/bench_secondary.py, bench_regimes.py, bench_e2e.py)")
    a = ap.parse_args()
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.workload in ("apply", "icp", "voxel", "c5"):
        S = importlib.import_module("bench_secondary")
        return S.c5_sharded(a) if (a.workload == "c5" and env_world > 1) else S.secondary(a)
    if a.workload == "regimes":
        return importlib.import_module("bench_regimes").regimes(a)
    if a.workload == "e2e":
        return importlib.import_module("bench_e2e").e2e_host(a)

    # N = 1: the other regimes of the launch and the end-to-end legs, by child processes BEFORE this one touches the GPU
    default_job = env_world == 1 and a.gpus == 1 and a.out_dtype == "float32" and a.frames == FRAMES_PER_GPU
    other = run_child("regimes") if default_job and not a.no_regimes else None
    e2e = importlib.import_module("bench_e2e").end_to_end_children() if default_job and not a.no_end_to_end else None
    A = importlib.import_module("bench_assemble")
    rccl_log_dir = A.rccl_debug_env(max(env_world, a.gpus))

    import torch
    import torch.distributed as dist

    world, rank, local_rank = env_world, int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (a.gpus, a.gpus))
        sys.exit("WORLD_SIZE=%d does not match --gpus %d" % (world, a.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no GPU is visible and there is no CPU fallback")
    # one rank per GPU; R3D_DIST_BACKEND=gloo lets several ranks share one GPU for a functional rehearsal
    backend = os.environ.get("R3D_DIST_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # R3D_BENCH_FORCE_COLLECTIVES=1 runs the N>1 code path even with one rank: a rehearsal of the RCCL calls on a 1-GPU box
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
    L = importlib.import_module("3d_reconstruction_system_amd._lib")
    stream = torch.cuda.current_stream(dev)
    ctx = r3d.Context(dev_index, stream=stream.cuda_stream)
    cam = ctx.camera(H, W, *r3d.REF_INTRINSICS)

    # synthetic job: rank r owns frames [r*F, (r+1)*F) of a world*F-frame sequence; `copies` distinct batches of them rotate
    F = a.frames
    rng = np.random.default_rng(1234 + rank)
    base = torch.from_numpy(rng.integers(1, 256, size=(F, H, W), dtype=np.uint8)).to(dev)
    n_copies = RASTER_COPIES if a.inputs == "fresh" else 1
    rasters = [base] + [base + (17 * k) for k in range(1, n_copies)]           # u8 wrap-around: another batch of frames
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
    turn = {"i": 0}

    def next_raster():
        d = rasters[turn["i"] % n_copies]
        turn["i"] += 1
        return d

    def fuse():                                                  # this rank's frames only, the next batch of them
        r3d.fuse_frames_device(ctx, cam, next_raster().data_ptr(), np.uint8, F, table.data_ptr(), shard.data_ptr(), out_np)

    def fuse_all():                                              # every rank's frames, from the gathered inputs
        r3d.fuse_frames_device(ctx, cam, depth_all.data_ptr(), np.uint8, world * F, pose_all.data_ptr(), full.data_ptr(), out_np)

    def fence():
        if use_dist:
            dist.barrier(device_ids=[dev_index]) if backend == "nccl" else dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(seconds):
        if not use_dist:
            return seconds
        tm = torch.tensor([seconds], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        return float(tm.item())

    # the step's duration sustained (also the clock ramp), the fused kernel alone, and the cached-input regime beside them
    bytes_per_launch = n_local * (1 + 3 * (4 if a.out_dtype == "float32" else 8))
    step_ms, step_mean_ms, step_n = sustained_ms(torch, stream, fuse)
    k_only_ms = kernel_only_ms(torch, stream, ctx, lambda: L.check(ctx.lib.r3d_cache_prefetch(ctx.handle, rasters[turn["i"] % n_copies].data_ptr(), n_local)), fuse)
    cached_ms, _, cached_n = sustained_ms(torch, stream, lambda: r3d.fuse_frames_device(
        ctx, cam, base.data_ptr(), np.uint8, F, table.data_ptr(), shard.data_ptr(), out_np), min_launches=1000, min_ms=50.0)
    kernel_label = "fuse_lane_kernel<u8,f32,pose>" if a.out_dtype == "float32" else "fuse_pair_kernel<u8,f64,pose>"

    def frac_of(ms, nbytes=bytes_per_launch):
        return round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)

    region = {}

    def make_line(mode, elapsed, region_ms):
        """`region_ms`: HIP-event duration of one step inside the timed region (first start to last end / K)."""
        total_pts = world * n_local * a.steps
        ach = bytes_per_launch / (step_ms * 1e-3) / 1e9
        traffic, traffic_source = pmc_traffic(F, a.out_dtype, regime=a.inputs)
        sweeps = region.get("sweeps")            # staging sweeps the library enqueued inside the timed region (counted, not guessed)
        sweep_ms = max(step_ms - k_only_ms, 0.0) if a.inputs == "fresh" else 0.0
        kernels = [{"name": kernel_label, "ms": round(step_ms - sweep_ms, 5), "launches_per_step": 1}]
        if a.inputs == "fresh":
            kernels.insert(0, {"name": "cache_touch_kernel", "ms": round(sweep_ms, 5), "launches_per_step": 1})
        after_h2d = (other or {}).get("after_h2d_upload", {}).get("auto_frac")
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
            "config": {"workload": "C2: 1280x384 u8 depth x %d frames per GPU -> %s xyz, fused unproject+SE(3); " % (F, a.out_dtype)
                                   + ("every step reads the NEXT of %d resident rasters (%d MB, beyond the 256 MiB Infinity Cache): fresh "
                                      "inputs from HBM, staged by the library's sweep, as in a pipeline that touches every frame once"
                                      % (n_copies, n_copies * n_local // 1000000) if a.inputs == "fresh" else
                                      "every step re-reads the SAME raster, served by the Infinity Cache from the 2nd launch on"),
                       "inputs": a.inputs, "frames_per_gpu": F, "points_per_step": world * n_local,
                       "step": A.step_text(mode), "assemble": mode, "assemble_choice": a.assemble,
                       "parallelism": "frames sharded, %d rank(s), one process per GPU" % world},
            # the two regimes side by side, as scalars: fresh rasters every step (a pipeline) / one raster re-read (rounds 1-4)
            "value_fresh_inputs": round(n_local / step_ms / 1e3, 1) if a.inputs == "fresh" else (other or {}).get("auto_Mpoints_s"),
            "value_cached_inputs": round(n_local / cached_ms / 1e3, 1),
            "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": frac_of(step_ms), "traffic": traffic, "traffic_source": traffic_source,
                         "frac_of_measured_copy": round(ach / HBM_COPY_GBS, 4),
                         "kernel": kernel_label + (" behind the library's cache_touch_kernel sweep of its raster" if a.inputs == "fresh" else ""),
                         "kernel_ms": round(step_ms, 5), "kernel_mean_ms": round(step_mean_ms, 5), "launches": step_n,
                         "frac_mean": frac_of(step_mean_ms),
                         "kernel_ms_timed_region": round(region_ms, 5) if region_ms else None,
                         "frac_timed_region": frac_of(region_ms) if region_ms else None,
                         "kernel_only_ms": round(k_only_ms, 5), "frac_kernel_only": frac_of(k_only_ms),
                         "frac_fresh_inputs": frac_of(step_ms) if a.inputs == "fresh" else (other or {}).get("auto_frac"),
                         "frac_cached_inputs": frac_of(cached_ms), "cached_inputs_ms": round(cached_ms, 5),
                         "frac_after_h2d": after_h2d,
                         # what HBM itself moves per second: with fresh inputs every algorithmic byte crosses it (read by the sweep,
                         # written by the kernel); with cached inputs only the 12 B/point of xyz do
                         "frac_hbm_side": frac_of(step_ms) if a.inputs == "fresh" else frac_of(step_ms, n_local * xyz_bytes),
                         "frac_hbm_side_cached_inputs": frac_of(cached_ms, n_local * xyz_bytes),
                         "kernels": kernels, "staging_sweeps_in_timed_region": sweeps,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "timing": "frac / kernel_ms: sustained MEDIAN of %d steps in groups of 100 between HIP events on the launch stream, "
                                   "before the timed region; *_timed_region: events around the %d timed steps (inside the wall-clock "
                                   "bracket of ms_per_step); kernel_only: events between the sweep and the fused kernel" % (step_n, a.steps),
                         "kernel_ms_timed_region_over_ms_per_step": round(region_ms / (elapsed / a.steps * 1e3), 4) if region_ms else None,
                         "other_regimes": other},
            "gpu_ms_per_step": round(region_ms, 5) if region_ms else None,
            "kernel_only_Mpoints_s_per_gpu": round(n_local / k_only_ms / 1e3, 1),
        }
        if use_dist:
            line["transport"] = asm.note
            line["assemble"] = asm.results
            line["comm"] = asm.comm_report(rccl_log_dir)
            # the quantity that scales with N: every rank fuses its own frames and keeps its shard (for the voxel / ICP stages)
            line["value_shards_resident"] = asm.results.get("none", {}).get("Mpoints_s")
            line["scaling_note"] = ("weak: every rank owns %d frames.  A strategy that leaves the WHOLE cloud on EVERY rank makes each GPU "
                                    "write world x %.0f MB into its own HBM, so its whole-job rate (`value`) cannot exceed one GPU's kernel "
                                    "rate; `value_shards_resident` (assemble.none) is the rate that scales" % (F, n_local * xyz_bytes / 1e6))
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

    mode, asm = "none", None
    if use_dist:
        B = types.SimpleNamespace(torch=torch, dist=dist, r3d=r3d, ctx=ctx, cam=cam, stream=stream, dev=dev, dev_index=dev_index,
                                  backend=backend, rank=rank, world=world, F=F, table=table, shard=shard, full=full, depth_all=depth_all,
                                  pose_all=pose_all, out_np=out_np, xyz_bytes=xyz_bytes, n_local=n_local, fuse=fuse, fuse_all=fuse_all,
                                  fence=fence, max_over_ranks=max_over_ranks, next_raster=next_raster,
                                  reset_rasters=lambda: turn.update(i=0))
        asm = A.Assembly(B)
        # the shards-stay-resident job FIRST, by the full contract: it is the line that goes out if the exchange wedges
        elapsed_none, gpu_none = headline(fuse)
        asm.results["none"] = {"ms_per_step": round(elapsed_none / a.steps * 1e3, 4),
                               "Mpoints_s": round(world * n_local * a.steps / elapsed_none / 1e6, 1), "fabric_bytes_in_per_gpu": 0}
        asm.arm(lambda: make_line("none", elapsed_none, gpu_none))
        asm.setup_transports()
        asm.results["none"]["same_bits_as_single_launch"] = asm.check_none()
        mode = asm.survey(a.assemble)
    if use_dist and mode == "none":
        elapsed, region_ms = elapsed_none, gpu_none        # already measured by the full contract
    else:
        elapsed, region_ms = headline(asm.make_step(mode) if use_dist else fuse)
    if asm is not None:
        asm.disarm()

    if rank == 0:
        line = make_line(mode, elapsed, region_ms)
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
    if asm is not None:
        asm.close()
    ctx.close()


if __name__ == "__main__":
    main()
