"""GPU: the driver's contract with bench.py -- one JSON line, the required keys, a self-consistent roofline object --
checked on the exact command shape the driver uses (short run), and on the N>1 code path rehearsed with one RCCL rank."""
import json
import os
import subprocess
import sys

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline"]


def run_bench(args, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, **(env or {})), cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_headline_line_is_complete_and_self_consistent():
    """SCHEMA and ARITHMETIC IDENTITIES only.  Nothing here compares one timing with another: two measurements of the same
    launch differ by several per cent from run to run (round 3: 0.0980 / 0.0994 / 0.1057 ms on three boxes), and a perf wobble
    must never fail the gate the parity tests sit behind.  Rates are checked for being rates (positive, below the physical
    peak); how large they are is on the JSON line for the reader, not for an assert."""
    d = run_bench(["--gpus", "1", "--steps", "20", "--warmup", "5"])          # the driver's command shape
    for k in REQUIRED + ["cpu_baseline", "end_to_end"]:
        assert k in d, k
    assert d["metric"].startswith("Mpoints/s fused (1280x384 depth") and d["unit"] == "Mpoints/s"
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["inputs"] == "fresh" and "Infinity Cache" in d["config"]["workload"]
    rf = d["roofline"]
    alg = 100 * 384 * 1280 * 13
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert abs(rf["frac_of_measured_copy"] - rf["achieved"] / 6290.0) < 1e-3
    assert rf["algorithmic_bytes_per_launch"] == alg
    # `frac` IS the sustained median (>= 4000 steps before the timed region), not the 20-step window behind the fence
    assert rf["launches"] >= 4000 and abs(rf["achieved"] - alg / (rf["kernel_ms"] * 1e-3) / 1e9) < 1.0
    assert abs(rf["frac_mean"] - alg / (rf["kernel_mean_ms"] * 1e-3) / 1e9 / 8000.0) < 1e-3
    # the timed region's own figure: HIP events inside the wall-clock bracket can only be shorter than the bracket
    assert 0 < rf["kernel_ms_timed_region"] <= d["ms_per_step"] and d["gpu_ms_per_step"] == rf["kernel_ms_timed_region"]
    assert abs(rf["kernel_ms_timed_region_over_ms_per_step"] - rf["kernel_ms_timed_region"] / d["ms_per_step"]) < 1e-3
    assert abs(rf["frac_timed_region"] - alg / (rf["kernel_ms_timed_region"] * 1e-3) / 1e9 / 8000.0) < 1e-3
    assert abs(d["value"] - 100 * 384 * 1280 / d["ms_per_step"] / 1e3) / d["value"] < 1e-3
    assert d["value_shards_resident"] == d["value"]
    # the regimes side by side, as SCALARS (the driver's parser drops nested objects): fresh rasters every step -- the
    # headline -- and one raster re-read every step; the fused kernel alone; the launch right after an H2D upload; HBM's side
    for k in ("frac", "frac_mean", "frac_timed_region", "frac_kernel_only", "frac_fresh_inputs", "frac_cached_inputs",
              "frac_after_h2d", "frac_hbm_side", "frac_hbm_side_cached_inputs"):
        assert isinstance(rf[k], float) and 0.0 < rf[k] < 1.0, (k, rf[k])
    assert rf["frac_fresh_inputs"] == rf["frac"] == rf["frac_hbm_side"]      # fresh inputs: every algorithmic byte crosses HBM
    assert abs(rf["frac_hbm_side_cached_inputs"] - rf["frac_cached_inputs"] * 12 / 13) < 1e-3
    for value, ms in ((d["value_fresh_inputs"], rf["kernel_ms"]), (d["value_cached_inputs"], rf["cached_inputs_ms"]),
                      (d["kernel_only_Mpoints_s_per_gpu"], rf["kernel_only_ms"])):
        assert value > 0 and abs(value - 49152.0 / ms) < 1e-3 * value      # (the durations on the line are rounded to 10 ns)
    # every kernel of a step is named with its own duration, and they sum to the step's duration: the library's staging sweep
    # (one per launch, by its own counter: none of the 16 rotating rasters is presumed cached) and the fused kernel
    names = [k["name"] for k in rf["kernels"]]
    assert names == ["cache_touch_kernel", "fuse_lane_kernel<u8,f32,pose>"] and rf["kernel"].startswith(names[-1])
    assert abs(sum(k["ms"] for k in rf["kernels"]) - rf["kernel_ms"]) <= 1e-4 and all(k["ms"] >= 0 for k in rf["kernels"])
    assert rf["staging_sweeps_in_timed_region"] == 20
    # traffic is a RECORDED figure (rocprofv3 --pmc passes, profiles/), and the line says so
    assert (rf["traffic"] is None) == (rf["traffic_source"] is None)
    if rf["traffic"] is not None:
        assert rf["traffic_source"].startswith("recorded: profiles/pmc_fuse_latest.json @")
        assert 0.95 < rf["traffic"] / alg < 1.25
    # the other regimes, MEASURED in this run by a child process: 16 rotating rasters plain / staged / library default,
    # one raster re-read, the launch right after an H2D upload, and config 4's 1000 frames at once
    cold = rf["other_regimes"]
    assert "failed" not in cold, cold
    assert cold["raster_copies"] * 49152000 > 2 * 256 * 2 ** 20
    for k in ("plain_frac", "staged_frac", "auto_frac"):
        assert 0.0 < cold[k] < 1.0, (k, cold[k])
    # ... and WHICH launches the library staged, by its own counter: every launch of the rotating rasters (none of them is
    # presumed cached), none of the same-raster loop, the one right after the H2D upload
    assert cold["plain_sweeps_per_launch"] == 0 and cold["staged_sweeps_per_launch"] == 1 and cold["auto_sweeps_per_launch"] == 1
    same = cold["same_raster_every_launch"]
    assert same["staging_off_sweeps"] == 0 and same["auto_sweeps"] <= 1 and same["staging_forced_sweeps"] == 500
    for k in ("staging_off_frac", "auto_frac", "staging_forced_frac"):
        assert 0.0 < same[k] < 1.0
    h2d = cold["after_h2d_upload"]
    assert h2d["plain_sweeps_last_launch"] == 0 and h2d["staged_sweeps_last_launch"] == 1 and h2d["auto_sweeps_last_launch"] == 1
    for k in ("plain_frac", "staged_frac", "auto_frac"):
        assert 0.0 < h2d[k] < 1.0
    assert rf["frac_after_h2d"] == h2d["auto_frac"]
    c4 = cold["c4_1000_frames_one_gpu"]
    assert c4["points"] == 1000 * 384 * 1280 and 0.0 < c4["frac"] < 1.0, c4
    assert cold["sweep_alone_cold_ms"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["unit"] == "Mpoints/s" and cb["value"] > 0 and cb["cpu_model"]
    f2f = cb["files_to_files"]
    assert f2f["s_per_frame"] > 384 * 1280 / cb["value"] / 1e6 and f2f["Mpoints_s"] > 0      # the PLY writer only adds time
    # end to end (schema; the figures are for the reader): host buffers over PCIe, and the drop-in script on 100 PNG files
    e2e = d["end_to_end"]
    hb = e2e["host_buffers"]
    assert "failed" not in hb, hb
    assert hb["identical_to_device_resident_launch"] is True and hb["floor_Gpoints_s"] == 2.0
    for kind in ("pinned", "pageable"):
        leg = hb[kind]
        assert leg["ms"] > 0 and abs(leg["Gpoints_s"] - 49.152 / leg["ms"]) < 0.02 * leg["Gpoints_s"]
        assert abs(leg["pcie_GBps_d2h"] - 12 * leg["pcie_GBps_h2d"]) <= 0.02 * leg["pcie_GBps_d2h"] + 0.1
        assert leg["pcie_GBps_d2h"] < 128.0                                   # PCIe 5 x16 cannot do more: a rate, not noise
    assert hb["meets_floor"] == (hb["pinned"]["Gpoints_s"] >= 2.0)
    drop = e2e["dropin_camera_to_world"]
    assert "failed" not in drop, drop
    assert drop["frames"] == 100 and drop["points"] == 49152000 and drop["wall_s"] > 0
    assert drop["bytes_written"] > 100 * 384 * 1280 * 30                       # ~20 MB camera txt per frame + world txt + PLY
    assert abs(drop["Mpoints_s"] - 49.152 / drop["wall_s"]) < 0.02 * drop["Mpoints_s"] + 0.1
    assert drop["cpu_reference_s_per_frame"] == f2f["s_per_frame"]


def test_cached_inputs_regime_is_still_a_flag_away():
    """`--inputs cached` = rounds 1-4's headline regime (ONE raster re-read every step, served by the Infinity Cache): the
    line says which regime it ran, no sweep is in its steps (the library's own counter), and the HBM-side fraction counts
    the 12 B/point of xyz only."""
    d = run_bench(["--gpus", "1", "--steps", "20", "--warmup", "5", "--inputs", "cached", "--no-regimes", "--no-end-to-end", "--no-cpu-baseline"])
    rf = d["roofline"]
    assert d["config"]["inputs"] == "cached" and "SAME raster" in d["config"]["workload"]
    assert [k["name"] for k in rf["kernels"]] == ["fuse_lane_kernel<u8,f32,pose>"] and rf["staging_sweeps_in_timed_region"] == 0
    assert abs(rf["frac_hbm_side"] - rf["frac"] * 12 / 13) < 1e-3 and rf["frac_fresh_inputs"] is None and rf["other_regimes"] is None
    assert rf["traffic"] is None            # the recorded PMC figure belongs to the fresh-input step


def test_multi_rank_code_path_rehearsed_with_one_rccl_rank(real_rccl):
    d = run_bench(["--gpus", "1", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"], env={"R3D_BENCH_FORCE_COLLECTIVES": "1"})
    assert "r3d_comm over RCCL" in d["transport"], d["transport"]
    modes = d["assemble"]
    for m in ("none", "outputs", "inputs", "outputs_direct", "inputs_direct", "inputs_overlap"):
        assert m in modes and "ms_per_step" in modes[m], (m, modes.get(m))
    assert d["config"]["assemble"] in modes and d["config"]["assemble"] != "none"
    for m, v in modes.items():                             # every strategy leaves the single-launch cloud, bit for bit
        assert v["same_bits_as_single_launch"] is True, (m, v)
    # first-contact instrumentation: the communicator's own account (RCCL's count, not the launcher's) and what RCCL logged
    comm = d["comm"]
    assert comm["launcher_world"] == 1 and comm["rccl"]["world"] == 1 and comm["rccl"]["rank"] == 0 and comm["rccl"]["version"] > 0
    assert isinstance(comm["rccl_log"]["picked"], dict) and isinstance(comm["rccl_log"]["init_lines"], list)


def test_secondary_workloads_print_a_roofline():
    d = run_bench(["--workload", "apply", "--steps", "200"])
    assert d["roofline"]["bound"] == "hbm" and 0.0 < d["roofline"]["frac"] < 1.0, d
    cb = d["cpu_baseline"]                                   # loop-faithful local_world (transfer_T_icp.py:71-97), 1 core
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "local_world" in cb["sample"]


def test_config5_workload_voxel_half_is_checked_against_the_oracle():
    """`bench.py --workload c5` (fuse 1080p RGBD with colour + voxel insert): the line carries a digest of the occupied set of
    its first frames' cloud; here the same frames are regenerated from the seed, fused by the library, and the set is formed
    by oracle/octomap_ref.py -- count, ignored points and both digests must agree exactly."""
    import importlib
    import numpy as np
    from helpers import PKG
    from oracle import octomap_ref as OM
    d = run_bench(["--workload", "c5", "--steps", "100", "--frames", "6"])
    assert d["roofline"]["bound"] == "hbm" and 0.0 < d["roofline"]["frac"] < 1.0, d
    chk = d["voxel_check"]
    F, k, h5, w5 = 6, chk["frames"], 1080, 1920
    assert k == 3 and chk["points"] == k * h5 * w5 and chk["overflow"] == 0
    r3d = importlib.import_module(PKG)
    rng = np.random.default_rng(chk["seed"])
    q, t = rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10
    depth = rng.random((F, h5, w5), dtype=np.float32)[:k] * 99.5 + 0.5
    cloud = r3d.fuse_frames(depth, q[:k], t[:k], intrinsics=(960.0, 960.0, 959.5, 539.5), out_dtype=np.float32)
    want, dropped = OM.occupied_set(cloud, chk["resolution"])
    assert chk["voxels"] == len(want) and chk["ignored_points"] == dropped
    assert chk["codes_xor"] == int(np.bitwise_xor.reduce(want)) and chk["codes_sum_mod_2_64"] == int(np.sum(want, dtype=np.uint64))
    assert d["voxels"] >= chk["voxels"]
    one = d["one_launch_cloud_and_voxels"]          # r3d_fuse_frames_voxel on the same frames: same three counters
    assert one["same_counters_as_two_calls"] is True and one["ms"] > 0


def test_two_rank_bench_over_the_c_abi_transport(mock_rccl):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run), two ranks sharing the one GPU: torch's own
    process group on gloo, the exchange step through r3d_comm_* bound to the mock transport (tests/c/mock_rccl.cpp).
    Every assembly strategy must run, agree (the pipelined one checks itself against the plain one) and be reported."""
    so = mock_rccl
    port = 29950 + os.getpid() % 40
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--frames", "8"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=dict(os.environ, MASTER_ADDR="127.0.0.1", R3D_DIST_BACKEND="gloo", R3D_BENCH_TRANSPORT="r3d",
                                R3D_RCCL_PATH=so, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "R3D_RCCL_PATH" in d["transport"], d.get("transport")
    for m in ("none", "outputs", "inputs", "outputs_direct", "inputs_direct", "inputs_overlap"):
        assert "ms_per_step" in d["assemble"][m], (m, d["assemble"][m])
        assert d["assemble"][m]["same_bits_as_single_launch"] is True, (m, d["assemble"][m])
        if m != "none":
            assert d["assemble"][m]["fabric_bytes_in_per_gpu"] > 0 and "xgmi_GBps_per_link" in d["assemble"][m]
    assert d["comm"]["launcher_world"] == 2 and d["comm"]["rccl"]["world"] == 2 and d["comm"]["rccl"]["version"] == -1   # (the stand-in)
    assert d["config"]["points_per_step"] == 2 * 8 * 384 * 1280 and d["config"]["assemble"] != "none"
    assert d["value_shards_resident"] == d["assemble"]["none"]["Mpoints_s"] > 0


def test_watchdog_prints_the_pre_measured_line_when_an_exchange_wedges(mock_rccl):
    """No multi-GPU node was available to the build, so the N>1 survey runs under a watchdog: if a strategy stalls (here:
    rank 1 never returns from its 3rd all-gather) rank 0 still prints ONE valid line -- the shards-stay-resident job that was
    measured by the full contract BEFORE the survey -- flagged with what hung, and every rank exits."""
    port = 29900 + os.getpid() % 40
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--frames", "8"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT,
                       env=dict(os.environ, MASTER_ADDR="127.0.0.1", R3D_DIST_BACKEND="gloo", R3D_BENCH_TRANSPORT="r3d",
                                R3D_RCCL_PATH=mock_rccl, OMP_NUM_THREADS="1", R3D_BENCH_WATCHDOG_S="8",
                                MOCK_RCCL_STALL_RANK="1", MOCK_RCCL_STALL_AFTER="2"))
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.returncode != 0                                  # a wedged exchange is a FAILED multi-GPU run, line or no line
    import re
    assert re.search(r"exitcode\s*:\s*4\b", r.stderr), r.stderr[-1500:]   # ... and the ranks said so with the watchdog's own code
    d = json.loads(lines[0])
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["config"]["assemble"] == "none"
    assert "made no progress" in d["watchdog"] and "ms_per_step" in d["assemble"]["none"]
    assert abs(d["value"] - 2 * 8 * 384 * 1280 / d["ms_per_step"] / 1e3) / d["value"] < 1e-3


def test_config5_shape_over_two_ranks(mock_rccl):
    """`bench.py --workload c5 --gpus 2` under the launcher: per-rank RGBD fuse + voxel insert + union of the sets through
    the C ABI (stand-in transport, ranks share the GPU; the workers never import torch)."""
    port = 29860 + os.getpid() % 40
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--workload", "c5", "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--frames", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT,
                       env=dict(os.environ, MASTER_ADDR="127.0.0.1", R3D_RCCL_PATH=mock_rccl, R3D_SHARE_GPU="1", OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    n = 2 * 1080 * 1920
    assert d["n_gpus"] == 2 and d["config"]["points_per_step"] == 2 * n and d["union_overflow"] == 0
    assert n < d["union_voxels"] <= 2 * n                 # random depth: nearly one voxel per point, both ranks' shards in ONE map
    assert d["fabric_bytes_in_per_gpu"] == 8 * d["union_voxels"] // 2
    assert abs(d["value"] - 2 * n / d["ms_per_step"] / 1e3) / d["value"] < 1e-2 and 0.0 < d["roofline"]["frac"] < 1.0
