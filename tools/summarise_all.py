#!/usr/bin/env python3
"""gpurun_out/prof_<round>/ (tools/collect_profiles.sh) -> tracked summaries under profiles/:
  <round>_fuse_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the driver's bench command (verbatim)
  <round>_fuse_summary.json       that kernel: stats row, trace median / steady-state mean, the bench line, PMC traffic
  pmc_fuse_latest.json            HBM bytes per launch of the headline kernel (read by bench.py when the config matches)
  <round>_all_kernels_stats.csv   stats of every kernel at BASELINE sizes (tools/profile_all.py), verbatim
  <round>_all_kernels.json        per kernel: calls, avg / median ns, PMC FETCH_SIZE x2 + WRITE_SIZE bytes per launch,
                                  algorithmic bytes, the ratio, achieved TB/s from the median
MI355X_MICROARCH.md (HBM section): the counters are KiB; on gfx950 FETCH_SIZE counts 128-B read requests as 64 B (x2)."""
import argparse
import csv
import glob
import json
import os
import shutil
import statistics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# algorithmic bytes per launch of tools/profile_all.py's launches (C2: 49,152,000 points; C3: 500k x 500k)
N_C2 = 100 * 384 * 1280
ALGO = {
    "fuse_lane_kernel<unsigned char, true, false>": ("fused unproject+SE(3), u8 -> f32 xyz", N_C2 * 13),
    "fuse_lane_kernel<unsigned char, false, true>": ("unproject only, u8 -> f32 xyz (dword loads + ds_bpermute)", N_C2 * 13),
    "fuse_pair_kernel<unsigned char, true>": ("fused, u8 -> f64 xyz (lane pairs)", N_C2 * 25),
    "fuse_rgb_kernel<unsigned char, true>": ("fused + colour, u8 depth + rgb -> f32 xyz + rgba", N_C2 * 20),
    "apply_lane_kernel<float, false>": ("apply-T 4x4, f32 -> f32", N_C2 * 24),
    "nn_cull_kernel<1, false>": ("culled exact NN + fused 18 sums, 500k x 500k (min bytes 12(N+M)+8N)", 500000 * (24 + 8)),
    "nn_warm_kernel": ("the same search started from the previous matches' distances (wave-local, no LDS), 500k x 500k", 500000 * (24 + 8 + 4)),
    "voxel_insert_kernel<true, false>": ("voxel insert of the C2 cloud (12 B/point read; scattered 8-B atomics)", N_C2 * 12),
    "fuse_voxel_kernel<unsigned char, true, false>": ("cloud + occupied voxels of the C2 frames in one launch (13 B/point; random depth: ~1 voxel per point, "
                                                      "the atomics' worst case)", N_C2 * 13),
    "voxel_compact_kernel": ("hash table (2^27 slots, 1.07 GB) -> dense list of its ~48 M codes (read table + write codes)", (1 << 27) * 8 + 48_000_000 * 8),
    "cache_touch_kernel": ("input staging sweep of the C2 raster (read-only, 49 MB)", N_C2),
    "voxel_keys_kernel": ("sort-merge insert, stage 1: 12 B/point in; 4-byte remainder + 2 digit bytes out (+ the first pass's histogram)", N_C2 * 18),
    "piece_scatter_kernel<1>": ("sort-merge insert, pass 1 (by the low piece byte): 6 B in, remainder + 1 byte out", N_C2 * 11),
    "byte_histogram_kernel": ("sort-merge insert: histogram of the carried digit byte (1 B/point read)", N_C2 * 1),
    "piece_scatter_kernel<2>": ("sort-merge insert, pass 2 (by the high piece byte): 5 B in, remainder out + the 65536 run starts", N_C2 * 9),
    "voxel_merge_kernel": ("sort-merge insert, merge: remainders (4 B) into their table regions in LDS; 2^27-slot table written back "
                           "(and read first unless nothing was inserted since clear)", N_C2 * 4 + (1 << 27) * 8),
    "digit_scatter_kernel": ("radix scatter pass, 49.2 M 64-bit words (8 B read + 8 B written per key); other sizes share the symbol: "
                             "see calls / min / max", N_C2 * 16),
    "digit_histogram_kernel": ("radix histogram pass, 49.2 M words (8 B read per key); other sizes share the symbol", N_C2 * 8),
    "bbox_kernel": ("bounding box of a 500k-point cloud (two-stage, no atomics; 6 MB read)", 500000 * 12),
    "normals_kernel": ("normals of a 480x640 organised cloud (12 B read + 12 B written per point; neighbours from cache)", 480 * 640 * 24),
    "plane_accumulate_kernel": ("29 point-to-plane sums over ~300k matched pairs (src 12 + idx 4 + d2 4 + tgt 12 + normal 12 B/pair)", 305000 * 44),
    "plane_residual_kernel": ("r^2 + direction class per pair (44 B read, 5 B written per pair)", 305000 * 49),
}


def fresh(hits):
    """gpurun merges every collection into the same local directory: keep the files of the NEWEST run only."""
    if not hits:
        return hits
    newest = max(os.path.getmtime(h) for h in hits)
    return [h for h in hits if newest - os.path.getmtime(h) < 900]


def one(pattern):
    hits = fresh(glob.glob(pattern, recursive=True))
    if not hits:
        raise SystemExit("nothing matches " + pattern)
    return hits[0]


def most_calls(pattern, kernel):
    """Of several <pid>_kernel_stats.csv files (bench.py's regimes child is traced too) the one whose process launched
    `kernel` most often: the bench process itself."""
    best, best_calls = None, -1
    for f in fresh(glob.glob(pattern, recursive=True)):
        calls = sum(int(r["Calls"]) for r in csv.DictReader(open(f)) if kernel in r["Name"])
        if calls > best_calls:
            best, best_calls = f, calls
    if best is None:
        raise SystemExit("nothing matches " + pattern)
    return best


def pmc(dirname, counter):
    out = {}
    f = one(os.path.join(dirname, "**", "*_counter_collection.csv"))
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            out.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return {k: statistics.median(v) for k, v in out.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r05")
    ap.add_argument("--unprofiled", default=None, help="a file holding the bench line of an UNPROFILED run on the same box")
    a = ap.parse_args()
    src = os.path.join(ROOT, "gpurun_out", "prof_" + a.round)
    dst = os.path.join(ROOT, "profiles")
    # ---- headline step under the driver's command: the library's staging sweep + the fused kernel, fresh rasters every step
    stats = most_calls(os.path.join(src, "trace", "**", "*_kernel_stats.csv"), "fuse_lane_kernel")
    shutil.copy(stats, os.path.join(dst, "%s_fuse_kernel_stats.csv" % a.round))
    all_rows = list(csv.DictReader(open(stats)))
    row = [r for r in all_rows if "fuse_lane_kernel" in r["Name"]][0]
    touch = [r for r in all_rows if "cache_touch_kernel" in r["Name"]]
    trace_file = stats.replace("_kernel_stats.csv", "_kernel_trace.csv")
    trace_rows = sorted(csv.DictReader(open(trace_file)), key=lambda r: int(r["Start_Timestamp"]))
    tr = [r for r in trace_rows if "fuse_lane_kernel" in r["Kernel_Name"]]
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tr]
    bench_line = None
    for line in open(os.path.join(src, "trace.log")):
        if line.startswith("{"):
            bench_line = json.loads(line)
    steps = bench_line["steps"] if bench_line else 20
    # a STEP as the trace sees it: a sweep immediately followed by a fused launch -> from the sweep's start to the kernel's end
    # (the inter-kernel gap included); launches without a sweep in front (the cached-input leg of the bench) are kept apart
    step_ns, fused_after_sweep, fused_alone = [], [], []
    for prev, cur in zip(trace_rows, trace_rows[1:]):
        if "fuse_lane_kernel" in cur["Kernel_Name"]:
            dur = int(cur["End_Timestamp"]) - int(cur["Start_Timestamp"])
            if "cache_touch_kernel" in prev["Kernel_Name"]:
                step_ns.append(int(cur["End_Timestamp"]) - int(prev["Start_Timestamp"]))
                fused_after_sweep.append(dur)
            else:
                fused_alone.append(dur)
    alg = N_C2 * 13

    def frac(ns):
        return round(alg / ns / 1e3 / 8.0, 4)

    summary = {"round": a.round, "command": "python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-regimes --no-end-to-end",
               "rocprof_kernel_stats": {"name": row["Name"], "calls": int(row["Calls"]), "average_ns": float(row["AverageNs"]),
                                        "min_ns": int(row["MinNs"]), "max_ns": int(row["MaxNs"])},
               "rocprof_kernel_trace": {"n": len(d), "median_ns": statistics.median(d), "last_half_mean_ns": statistics.mean(d[len(d) // 2:]),
                                        "first_30_ns": d[:30], "grid": tr[0]["Grid_Size_X"], "workgroup": tr[0]["Workgroup_Size_X"],
                                        "vgpr": tr[0]["VGPR_Count"], "sgpr": tr[0]["SGPR_Count"], "lds": tr[0]["LDS_Block_Size"]}}
    if touch:
        t = touch[0]
        summary["input_staging_sweep"] = {"name": t["Name"], "calls": int(t["Calls"]), "average_ns": float(t["AverageNs"]),
                                          "note": "fresh rasters: the sweep runs in front of every launch whose raster is not presumed cached"}
    summary["algorithmic_TBps_at_kernel_average"] = alg / float(row["AverageNs"]) / 1e3
    summary["sweeps_per_fused_launch"] = (int(touch[0]["Calls"]) if touch else 0) / int(row["Calls"])
    summary["bench_line_under_rocprof"] = bench_line
    # the reconciliation VERDICT r4 asked for: the line's figures (HIP events, same process) beside the trace's
    rf = (bench_line or {}).get("roofline", {})
    touch_avg = float(touch[0]["AverageNs"]) if touch else 0.0
    reconcile = {
        "round": a.round, "what": "bench.py's roofline figures against the rocprofv3 kernel trace OF THE SAME PROCESS (so profiler overhead, "
                                  "if any, is in both); fractions of 8 TB/s at 13 B/point x 49,152,000 points",
        "trace": {
            "fused_kernel": {"calls": len(d), "mean_ns": statistics.mean(d), "median_ns": statistics.median(d),
                             "mean_of_last_%d_launches_ns" % steps: statistics.mean(d[-steps:]),
                             "frac_at_mean": frac(statistics.mean(d)), "frac_at_median": frac(statistics.median(d)),
                             "frac_at_mean_of_last_%d" % steps: frac(statistics.mean(d[-steps:]))},
            "fused_kernel_right_after_a_sweep": {"n": len(fused_after_sweep), "mean_ns": statistics.mean(fused_after_sweep) if fused_after_sweep else None,
                                                 "median_ns": statistics.median(fused_after_sweep) if fused_after_sweep else None},
            "fused_kernel_without_a_sweep_in_front": {"n": len(fused_alone), "median_ns": statistics.median(fused_alone) if fused_alone else None,
                                                      "note": "the cached-input leg (one raster re-read) and the first launches"},
            "sweep": {"calls": int(touch[0]["Calls"]) if touch else 0, "mean_ns": touch_avg},
            "step_sweep_start_to_kernel_end": {"n": len(step_ns), "mean_ns": statistics.mean(step_ns) if step_ns else None,
                                               "median_ns": statistics.median(step_ns) if step_ns else None,
                                               "mean_of_last_%d_ns" % steps: statistics.mean(step_ns[-steps:]) if step_ns else None,
                                               "frac_at_mean": frac(statistics.mean(step_ns)) if step_ns else None,
                                               "frac_at_median": frac(statistics.median(step_ns)) if step_ns else None,
                                               "frac_at_mean_of_last_%d" % steps: frac(statistics.mean(step_ns[-steps:])) if step_ns else None},
            "stats_csv_average_sum_ns": float(row["AverageNs"]) + touch_avg,
            "frac_at_stats_csv_average_sum": frac(float(row["AverageNs"]) + touch_avg)},
        "line_same_process": {k: rf.get(k) for k in ("frac", "frac_mean", "kernel_ms", "kernel_mean_ms", "launches", "frac_timed_region",
                                                     "kernel_ms_timed_region", "frac_kernel_only", "kernel_only_ms", "frac_cached_inputs",
                                                     "cached_inputs_ms")}}
    if step_ns and rf.get("kernel_ms"):
        reconcile["agreement"] = {
            "line_frac_over_trace_step_median": round(rf["frac"] / frac(statistics.median(step_ns)), 4),
            "line_frac_over_stats_csv_average_sum": round(rf["frac"] / frac(float(row["AverageNs"]) + touch_avg), 4),
            "line_kernel_only_over_trace_fused_median": round(rf["frac_kernel_only"] / frac(statistics.median(fused_after_sweep)), 4),
            "line_timed_region_over_trace_last_steps": round(rf["frac_timed_region"] / frac(statistics.mean(step_ns[-steps:])), 4)}
    if a.unprofiled and os.path.exists(a.unprofiled):
        for line in open(a.unprofiled):
            if line.startswith("{"):
                u = json.loads(line).get("roofline", {})
                reconcile["line_unprofiled_run_same_box"] = {k: u.get(k) for k in reconcile["line_same_process"]}
    json.dump(reconcile, open(os.path.join(dst, "%s_fuse_reconcile.json" % a.round), "w"), indent=1)
    # the other regimes of the launch (same kernel symbols), profiled as their own command: kept apart on purpose
    child = fresh(glob.glob(os.path.join(src, "trace_regimes", "**", "*_kernel_stats.csv"), recursive=True))
    if child:
        shutil.copy(child[0], os.path.join(dst, "%s_fuse_regimes_kernel_stats.csv" % a.round))
        for line in open(os.path.join(src, "trace_regimes.log")):
            if line.startswith("{"):
                summary["regimes_line_under_rocprof"] = json.loads(line)
    fetch, write = pmc(os.path.join(src, "pmc_fetch"), "FETCH_SIZE"), pmc(os.path.join(src, "pmc_write"), "WRITE_SIZE")
    kf = [k for k in fetch if "fuse_lane_kernel" in k][0]
    rd, wr = fetch[kf] * 1024 * 2, write[kf] * 1024
    kt = [k for k in fetch if "cache_touch_kernel" in k]
    rd_t, wr_t = (fetch[kt[0]] * 1024 * 2, write.get(kt[0], 0.0) * 1024) if kt else (0.0, 0.0)
    summary.update({"pmc_raw_KiB": {"FETCH_SIZE": fetch[kf], "WRITE_SIZE": write[kf]}, "hbm_read_bytes_per_launch_corrected_x2": rd,
                    "hbm_write_bytes_per_launch": wr, "staging_sweep_hbm_read_bytes_x2": rd_t, "staging_sweep_hbm_write_bytes": wr_t,
                    "hbm_bytes_per_launch": rd + wr + rd_t + wr_t, "algorithmic_bytes_per_launch": alg,
                    "traffic_over_algorithmic": (rd + wr + rd_t + wr_t) / alg,
                    "traffic_note": "per step = the staging sweep (reads the raster from HBM into the Infinity Cache) + the fused kernel "
                                    "(FETCH_SIZE counts its reads at the L2/fabric boundary: they are served by the Infinity Cache, so the "
                                    "raster is counted twice here and crosses HBM once); medians over the launches of the PMC passes, which "
                                    "ran bench.py --no-regimes --no-end-to-end (a profiled process must not start another program)"})
    rd, wr = rd + rd_t, wr + wr_t
    json.dump(summary, open(os.path.join(dst, "%s_fuse_summary.json" % a.round), "w"), indent=1)
    json.dump({"round": a.round, "config": {"frames": 100, "out_dtype": "float32", "depth": "u8", "inputs": "fresh"},
               "hbm_bytes_per_launch": rd + wr, "read_bytes_x2_corrected": rd, "write_bytes": wr,
               "of_which_staging_sweep_read_bytes": rd_t,
               "raw_KiB": {"FETCH_SIZE": fetch[kf], "WRITE_SIZE": write[kf], "sweep_FETCH_SIZE": fetch[kt[0]] if kt else None},
               "collected": "round %s, rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; per step = sweep + fused kernel" % a.round,
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --gpus 1 --steps 20 --warmup 5`"},
              open(os.path.join(dst, "pmc_fuse_latest.json"), "w"), indent=1)
    # ---- every kernel
    stats = one(os.path.join(src, "all", "**", "*_kernel_stats.csv"))
    shutil.copy(stats, os.path.join(dst, "%s_all_kernels_stats.csv" % a.round))
    rows = list(csv.DictReader(open(stats)))
    trace = list(csv.DictReader(open(one(os.path.join(src, "all", "**", "*_kernel_trace.csv")))))
    fetch, write = pmc(os.path.join(src, "all_pmc_FETCH_SIZE"), "FETCH_SIZE"), pmc(os.path.join(src, "all_pmc_WRITE_SIZE"), "WRITE_SIZE")
    out = {}
    for r in rows:
        name = r["Name"]
        d = [int(t["End_Timestamp"]) - int(t["Start_Timestamp"]) for t in trace if t["Kernel_Name"] == name]
        e = {"calls": int(r["Calls"]), "average_ns": float(r["AverageNs"]), "median_ns": statistics.median(d) if d else None,
             "min_ns": int(r["MinNs"]), "max_ns": int(r["MaxNs"])}
        if name in fetch and name in write:
            e["pmc_hbm_read_bytes_x2"] = fetch[name] * 2048
            e["pmc_hbm_write_bytes"] = write[name] * 1024
        for key, (what, algo) in ALGO.items():
            if key in name:
                e["what"] = what
                e["algorithmic_bytes"] = algo
                if "pmc_hbm_read_bytes_x2" in e:
                    e["traffic_over_algorithmic"] = round((e["pmc_hbm_read_bytes_x2"] + e["pmc_hbm_write_bytes"]) / algo, 4)
                if e["median_ns"]:
                    e["TBps_algorithmic_at_median"] = round(algo / e["median_ns"] / 1e3, 3)
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")
        out[short.split("(")[0]] = e
    hip_events = None
    for line in open(os.path.join(src, "all.log")):
        pass
    text = open(os.path.join(src, "all.log")).read()
    if "{" in text:
        try:
            hip_events = json.loads(text[text.index("{\n"):text.rindex("}") + 1])
        except Exception:
            hip_events = None
    json.dump({"round": a.round, "kernels": out, "hip_event_timings_profile_all": hip_events,
               "note": "PMC columns: separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/profile_all.py --short; "
                       "FETCH_SIZE x2 per MI355X_MICROARCH.md (gfx950 counts 128-B reads as 64 B)"},
              open(os.path.join(dst, "%s_all_kernels.json" % a.round), "w"), indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk in ("calls", "median_ns", "traffic_over_algorithmic", "TBps_algorithmic_at_median")}
                      for k, v in out.items() if "what" in v}, indent=1))
    print(json.dumps({k: summary[k] for k in ("rocprof_kernel_stats", "traffic_over_algorithmic")}, indent=1))


if __name__ == "__main__":
    main()
