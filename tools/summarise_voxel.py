#!/usr/bin/env python3
"""gpurun_out/voxel_<round>/ (tools/collect_voxel_profile.sh) -> profiles/<round>_voxel_stage_pmc.json + _voxel_kernel_stats.csv:
per stage of the sort-merge insert its calls, average / median duration, HBM bytes per launch (rocprofv3 FETCH_SIZE x 2 +
WRITE_SIZE, separate passes; MI355X_MICROARCH.md's gfx950 correction), the designed bytes, and the totals against the insert's
algorithmic bytes (12 B/point read + 8 B per distinct voxel written)."""
import csv
import glob
import json
import os
import re
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
src = os.path.join(ROOT, "gpurun_out", "voxel_" + rnd)
N = 100 * 384 * 1280
DESIGNED = {"voxel_bin_kernel": 17 * N, "segment_histogram_kernel": 1 * N, "segment_scatter_kernel": 9 * N,
            "voxel_merge": 4 * N + (1 << 27) * 8, "digit_scan_kernel": 2 * 16.8e6, "voxel_spill_kernel": 0}


def newest(pattern):
    hits = glob.glob(pattern, recursive=True)
    return max(hits, key=os.path.getmtime)


def short(name):
    for k in DESIGNED:
        if k in name:
            return k
    return None


def pmc(counter):
    out = {}
    for r in csv.DictReader(open(newest(os.path.join(src, "pmc_" + counter, "**", "*counter_collection.csv")))):
        k = short(r["Kernel_Name"])
        if k and r["Counter_Name"] == counter:
            out.setdefault(k, []).append(float(r["Counter_Value"]))
    return {k: statistics.median(v) for k, v in out.items()}


stats = newest(os.path.join(src, "trace", "**", "*kernel_stats.csv"))
shutil.copy(stats, os.path.join(ROOT, "profiles", "%s_voxel_kernel_stats.csv" % rnd))
trace = list(csv.DictReader(open(stats.replace("_kernel_stats.csv", "_kernel_trace.csv"))))
fetch, write = pmc("FETCH_SIZE"), pmc("WRITE_SIZE")
stages, total_us, total_bytes = {}, 0.0, 0.0
for k in DESIGNED:
    d = [int(t["End_Timestamp"]) - int(t["Start_Timestamp"]) for t in trace if short(t["Kernel_Name"]) == k]
    if not d:
        continue
    per_insert = len(d) / 7.0          # voxel_sort_once.py 2 6 = 7 inserts
    rd, wr = fetch.get(k, 0.0) * 2048, write.get(k, 0.0) * 1024
    stages[k] = {"launches_per_insert": round(per_insert, 2), "median_us": round(statistics.median(d) / 1e3, 1), "mean_us": round(statistics.mean(d) / 1e3, 1),
                 "hbm_read_bytes_x2": rd, "hbm_write_bytes": wr, "designed_bytes": DESIGNED[k],
                 "TBps_at_median": round((rd + wr) / statistics.median(d) / 1e3, 2)}
    total_us += per_insert * statistics.median(d) / 1e3
    total_bytes += per_insert * (rd + wr)
line = open(os.path.join(src, "unprofiled.log")).read().strip().splitlines()[-1]
ms = [float(x) for x in re.findall(r"(\d+\.\d+)(?= |ms)", line.split(":")[1].split("ms")[0])]
voxels = int(re.search(r"(\d+) voxels", line).group(1))
alg = 12 * N + 8 * voxels
out = {"round": rnd, "what": "sort-merge insert of C2's worst-case cloud (49,152,000 points -> %d voxels, 2^27-slot table), tools/voxel_sort_once.py" % voxels,
       "stages": stages, "sum_of_stage_medians_us": round(total_us, 1), "hbm_bytes_per_insert": total_bytes,
       "algorithmic_bytes": alg, "traffic_over_algorithmic": round(total_bytes / alg, 3), "designed_bytes_per_point": 52.8,
       "unprofiled_ms_per_insert": ms, "unprofiled_median_ms": statistics.median(ms),
       "frac_of_hbm_peak": round(alg / (statistics.median(ms) * 1e-3) / 8e12, 4),
       "cas_path_same_cloud": open(os.path.join(src, "unprofiled_cas.log")).read().strip().splitlines()[-1],
       "round5_first_form": {"what": "key kernel + dense first pass behind its histogram (voxel_keys_kernel, piece_scatter_kernel<1|2>, byte_histogram_kernel)",
                             "ms": "0.87-0.97", "traffic_over_algorithmic": 3.519, "bytes_per_point": 64.8, "hbm_bytes_per_insert": 3435142992,
                             "stage_medians_us": {"voxel_keys_kernel": 208.2, "piece_scatter_kernel<1>": 164.4, "byte_histogram_kernel": 20.9,
                                                  "piece_scatter_kernel<2>": 141.5, "voxel_merge": 359.1, "digit_scan_kernel x2": 21.2, "voxel_spill_kernel": 15.4},
                             "same_process_ab": "tools/voxel_front_ab.py at the commits in between: 0.94-0.96 ms (this form) vs 0.74 / 0.70 ms"},
       "round4": {"ms": "1.09-1.20", "traffic_over_algorithmic": 4.8, "bytes_per_point": 97.8, "source": "profiles/r04_voxel_sort_merge_stages.txt"}}
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_voxel_stage_pmc.json" % rnd), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("sum_of_stage_medians_us", "traffic_over_algorithmic", "unprofiled_median_ms", "frac_of_hbm_peak")}))
for k, v in stages.items():
    print("%-26s %6.1f us  %7.1f MB read  %7.1f MB written (designed %7.1f MB)" % (k, v["median_us"], v["hbm_read_bytes_x2"] / 1e6, v["hbm_write_bytes"] / 1e6, v["designed_bytes"] / 1e6))
