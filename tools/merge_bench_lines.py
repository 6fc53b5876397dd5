#!/usr/bin/env python3
"""gpurun_out/bench_lines_<round>/*.json (tools/collect_bench_lines.sh) -> profiles/<round>_bench_lines.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "gpurun_out", "bench_lines_" + rnd)
out = {}
for name in ("driver", "default", "apply", "icp", "voxel", "c5"):
    p = os.path.join(src, name + ".json")
    if not os.path.exists(p):
        continue
    lines = [ln for ln in open(p) if ln.startswith("{")]
    if lines:
        out[name] = json.loads(lines[-1])
json.dump(out, open(os.path.join(ROOT, "profiles", rnd + "_bench_lines.json"), "w"), indent=1)
for k, v in out.items():
    rf = v.get("roofline", {})
    print("%-8s value %-12s %-40s frac %s" % (k, v.get("value"), v.get("unit", "")[:40], rf.get("frac")))
