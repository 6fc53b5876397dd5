#!/usr/bin/env python3
"""print a rocprofv3 *_kernel_stats.csv: python tools/print_stats.py FILE [repetitions of the profiled unit] [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
top = int(sys.argv[3]) if len(sys.argv) > 3 else 16
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:top]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), "%9.1f us avg" % (float(r["AverageNs"]) / 1e3),
          "%5.1f %%" % (100 * float(r["TotalDurationNs"]) / tot))
print("total kernel time per repetition: %.2f ms" % (tot / reps / 1e6))
