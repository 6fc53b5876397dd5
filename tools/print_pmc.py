#!/usr/bin/env python3
"""Medians per kernel of every counter in the rocprofv3 --pmc CSVs below a directory: print_pmc.py gpurun_out/voxel_sq"""
import collections
import csv
import glob
import statistics
import sys

vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        vals[r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-44:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(vals):
    if "voxel" in k or "segment" in k or "piece" in k or "scan" in k or "histogram" in k:
        print(k)
        for c in sorted(vals[k]):
            print("   %-28s %.4g" % (c, statistics.median(vals[k][c])))
