#!/bin/bash
# Run ON THE GPU BOX: the sort-merge voxel insert's stages (rocprofv3 kernel trace) and their HBM traffic (separate PMC passes),
# C2's worst-case cloud.  Output gpurun_out/voxel_$1/...; summarised by tools/summarise_voxel.py into profiles/.
R=${1:-r05}
OUT=gpurun_out/voxel_$R
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/voxel_sort_once.py 2 6 > $OUT/trace.log 2>&1 || echo "trace rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 tools/voxel_sort_once.py 2 3 > $OUT/pmc_$c.log 2>&1 || echo "pmc $c rc=$?"
done
python3 tools/voxel_sort_once.py 2 6 > $OUT/unprofiled.log 2>&1
python3 tools/voxel_sort_once.py 1 3 > $OUT/unprofiled_cas.log 2>&1
tail -n 1 $OUT/unprofiled.log; tail -n 1 $OUT/unprofiled_cas.log
