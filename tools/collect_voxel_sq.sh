#!/bin/bash
# Run ON THE GPU BOX: SQ counters of the sort-merge insert's kernels, four counters per pass (each its own run), and the HBM
# traffic passes.  Output gpurun_out/voxel_sq/<set>/...; tools/print_pmc.py prints the medians per kernel.
OUT=gpurun_out/voxel_sq
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/set$i -- python3 tools/voxel_sort_once.py ${1:-2} 3 > $OUT/set$i.log 2>&1 || echo "set $i rc=$?"
done
python3 tools/print_pmc.py $OUT
