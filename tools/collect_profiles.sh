#!/bin/bash
# Run ON THE GPU BOX (gpurun): rocprofv3 kernel-trace/stats of the driver's exact bench command, separate PMC passes for
# the headline kernel (HBM traffic), and stats of every kernel at BASELINE sizes.  Output: gpurun_out/prof_$1/...
# Summaries are made afterwards with tools/summarise_profile.py and committed under profiles/.
set -o pipefail
R=${1:-r05}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
# a profiled process must not start another program (the profiler's preload has initialised the GPU): every pass runs the
# bench with --no-regimes / --no-end-to-end, and the other regimes of the launch are profiled as their own direct command
BENCH_PMC="$BENCH --no-regimes --no-end-to-end"
echo "== kernel trace of: $BENCH_PMC"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH_PMC > $OUT/trace.log 2>&1 || echo "trace rc=$?"
tail -c 600 $OUT/trace.log
echo "== kernel trace of the regimes (rotating rasters, after H2D, config 4 as one launch), own process"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_regimes -- python3 bench.py --workload regimes > $OUT/trace_regimes.log 2>&1 || echo "regimes trace rc=$?"
tail -c 300 $OUT/trace_regimes.log
echo "== PMC FETCH_SIZE"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH_PMC > $OUT/pmc_fetch.log 2>&1 || echo "pmc fetch rc=$?"
echo "== PMC WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH_PMC > $OUT/pmc_write.log 2>&1 || echo "pmc write rc=$?"
echo "== all kernels (tools/profile_all.py)"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/all -- python3 tools/profile_all.py > $OUT/all.log 2>&1 || echo "all rc=$?"
tail -c 300 $OUT/all.log
echo "== PMC for apply / f64 fuse / rgb fuse / NN (FETCH then WRITE)"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/all_pmc_$c -- python3 tools/profile_all.py --short > $OUT/all_pmc_$c.log 2>&1 || echo "all pmc $c rc=$?"
done
# keep only what the summariser reads (CSV), drop bulky extras
find $OUT -name "*.db" -delete 2>/dev/null
du -sh $OUT
