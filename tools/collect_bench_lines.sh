#!/bin/bash
# Run ON THE GPU BOX (gpurun): every bench.py workload's JSON line -> gpurun_out/bench_lines_$1/<name>.json; merged afterwards
# into profiles/$1_bench_lines.json by `python tools/collect_bench_lines.sh`-independent code at the bottom of this file's
# companion, tools/merge_bench_lines.py.
R=${1:-r05}
OUT=gpurun_out/bench_lines_$R
mkdir -p $OUT
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver.json 2> $OUT/driver.err || echo "driver rc=$?"
python3 bench.py > $OUT/default.json 2> $OUT/default.err || echo "default rc=$?"
for w in apply icp voxel c5; do
  python3 bench.py --workload $w > $OUT/$w.json 2> $OUT/$w.err || echo "$w rc=$?"
done
ls -la $OUT
