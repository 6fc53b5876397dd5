#!/bin/bash
# Every seeded stress run, one after the other (GPU box for the first five, CPU for the decoders): seconds per tool and the
# first seed as arguments.   usage: bash tools/stress_all.sh [seconds=120] [seed=$(date +%s)]
S=${1:-120}
SEED=${2:-$(date +%s)}
set -e
cd "$(dirname "$0")/.."
for t in stress_random stress_voxel stress_nn stress_hostpipe stress_dropin stress_jpeg stress_png; do
  echo "== $t (seed $SEED)"
  timeout -k 10 $((S + 120)) python3 tools/$t.py "$S" "$SEED" | tail -1
  SEED=$((SEED + 1))
done
