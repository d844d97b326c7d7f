#!/bin/bash
# Kernel trace + stats of one bench.py configuration (no counters).
# usage: tools/trace.sh <tag> [bench args...]      outputs under gpurun_out/prof_<tag>/
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
grep "^{" "$OUT/trace.log" > "$OUT/bench.json"
python3 $ROOT/tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
head -16 "$OUT/summary.txt"
find "$OUT" -name "*.csv" -size +2000k -delete
