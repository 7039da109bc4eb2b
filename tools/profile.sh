#!/bin/bash
# rocprofv3 kernel-trace summaries of bench.py on the GPU box; run as:  gpurun -- 'bash tools/profile.sh <tag> <bench args...>'
# writes gpurun_out/prof_<tag>/ and a copy of the per-kernel stats as gpurun_out/<tag>_kernel_stats.csv
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d "$OUT" -o run -- python3 bench.py "$@" --cpu-budget 0 --no-roofline --no-epoch --no-configs2 > "$OUT/bench.log" 2>&1
f=$(find "$OUT" -name '*kernel_stats.csv' | head -1)
cp "$f" "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_stats.csv"
grep '^{' "$OUT/bench.log" > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench.json" || true
find "$OUT" -name '*kernel_trace.csv' -delete      # large
tail -2 "$OUT/bench.log"
