#!/bin/bash
# rocprofv3 kernel-trace summary of bench.py on the GPU box; run as:  gpurun -- 'bash tools/profile.sh <tag> <bench args...>'
# writes gpurun_out/<tag>_kernel_stats.csv (per-kernel calls / total / average, from the rocpd database through
# tools/kstats.py) and gpurun_out/<tag>_bench.json (the bench line of the profiled run)
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d "$OUT" -o run -- python3 bench.py "$@" --cpu-budget 0 --no-roofline --no-epoch --no-configs2 --no-trials > "$OUT/bench.log" 2>&1
db=$(find "$OUT" -name '*_results.db' | head -1)
python3 tools/kstats.py "$db" "gpurun_out/${TAG}_kernel_stats.csv"
grep '^{' "$OUT/bench.log" > "gpurun_out/${TAG}_bench.json" || true
rm -rf "$OUT"
head -12 "gpurun_out/${TAG}_kernel_stats.csv" | cut -c1-150
