#!/bin/bash
# One rocprofv3 hardware-counter pass of bench.py (counters in their own run: --kernel-trace + --pmc only).
#   gpurun -- 'bash tools/pmc.sh <tag> "<COUNTER ...>" <bench args...>'   ->   gpurun_out/<tag>.db (rocpd database)
set -e
TAG=$1; CTRS=$2; shift 2
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --pmc $CTRS -d "$OUT" -o run -- python3 bench.py "$@" --cpu-budget 0 --no-roofline --no-epoch --no-configs2 --no-trials > "$OUT/bench.log" 2>&1 || { tail -5 "$OUT/bench.log"; exit 1; }
db=$(find "$OUT" -name '*_results.db' | head -1)
cp "$db" "gpurun_out/${TAG}.db"
rm -rf "$OUT"
ls -la "gpurun_out/${TAG}.db"
