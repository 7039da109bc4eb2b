R=r3
for wl in "compact_b256 --steps 100 --warmup 10" "compact_b4096 --batch 4096 --rows 100000 --steps 30 --warmup 5"; do
  set -- $wl; name=$1; shift
  (cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_tl && mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_tl && cd $GRAFT_REPO_ROOT &&
   rocprofv3 --kernel-trace -d gpurun_out/prof_tl -o run -- python3 bench.py "$@" --cpu-budget 0 --no-roofline --no-epoch --no-configs2 --no-trials > gpurun_out/prof_tl/bench.log 2>&1 &&
   python3 tools/timeline.py $(find gpurun_out/prof_tl -name '*_results.db' | head -1) 3 all > gpurun_out/${R}_timeline_$name.txt; rm -rf gpurun_out/prof_tl)
done
wc -l gpurun_out/r3_timeline_*
