#!/bin/bash
# Everything under profiles/ for one round, in one gpurun call:  gpurun --timeout 1150 -- 'bash tools/collect_profiles.sh r2'
# (kernel-trace statistics, FETCH_SIZE / WRITE_SIZE in separate --pmc passes, one MFMA counter pass, bench lines)
R=${1:-r3}
set -x
bash tools/profile.sh ${R}_compact_b256 --steps 200 --warmup 20 > /dev/null
bash tools/profile.sh ${R}_compact_b4096 --batch 4096 --rows 100000 --steps 30 --warmup 5 > /dev/null
bash tools/profile.sh ${R}_fc_b256 --ae-form FC --steps 200 --warmup 20 > /dev/null
bash tools/profile.sh ${R}_fc_b4096 --ae-form FC --batch 4096 --rows 100000 --steps 30 --warmup 5 > /dev/null
for wl in "compact_b256 --steps 20 --warmup 4" "compact_b4096 --batch 4096 --rows 100000 --steps 6 --warmup 3" "fc_b4096 --ae-form FC --batch 4096 --rows 100000 --steps 6 --warmup 3"; do
  set -- $wl; name=$1; shift
  bash tools/pmc.sh ${R}_pmc_fetch_$name "FETCH_SIZE" "$@" > /dev/null
  bash tools/pmc.sh ${R}_pmc_write_$name "WRITE_SIZE" "$@" > /dev/null
  python3 tools/pmc_summary.py traffic gpurun_out/${R}_pmc_fetch_$name.db gpurun_out/${R}_pmc_write_$name.db gpurun_out/${R}_pmc_traffic_$name.json "$*"
  rm -f gpurun_out/${R}_pmc_fetch_$name.db gpurun_out/${R}_pmc_write_$name.db
done
bash tools/pmc.sh ${R}_pmc_mfma_fc_b4096 "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" --ae-form FC --batch 4096 --rows 100000 --steps 6 --warmup 3 > /dev/null
python3 tools/pmc_summary.py counters gpurun_out/${R}_pmc_mfma_fc_b4096.db gpurun_out/${R}_pmc_mfma_fc_b4096.json "--ae-form FC --batch 4096 --rows 100000 --steps 6 --warmup 3"
rm -f gpurun_out/${R}_pmc_mfma_fc_b4096.db
bash tools/pmc.sh ${R}_pmc_mfma_compact_b4096 "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" --batch 4096 --rows 100000 --steps 6 --warmup 3 > /dev/null
python3 tools/pmc_summary.py counters gpurun_out/${R}_pmc_mfma_compact_b4096.db gpurun_out/${R}_pmc_mfma_compact_b4096.json "--batch 4096 --rows 100000 --steps 6 --warmup 3"
rm -f gpurun_out/${R}_pmc_mfma_compact_b4096.db
mkdir -p profiles; cp gpurun_out/${R}_pmc_traffic_*.json profiles/      # bench.py reads roofline.traffic from them
python3 bench.py > gpurun_out/${R}_bench_compact.log 2>&1; grep '^{' gpurun_out/${R}_bench_compact.log > gpurun_out/${R}_bench_compact.json
python3 bench.py --ae-form FC > gpurun_out/${R}_bench_fc.log 2>&1; grep '^{' gpurun_out/${R}_bench_fc.log > gpurun_out/${R}_bench_fc.json
python3 tools/rank_scale.py > gpurun_out/${R}_rank_scale.log 2>&1
# per block shape, every kernel ALONE (HIP events, all streams drained before each measurement)
python3 bench.py --batch 4096 --rows 100000 --steps 80 --warmup 5 --cpu-budget 0 --no-epoch --no-configs2 --no-trials --roofline-detail 2> /dev/null | grep '^{' > gpurun_out/${R}_bench_compact_b4096_per_shape.json
# timelines of one step (rocprofv3 kernel trace -> tools/timeline.py): queues, gaps, overlap
for wl in "compact_b256 --steps 100 --warmup 10" "compact_b4096 --batch 4096 --rows 100000 --steps 30 --warmup 5"; do
  set -- $wl; name=$1; shift
  (cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_tl && mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_tl && cd $GRAFT_REPO_ROOT &&
   rocprofv3 --kernel-trace -d gpurun_out/prof_tl -o run -- python3 bench.py "$@" --cpu-budget 0 --no-roofline --no-epoch --no-configs2 --no-trials > gpurun_out/prof_tl/bench.log 2>&1 &&
   python3 tools/timeline.py $(find gpurun_out/prof_tl -name '*_results.db' | head -1) 3 all > gpurun_out/${R}_timeline_$name.txt; rm -rf gpurun_out/prof_tl)
done
ls -la gpurun_out/${R}_*
