#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh TAG
# 1. rocprofv3 --kernel-trace --stats of the default bench command (C3)   2. PMC passes (FETCH_SIZE | WRITE_SIZE | clock + MFMA busy)
#    on the direct fp32 tile kernel   3. the same on the pre-screened path.  Everything lands in gpurun_out/prof_TAG/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=$1; O=gpurun_out/prof_$T; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 3 --warmup 1 --no-secondary > $O/bench.json 2> $O/bench.err || exit 1     # (the secondary block would put C2 / shard launches of the same kernel into the per-kernel averages)
echo "trace done" 
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_direct_$n -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-prescreen --no-secondary > $O/pmc_direct_$n.json 2> $O/pmc_direct_$n.err || exit 1
  echo "direct $n done"
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_pre_$n -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --opt prescreen=1 > $O/pmc_pre_$n.json 2> $O/pmc_pre_$n.err || exit 1
  echo "prescreen $n done"
done
python3 tools/pmc_summary.py $O $T
# keep what goes to profiles/ small: the stats, the summaries, the bench lines
find $O -name "*_kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${T}_c3_kernel_stats.csv
cp $O/bench.json $O/${T}_c3_bench_under_rocprof.json
find $O -name "*_counter_collection.csv" -delete; find $O -name "*_kernel_trace.csv" -delete
