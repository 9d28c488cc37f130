#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the evidence set of a round on the current sources, everything under gpurun_out/$1/.
#   bash scripts/collect_round.sh r05
#   -> GPU suite log, default bench line, kernel stats + PMC summaries (bench / shard / mult / d256), instruction mix, step timelines
set -u
TAG=${1:-round}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > $OUT/gpu_suite.log 2>&1; tail -2 $OUT/gpu_suite.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-300 $OUT/bench_default.json
for what in bench shard mult d256; do
  bash scripts/collect_profiles.sh ${TAG}_prof_$what $what > /dev/null 2>&1
  P=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_$what
  cp $P/kernel_stats.csv $OUT/${what}_kernel_stats.csv 2>/dev/null
  cp $P/pmc_summary.json $OUT/${what}_pmc_summary.json 2>/dev/null
  cp $P/under_rocprof.json $OUT/${what}_under_rocprof.json 2>/dev/null
  cp $P/latest_${what}_pmc_summary.json $OUT/ 2>/dev/null
done
bash scripts/pmc_insts.sh ${TAG}_insts bench > /dev/null 2>&1; cp $GRAFT_REPO_ROOT/gpurun_out/${TAG}_insts/pmc_summary.json $OUT/bench_insts_pmc_summary.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tl() {  # name, first kernel of a step, command...
  local name=$1; local key=$2; shift; shift
  rm -rf $OUT/tl_$name
  rocprofv3 --kernel-trace --output-format csv -d $OUT/tl_$name -o t -- "$@" > /dev/null 2> $OUT/tl_$name.err
  python3 scripts/step_timeline.py $OUT/tl_$name $key > $OUT/step_timeline_$name.txt 2>&1
  rm -rf $OUT/tl_$name $OUT/tl_$name.err
}
tl n1e7 niw_lean_kernel python3 scripts/config_step.py niw 64 10000000 60 notiming
tl shard niw_lean_kernel python3 scripts/config_step.py niw 64 1250000 100 notiming
tl var4 niw_sweep_direct python3 scripts/config_step.py niw 64 10000000 60 notiming x 4
tl c4 sweep python3 scripts/config_step.py mult 1000 1000000 60 notiming
tl c5_shard niw_sweep_kernel python3 scripts/config_step.py niw 256 625000 40 notiming
ls -la $OUT
