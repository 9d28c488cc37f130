#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 evidence.  Outputs under gpurun_out/$1/ (copy the summaries into profiles/).
#   pass 1: kernel trace + stats; further passes: PMC counters, each in its own run (never combined with other trace domains).
#   $2 = what to profile: "bench" (headline NIW D=64 N=1e7, default), "mult" (C4), "d256" (C5 shard), "shard" (C3 shard), "var4" (headline shape, MixtureVar 4)
set -u
TAG=${1:-prof}
WHAT=${2:-bench}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
case $WHAT in
  bench) CMD="python3 bench.py --steps 20 --no-cpu-baseline --no-growth --no-dense --no-legs --blocks 0"; SHORT="python3 bench.py --steps 5 --warmup 1 --settle 10 --no-cpu-baseline --no-growth --no-dense --no-legs --blocks 0";;
  shard) CMD="python3 scripts/config_step.py niw 64 1250000 100"; SHORT="python3 scripts/config_step.py niw 64 1250000 5";;
  mult)  CMD="python3 scripts/config_step.py mult 1000 1000000 20"; SHORT="python3 scripts/config_step.py mult 1000 1000000 5";;
  d256)  CMD="python3 scripts/config_step.py niw 256 625000 20"; SHORT="python3 scripts/config_step.py niw 256 625000 5";;
  var4)  CMD="python3 scripts/config_step.py niw 64 10000000 40 timing x 4"; SHORT="python3 scripts/config_step.py niw 64 10000000 5 timing x 4";;      # overlapping clusters (MixtureVar 4): niw_lean_kernel_dir
esac
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/under_rocprof.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- $SHORT > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- $SHORT > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $OUT/pmc_mfma -o p -- $SHORT > /dev/null 2> $OUT/pmc_mfma.err
python3 scripts/pmc_summary.py $OUT "$SHORT" > $OUT/pmc_summary.json
cp $OUT/pmc_summary.json $OUT/latest_${WHAT}_pmc_summary.json      # -> profiles/latest_${WHAT}_pmc_summary.json (bench.py's roofline.traffic)
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
# the raw traces are large: keep only the summaries
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma
ls -la $OUT
