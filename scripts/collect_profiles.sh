#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 evidence for the headline bench.  Outputs under gpurun_out/$1/.
#   pass 1: kernel trace + stats; passes 2-4: PMC counters, each in its own run (never combined with other trace domains).
set -u
TAG=${1:-prof}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 bench.py --steps 20 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $OUT/pmc_mfma -o p -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_mfma.err
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 scripts/pmc_summary.py $OUT > $OUT/pmc_summary.json
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
# the raw traces are large: keep only the summaries
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma
ls -la $OUT
