#!/bin/bash
# Run ON THE GPU BOX: instruction-cache counters of a sweep kernel.  bash scripts/pmc_icache.sh <tag> <D> <N>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SHORT="python3 scripts/config_step.py niw $2 $3 5"
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $OUT/pmc_a -o p -- $SHORT > /dev/null 2> $OUT/pmc_a.err
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQC_TC_INST_REQ SQC_ICACHE_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_b -o p -- $SHORT > /dev/null 2> $OUT/pmc_b.err
python3 scripts/pmc_summary.py $OUT > $OUT/pmc_summary.json
rm -rf $OUT/pmc_a $OUT/pmc_b
