#!/bin/bash
# Run ON THE GPU BOX: instruction-mix counters of the headline sweep kernel (two PMC passes, each its own run).
set -u
#   $2 = bench (default) | d256 | d128 | mult
TAG=${1:-insts}
WHAT=${2:-bench}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
case $WHAT in
  bench) SHORT="python3 bench.py --steps 5 --warmup 1 --settle 10 --no-cpu-baseline --no-growth --no-dense --no-legs --blocks 0";;
  d256)  SHORT="python3 scripts/config_step.py niw 256 625000 5";;
  d128)  SHORT="python3 scripts/config_step.py niw 128 1250000 5";;
  mult)  SHORT="python3 scripts/config_step.py mult 1000 1000000 5";;
esac
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS --output-format csv -d $OUT/pmc_a -o p -- $SHORT > /dev/null 2> $OUT/pmc_a.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAVES --output-format csv -d $OUT/pmc_b -o p -- $SHORT > /dev/null 2> $OUT/pmc_b.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_c -o p -- $SHORT > /dev/null 2> $OUT/pmc_c.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/pmc_d -o p -- $SHORT > /dev/null 2> $OUT/pmc_d.err
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_e -o p -- $SHORT > /dev/null 2> $OUT/pmc_e.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_f -o p -- $SHORT > /dev/null 2> $OUT/pmc_f.err
python3 scripts/pmc_summary.py $OUT > $OUT/pmc_summary.json
rm -rf $OUT/pmc_a $OUT/pmc_b $OUT/pmc_c $OUT/pmc_d $OUT/pmc_e $OUT/pmc_f
