#!/bin/bash
# Mutation check of the draw tests (run on the GPU box, on the throw-away snapshot gpurun makes): a degrees-of-freedom error of one in the
# Bartlett diagonal must FAIL the law tests, an error in the draw kernel alone must FAIL the value-level test.
#   gpurun -- 'bash scripts/dof_bug_check.sh > gpurun_out/dof_bug_check.log 2>&1'
set -u
F=dpmmsubclusters.jl_amd/csrc/niw_master.hip
T="tests/test_gpu_master.py"
echo "== unmodified"
python -m pytest $T -m gpu -q -k "value_level or small_nu" 2>&1 | tail -3
cp $F /tmp/niw_master.orig
echo "== mutation 1: chi with nu - r + 1 degrees of freedom in the shared helper (inputs and draw agree, the law is wrong)"
sed -i 's|0.5 \* (nu - (double)r)|0.5 * (nu - (double)r + 1.0)|' $F
make -s -C dpmmsubclusters.jl_amd/csrc 2>&1 | tail -2
python -m pytest $T -m gpu -q -k "value_level or small_nu" 2>&1 | tail -12
cp /tmp/niw_master.orig $F
echo "== mutation 2: the draw kernel alone uses nu + 1"
sed -i 's|Y\[(int64_t)r \* DP + r\] = bartlett_diag(A.seed, id, epoch, r, nu);|Y[(int64_t)r * DP + r] = bartlett_diag(A.seed, id, epoch, r, nu + 1.0);|' $F
make -s -C dpmmsubclusters.jl_amd/csrc 2>&1 | tail -2
python -m pytest $T -m gpu -q -k "value_level or small_nu" 2>&1 | tail -12
cp /tmp/niw_master.orig $F
make -s -C dpmmsubclusters.jl_amd/csrc 2>&1 | tail -2
