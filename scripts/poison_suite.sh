#!/bin/bash
# The GPU suite with LDS and the register files of every CU refilled with a NaN pattern before every worker call (tests/tools/poison.py):
# a test that fails here and passes without the plugin reads something it never wrote.   bash scripts/poison_suite.sh [pattern ...]
# POISON_LEVEL=kernels bash scripts/poison_suite.sh ...: in front of every kernel launch inside the library (slower, stronger)
cd "$(dirname "$0")/.."
export PYTHONPATH=tests
for pat in ${@:-0xffffffff 0x7fc00000}; do
  echo "== pattern $pat"
  POISON_PAT=$pat python -m pytest -p tools.poison_plugin tests -m gpu -q --deselect tests/test_gpu_multirank.py --deselect tests/test_gpu_uninit.py 2>&1 | grep -v amdgpu.ids | grep "poison plugin\|FAILED\|passed\|failed" | cut -c1-200
done
