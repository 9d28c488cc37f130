#!/bin/bash
# box identity + the failing test; on failure run the option toggles in the same call
echo "== box: $(hostname) cpus $(nproc) uptime $(cut -d' ' -f1 /proc/uptime)"
grep -m1 "model name" /proc/cpuinfo
/opt/rocm/bin/rocm-smi --showuniqueid --showserial 2>/dev/null | grep -i "unique\|serial" | head -3
fail=0
for i in 1 2; do
  python -m pytest tests/test_gpu_multirank.py -q -s -x -m gpu -k "niw64_dev and host_transport" > /tmp/t_$i.log 2>&1 || fail=1
  grep "passed\|failed\|K history" /tmp/t_$i.log
done
if [ $fail = 1 ]; then
  echo "== FAILED here: toggles"
  grep -B2 -A12 "AssertionError\|assert " /tmp/t_1.log | head -60
  python -m pytest scripts/tmp/test_tmp_mr.py -q -s -m gpu -p no:cacheprovider 2>&1 | grep "opts\|passed\|failed"
  python scripts/tmp/run_case.py niw64_dev 2>&1 | grep -v amdgpu.ids | tail -2
  python scripts/race_hunt.py 6 100 2>&1 | tail -4
fi
