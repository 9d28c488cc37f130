mkdir -p gpurun_out/r06d
python -m pytest tests/test_gpu_niw.py tests/test_gpu_derive.py tests/test_gpu_multirank.py tests/test_gpu_fit.py -m gpu -q -x -k "not bench" > gpurun_out/r06d/gpu_subset.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06d/gpu_subset.log
grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r06d/gpu_subset.log | tail -30
python scripts/parts_trace.py 10000000 2>&1 | grep "^lean" | tee gpurun_out/r06d/parts_trace.txt | awk '{print $2, $4}' | tr '\n' ';'; echo
for cfg in "29 23 31 1250000 6" "29 0 31 1250000 6" "29 23 31 10000000 4" "29 0 31 10000000 4"; do
  echo "== ab_option $cfg"; python scripts/ab_option.py $cfg 2>&1 | grep "^option" ; done | tee gpurun_out/r06d/ab.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r06d/tl_shard -o t -- python3 scripts/config_step.py niw 64 1250000 100 notiming > gpurun_out/r06d/tl_shard.json 2>gpurun_out/r06d/tl_shard.err
python3 scripts/step_timeline.py gpurun_out/r06d/tl_shard lean > gpurun_out/r06d/step_timeline_shard.txt; head -12 gpurun_out/r06d/step_timeline_shard.txt
rm -rf gpurun_out/r06d/tl_shard
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r06d/tl_1e7 -o t -- python3 scripts/config_step.py niw 64 10000000 60 notiming > gpurun_out/r06d/tl_1e7.json 2>gpurun_out/r06d/tl_1e7.err
python3 scripts/step_timeline.py gpurun_out/r06d/tl_1e7 lean > gpurun_out/r06d/step_timeline_n1e7.txt; head -26 gpurun_out/r06d/step_timeline_n1e7.txt
rm -rf gpurun_out/r06d/tl_1e7
