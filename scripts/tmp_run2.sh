mkdir -p gpurun_out/r06b
python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r06b/gpu_suite.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06b/gpu_suite.log
grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r06b/gpu_suite.log | tail -40
for cfg in "28 0 1 1250000 6" "29 0 3 1250000 6" "29 1 2 1250000 6" "28 0 1 10000000 4" "29 0 3 10000000 4"; do
  echo "== ab_option $cfg"; python scripts/ab_option.py $cfg 2>&1 | grep "^option" ; done | tee gpurun_out/r06b/ab.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r06b/tl_shard -o t -- python3 scripts/config_step.py niw 64 1250000 100 notiming > gpurun_out/r06b/tl_shard.json 2>gpurun_out/r06b/tl_shard.err
python3 scripts/step_timeline.py gpurun_out/r06b/tl_shard lean > gpurun_out/r06b/step_timeline_shard.txt; head -20 gpurun_out/r06b/step_timeline_shard.txt
rm -rf gpurun_out/r06b/tl_shard
