mkdir -p gpurun_out/r06h
python -m pytest tests/test_gpu_niw.py tests/test_gpu_uninit.py -m gpu -q -x > gpurun_out/r06h/gpu_niw.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06h/gpu_niw.log
grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r06h/gpu_niw.log | tail -30
for mv in 4 1; do
 for opt in "x" "30=0"; do
  export DPMM_STEP_OPTS=""; [ "$opt" != "x" ] && export DPMM_STEP_OPTS="$opt"
  echo "=== MixtureVar $mv opts=$DPMM_STEP_OPTS"; python3 scripts/config_step.py niw 64 10000000 40 timing x $mv 2>&1 | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print({k:d[k] for k in ('ms_per_step','sweep_kernel_ms','stats_kernels_ms','label_agreement')}, {k:round(v,2) for k,v in d['work'].items() if k in ('wave_tiles','full_evals','direction_screens','bf16_bottom_screens','b3_evals')})"
 done
done
export DPMM_STEP_OPTS=""
python3 scripts/config_step.py niw 64 10000000 40 timing 2>&1 | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('headline shape', {k:d[k] for k in ('ms_per_step','sweep_kernel_ms','stats_kernels_ms')})"
python3 scripts/config_step.py niw 64 1250000 100 timing 2>&1 | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('shard', {k:d[k] for k in ('ms_per_step','sweep_kernel_ms','stats_kernels_ms')})"
