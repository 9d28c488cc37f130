mkdir -p gpurun_out/r06i
python -m pytest tests/test_gpu_niw.py tests/test_gpu_uninit.py tests/test_gpu_fit.py tests/test_gpu_master.py -m gpu -q -x > gpurun_out/r06i/gpu_niw.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06i/gpu_niw.log
grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r06i/gpu_niw.log | tail -8
for mv in 4 1; do
  echo "=== MixtureVar $mv"; python3 scripts/config_step.py niw 64 10000000 40 timing x $mv 2>&1 | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print({k:d[k] for k in ('ms_per_step','sweep_kernel_ms','stats_kernels_ms','label_agreement')})"
done
python3 scripts/parts_trace.py 10000000 2>&1 | grep "^lean" | awk '{print $2, $4}' | tr '\n' ';'; echo
python3 scripts/config_step.py niw 64 10000000 60 notiming 2>&1 | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('headline shape', {k:d[k] for k in ('ms_per_step','ms_per_step_min')})"
