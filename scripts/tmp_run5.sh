mkdir -p gpurun_out/r06e
python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r06e/gpu_suite.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06e/gpu_suite.log
grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r06e/gpu_suite.log | tail -40
python scripts/parts_trace.py 10000000 2>&1 | grep "^lean" | tee gpurun_out/r06e/parts_trace.txt | awk '{print $2, $4}' | tr '\n' ';'; echo
python bench.py > gpurun_out/r06e/bench_stdout.txt 2> gpurun_out/r06e/bench_stderr.txt; echo "bench rc=$?"; cp bench_details.json gpurun_out/r06e/; tail -c 3500 gpurun_out/r06e/bench_stdout.txt
