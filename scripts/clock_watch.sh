#!/bin/bash
# Dev helper (GPU box): sample sclk / power with rocm-smi while a workload runs.  bash scripts/clock_watch.sh "<command>"
( for i in $(seq 1 40); do /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/clock_watch.txt &
W=$!
eval "$1" > gpurun_out/clock_watch_cmd.txt 2>&1
wait $W
