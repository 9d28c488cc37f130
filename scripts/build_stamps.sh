#!/bin/bash
# Diagnostic build of the worker library with s_memtime phase stamps in the NIW sweep kernels -> lib/libdpmmhip_stamps.so
# (read by scripts/stamps_bench.py [D] [N]).  Objects go to a build directory of their own.
set -e
cd "$(dirname "$0")/../dpmmsubclusters.jl_amd/csrc"
mkdir -p build_stamps
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wno-unused-result -Wno-unused-value -Wno-pass-failed -DDPMM_STAMPS"
for f in niw_sweep.hip niw_lean.hip mult_sweep.hip labels.hip suffstats.hip niw_master.hip mult_master.hip dpmm_api.cpp; do
  /opt/rocm/bin/hipcc $FLAGS -c $f -o build_stamps/${f%.*}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../lib/libdpmmhip_stamps.so build_stamps/*.o
