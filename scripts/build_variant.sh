#!/bin/bash
# Experimental build of the worker library with extra preprocessor flags -> lib/libdpmmhip_<name>.so (objects in build_<name>/)
#   bash scripts/build_variant.sh xcache -DDPMM_EXP_XCACHE ; python scripts/run_with_lib.py ... or scripts/variant_time.py
set -e
NAME=$1; shift
cd "$(dirname "$0")/../dpmmsubclusters.jl_amd/csrc"
mkdir -p build_$NAME
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wno-unused-result -Wno-unused-value -Wno-pass-failed $*"
for f in niw_sweep.hip niw_lean.hip mult_sweep.hip labels.hip suffstats.hip niw_master.hip mult_master.hip dpmm_api.cpp; do
  /opt/rocm/bin/hipcc $FLAGS -c $f -o build_$NAME/${f%.*}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../lib/libdpmmhip_$NAME.so build_$NAME/*.o
