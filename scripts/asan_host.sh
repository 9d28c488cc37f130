#!/bin/bash
# AddressSanitizer + UBSan build of the native master (libdpmmhost.so) under the CPU test suite (sanitizers run on the CPU build only).
#   bash scripts/asan_host.sh [pytest args]      -- the normal library is put back afterwards
set -e
cd "$(dirname "$0")/.."
LIB=dpmmsubclusters.jl_amd/lib/libdpmmhost.so
SRC=dpmmsubclusters.jl_amd/host/csrc
cp $LIB /tmp/libdpmmhost.keep
trap 'cp /tmp/libdpmmhost.keep '"$LIB" EXIT
g++ -O1 -g -std=c++17 -fPIC -pthread -fopenmp-simd -mavx2 -mfma -fvisibility=hidden -fsanitize=address,undefined -fno-omit-frame-pointer \
    -shared -o $LIB $SRC/dpmm_host.cpp $SRC/dpmm_model.cpp -lm
touch $LIB
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 \
    python -m pytest tests -q -m "not gpu" -p no:cacheprovider "$@" 2>&1 | tee /tmp/asan_host.log | grep -E "ERROR: AddressSanitizer|runtime error|SUMMARY|passed|failed" | sort | uniq -c | head -40
