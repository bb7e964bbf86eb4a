#!/bin/bash
# CPU-only: build libmvf_host (the C++ MVF reader/writer) with AddressSanitizer + UBSan and run its tests, the footer fuzz
# pass included, against that build (MVF_HOST_LIB_PATH).  GPU sanitizers are not available on the pool; this covers the
# code that parses untrusted files.
set -e
cd "$(dirname "$0")/.."
O=${TMPDIR:-/tmp}/mvf_asan
mkdir -p "$O"
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude \
    metrovector_amd/csrc/mvf_file.cpp -o "$O/libmvf_host.so"
LD_PRELOAD="$(gcc -print-file-name=libasan.so)" ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1 \
    MVF_HOST_LIB_PATH="$O/libmvf_host.so" python -m pytest tests/test_host_mvf.py -q -x -s -m "not gpu" -p no:cacheprovider
