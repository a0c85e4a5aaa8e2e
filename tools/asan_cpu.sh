#!/bin/bash
# The library's host code (schedule.cpp, capi.cpp, io.cpp, dsgd.cpp) under AddressSanitizer + UBSan, CPU tests only
# (GPU ASan is not available on the pool).  Builds into /tmp/asan, leaves the in-tree build alone.
#   tools/asan_cpu.sh        -> "N passed" and the number of sanitizer reports (0 expected)
set -e
cd "$(dirname "$0")/../matrixfactorizationsgd.java_amd/csrc"
make > /dev/null
mkdir -p /tmp/asan
for f in schedule capi io dsgd; do
    /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -Wno-option-ignored -c $f.cpp -o /tmp/asan/$f.o
done
/opt/rocm/bin/hipcc -shared -fsanitize=address,undefined -Wno-option-ignored -o /tmp/asan/libmfsgd_asan.so kernels.o /tmp/asan/schedule.o /tmp/asan/capi.o \
    /tmp/asan/io.o ingest.o pack.o recommend.o /tmp/asan/dsgd.o -pthread -ldl -lrt
cd ../..
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 MFSGD_LIBRARY=/tmp/asan/libmfsgd_asan.so \
    python3 -m pytest tests/test_schedule_cpu.py tests/test_capi_cpu.py tests/test_io_cpu.py tests/test_dsgd_plan_cpu.py tests/test_dsgd_gloo.py \
    -x -q -s > /tmp/asan/out.log 2>&1 || true
tail -1 /tmp/asan/out.log
echo "sanitizer reports: $(grep -c 'runtime error\|AddressSanitizer' /tmp/asan/out.log)"
