#!/bin/bash
# Sanitizer runs of the HOST logic of libgomel_hip.so (SURVEY section 5 "race detection / sanitizers"), on the CPU box -- never on a GPU box.
# Since round 5 the device-free host logic (gomel_amd/csrc/gomel_host.hpp + gomel_host.cpp: table builder, schedule, planners, device-buffer
# cache, pinned-range registry, host-only entry points) is plain C++17: it is built ALONE into libgomel_host.so with the sanitizer -- no hipcc,
# no device code -- and tests/host_stress.cpp (6 threads on the pin registry against a fake backend, table builder / chunk planner / schedule
# fuzz, the device-buffer cache against host malloc, the device gate against fake events) runs against it under
#   1. AddressSanitizer + UndefinedBehaviorSanitizer,   2. ThreadSanitizer.
# Optional third stage (only where hipcc is on PATH, skipped with GOMEL_SAN_HOST_ONLY=1): the WHOLE library with ASan + UBSan on its host
# code (-fno-gpu-sanitize) and the CPU ABI tests of the Python suite against that build.
# usage: scripts/sanitize_host.sh [OUTDIR] [ITERS]   -> OUTDIR/{asan,tsan}/..., OUTDIR/sanitizers.log; exit code 0 = clean
OUT=${1:-/tmp/gomel_san}; ITERS=${2:-300}
mkdir -p $OUT/asan $OUT/tsan
CXX=${CXX:-$(command -v clang++ || echo /opt/rocm/lib/llvm/bin/clang++)}
[ -x "$CXX" ] || CXX=g++
LOG=$OUT/sanitizers.log; : > $LOG
fail=0
run() { echo "\$ $*" >> $LOG; "$@" >> $LOG 2>&1; rc=$?; echo "[exit $rc]" >> $LOG; [ $rc -ne 0 ] && fail=1; return $rc; }
COMMON="-O1 -g -std=c++17 -fPIC -fno-omit-frame-pointer -Wall -Wno-unused-function -Iinclude"
echo "== host logic alone, AddressSanitizer + UndefinedBehaviorSanitizer ($(date -u +%F), $CXX)" >> $LOG
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined"
run $CXX $COMMON $SAN -shared gomel_amd/csrc/gomel_host.cpp -o $OUT/asan/libgomel_host.so -pthread
run $CXX $COMMON $SAN tests/host_stress.cpp -o $OUT/asan/host_stress -L$OUT/asan -lgomel_host -Wl,-rpath,$OUT/asan -pthread
ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 run $OUT/asan/host_stress $ITERS
echo "== host logic alone, ThreadSanitizer" >> $LOG
run $CXX $COMMON -fsanitize=thread -shared gomel_amd/csrc/gomel_host.cpp -o $OUT/tsan/libgomel_host.so -pthread
run $CXX $COMMON -fsanitize=thread tests/host_stress.cpp -o $OUT/tsan/host_stress -L$OUT/tsan -lgomel_host -Wl,-rpath,$OUT/tsan -pthread
TSAN_OPTIONS=halt_on_error=0:second_deadlock_stack=1 run $OUT/tsan/host_stress $ITERS
if [ -z "$GOMEL_SAN_HOST_ONLY" ] && command -v hipcc > /dev/null; then
  echo "== whole library, ASan + UBSan on the host code, CPU ABI tests" >> $LOG
  CLANG=/opt/rocm/lib/llvm/bin/clang++
  RT=$(dirname $($CLANG -print-file-name=libclang_rt.asan-x86_64.so))
  run $CLANG $COMMON $SAN -shared-libasan -c gomel_amd/csrc/gomel_host.cpp -o $OUT/asan/gomel_host.o
  run hipcc -O1 -g --offload-arch=gfx950 -fno-gpu-sanitize -std=c++17 -fPIC -fno-omit-frame-pointer -w -Xclang -target-feature -Xclang -load-store-opt $SAN -shared-libasan -c gomel_amd/csrc/gomel_hip.hip -o $OUT/asan/gomel_hip.o
  run hipcc --offload-arch=gfx950 -shared -fPIC $SAN -shared-libasan -o $OUT/asan/libgomel_hip.so $OUT/asan/gomel_hip.o $OUT/asan/gomel_host.o
  LD_PRELOAD=$RT/libclang_rt.asan-x86_64.so ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 GOMEL_LIB=$OUT/asan/libgomel_hip.so \
      run python3 -m pytest tests/test_abi.py tests/test_tables.py tests/test_chunk_plan.py tests/test_dist.py -q -p no:cacheprovider -k "not links_from_c and not gloo"   # (the gcc-built C caller cannot link a clang-ASan library; the normal suite runs it)
fi
if grep -q "ERROR: AddressSanitizer\|runtime error:\|WARNING: ThreadSanitizer\|ERROR: ThreadSanitizer\|ERROR: LeakSanitizer" $LOG; then fail=1; fi
echo "== result: $([ $fail -eq 0 ] && echo CLEAN || echo FINDINGS)" >> $LOG
tail -5 $LOG
exit $fail
