#!/bin/bash
# Sanitizer runs of the HOST side of libgomel_hip.so (SURVEY section 5 "race detection / sanitizers"), on the CPU box -- never on a GPU box.
# The library is built as usual except that its HOST code carries the sanitizer (-fno-gpu-sanitize: the device code is the product's) with
#   1. AddressSanitizer + UndefinedBehaviorSanitizer: tests/host_stress.cpp (6 threads on the pin registry against a fake backend, table
#      builder / chunk planner / schedule fuzz, the device-buffer cache against host malloc), then the CPU ABI tests of the Python suite
#      (tests/test_abi.py incl. the strict-C99 caller, tests/test_tables.py, tests/test_chunk_plan.py) against that build;
#   2. ThreadSanitizer: tests/host_stress.cpp.
# usage: scripts/sanitize_host.sh [OUTDIR] [ITERS]   -> OUTDIR/{asan,tsan}/..., OUTDIR/sanitizers.log; exit code 0 = clean
OUT=${1:-/tmp/gomel_san}; ITERS=${2:-300}
mkdir -p $OUT/asan $OUT/tsan
CLANG=/opt/rocm/lib/llvm/bin/clang++
RT=$(dirname $($CLANG -print-file-name=libclang_rt.asan-x86_64.so))
HIPFLAGS="-O1 -g --offload-arch=gfx950 -fno-gpu-sanitize -std=c++17 -shared -fPIC -fno-omit-frame-pointer -w -Xclang -target-feature -Xclang -load-store-opt"
LOG=$OUT/sanitizers.log; : > $LOG
fail=0
run() { echo "\$ $*" >> $LOG; "$@" >> $LOG 2>&1; rc=$?; echo "[exit $rc]" >> $LOG; [ $rc -ne 0 ] && fail=1; return $rc; }
echo "== AddressSanitizer + UndefinedBehaviorSanitizer ($(date -u +%F))" >> $LOG
run hipcc $HIPFLAGS -fsanitize=address,undefined -fno-sanitize-recover=undefined -shared-libasan -o $OUT/asan/libgomel_hip.so gomel_amd/csrc/gomel_hip.hip
run $CLANG -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -shared-libasan -Iinclude tests/host_stress.cpp -o $OUT/asan/host_stress \
    -L$OUT/asan -lgomel_hip -Wl,-rpath,$OUT/asan -Wl,-rpath,$RT -pthread
ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 run $OUT/asan/host_stress $ITERS
LD_PRELOAD=$RT/libclang_rt.asan-x86_64.so ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 GOMEL_LIB=$OUT/asan/libgomel_hip.so \
    run python3 -m pytest tests/test_abi.py tests/test_tables.py tests/test_chunk_plan.py -q -p no:cacheprovider -k "not links_from_c"   # (the gcc-built C caller cannot link a clang-ASan library; the normal suite runs it)
echo "== ThreadSanitizer" >> $LOG
run hipcc $HIPFLAGS -fsanitize=thread -o $OUT/tsan/libgomel_hip.so gomel_amd/csrc/gomel_hip.hip
run $CLANG -O1 -g -std=c++17 -fsanitize=thread -Iinclude tests/host_stress.cpp -o $OUT/tsan/host_stress -L$OUT/tsan -lgomel_hip -Wl,-rpath,$OUT/tsan -pthread
TSAN_OPTIONS=halt_on_error=0:second_deadlock_stack=1 run $OUT/tsan/host_stress $ITERS
if grep -q "ERROR: AddressSanitizer\|runtime error:\|WARNING: ThreadSanitizer\|ERROR: ThreadSanitizer" $LOG; then fail=1; fi
echo "== result: $([ $fail -eq 0 ] && echo CLEAN || echo FINDINGS)" >> $LOG
tail -5 $LOG
exit $fail
