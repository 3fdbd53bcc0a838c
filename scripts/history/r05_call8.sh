#!/bin/bash
# Round 5, GPU call 8: the cheaper fp64 arithmetic as the product path: A/B against the previous product build, then the whole GPU suite
OUT=gpurun_out/r05c8; mkdir -p $OUT
for g in T D S; do GOMEL_RUNS=1 GOMEL_FP64_ITERS=-1 timeout -k 10 600 python3 scripts/ab_time.py 7 $g abtest/r05/lib_fin.so abtest/r05/lib_fmatw.so abtest/r05/lib_new.so > $OUT/ab_fp64_new_$g.txt 2>&1; echo "fp64 $g"; cat $OUT/ab_fp64_new_$g.txt; done
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $OUT/tests_gpu.log 2>&1; echo "gpu tests rc $?"; tail -4 $OUT/tests_gpu.log
