#!/bin/bash
# Round 5, GPU call 12: timing ablation -- the SH-row slides of the input window and the accumulator on every other frame only (what unrolling the
# frame loop by two could save at most); results garbage
OUT=gpurun_out/r05c12; mkdir -p $OUT
GOMEL_RUNS=1 GOMEL_FP64_ITERS=-1 timeout -k 10 600 python3 scripts/ab_time.py 7 T abtest/r05/lib_cur.so abtest/r05/lib_hs.so > $OUT/ab_fp64_halfslide_T.txt 2>&1; cat $OUT/ab_fp64_halfslide_T.txt
GOMEL_RUNS=1 GOMEL_FP64_ITERS=-1 timeout -k 10 600 python3 scripts/ab_time.py 7 D abtest/r05/lib_cur.so abtest/r05/lib_hs.so > $OUT/ab_fp64_halfslide_D.txt 2>&1; cat $OUT/ab_fp64_halfslide_D.txt
