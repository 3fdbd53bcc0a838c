#!/bin/bash
# Round 5, GPU call 10: LDS stores of the inverse FFT's two butterfly passes leave column by column (GL_D_COLS)
OUT=gpurun_out/r05c10; mkdir -p $OUT
GOMEL_PRECISION=64 timeout -k 10 300 python3 scripts/cmp_builds.py abtest/r05/lib_cur.so abtest/r05/lib_cols.so > $OUT/cmp.txt 2>&1; cat $OUT/cmp.txt
GOMEL_RUNS=1 GOMEL_FP64_ITERS=-1 timeout -k 10 600 python3 scripts/ab_time.py 9 T abtest/r05/lib_cur.so abtest/r05/lib_cols.so > $OUT/ab_fp64_cols_T.txt 2>&1; cat $OUT/ab_fp64_cols_T.txt
