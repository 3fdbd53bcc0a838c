#!/bin/bash
# Round 5, GPU call 9: three-address overlap-add accumulate in k_gl_fused_d (GL_D_OLA3)
OUT=gpurun_out/r05c9; mkdir -p $OUT
GOMEL_PRECISION=64 timeout -k 10 300 python3 scripts/cmp_builds.py abtest/r05/lib_cur.so abtest/r05/lib_ola3.so > $OUT/cmp.txt 2>&1; cat $OUT/cmp.txt
for g in T D S; do GOMEL_RUNS=1 GOMEL_FP64_ITERS=-1 timeout -k 10 600 python3 scripts/ab_time.py 7 $g abtest/r05/lib_cur.so abtest/r05/lib_ola3.so > $OUT/ab_fp64_ola3_$g.txt 2>&1; echo "fp64 $g"; cat $OUT/ab_fp64_ola3_$g.txt; done
