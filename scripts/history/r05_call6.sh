#!/bin/bash
# Round 5, GPU call 6: fp64 spectral step with a third-order rsqrt step (GL_D_RSQ3=1) and class-based range test (=2)
OUT=gpurun_out/r05c6; mkdir -p $OUT
for v in rsq3 rsq3b; do echo "== cmp fin vs $v" >> $OUT/cmp.txt; GOMEL_PRECISION=64 timeout -k 10 300 python3 scripts/cmp_builds.py abtest/r05/lib_fin.so abtest/r05/lib_$v.so >> $OUT/cmp.txt 2>&1; done; cat $OUT/cmp.txt
for g in T D; do GOMEL_RUNS=1 GOMEL_FP64_ITERS=-1 timeout -k 10 600 python3 scripts/ab_time.py 7 $g abtest/r05/lib_fin.so abtest/r05/lib_rsq3.so abtest/r05/lib_rsq3b.so > $OUT/ab_fp64_rsq3_$g.txt 2>&1; echo "fp64 $g"; cat $OUT/ab_fp64_rsq3_$g.txt; done
