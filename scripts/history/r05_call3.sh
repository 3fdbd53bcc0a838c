#!/bin/bash
# Round 5, GPU call 3: GL_PAIRED 0 / 1 / 3 on both fused kernels, all geometries
OUT=gpurun_out/r05c3; mkdir -p $OUT
for v in p1 p3; do
  echo "== cmp p0 vs $v" >> $OUT/cmp.txt
  timeout -k 10 300 python3 scripts/cmp_builds.py abtest/r05/lib_p0.so abtest/r05/lib_$v.so >> $OUT/cmp.txt 2>&1 || echo "CMP FAILED $v" >> $OUT/cmp.txt
done
cat $OUT/cmp.txt
L="abtest/r05/lib_p0.so abtest/r05/lib_p1.so abtest/r05/lib_p3.so"
for g in T D S; do
  GOMEL_RUNS=1 GOMEL_FP64_ITERS=-1 timeout -k 10 900 python3 scripts/ab_time.py 7 $g $L > $OUT/ab_fp64_$g.txt 2>&1; echo "fp64 $g"; cat $OUT/ab_fp64_$g.txt
  GOMEL_RUNS=1 GOMEL_FP64_ITERS=0 timeout -k 10 900 python3 scripts/ab_time.py 7 $g $L > $OUT/ab_fp32_$g.txt 2>&1; echo "fp32 $g"; cat $OUT/ab_fp32_$g.txt
done
