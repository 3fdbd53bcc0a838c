#!/bin/bash
# Round 5, GPU call 2: PAIRED + NYQ_ONCE (+ WN_EARLY), and the polling ablation (results garbage) on the fp64 fused kernel
OUT=gpurun_out/r05c2; mkdir -p $OUT
for v in nyq nyqwne; do
  echo "== cmp base vs $v" >> $OUT/cmp.txt
  timeout -k 10 300 python3 scripts/cmp_builds.py abtest/r05/lib_base.so abtest/r05/lib_$v.so >> $OUT/cmp.txt 2>&1 || echo "CMP FAILED $v" >> $OUT/cmp.txt
done
cat $OUT/cmp.txt
L="abtest/r05/lib_base.so abtest/r05/lib_base2.so abtest/r05/lib_paired.so abtest/r05/lib_nyq.so abtest/r05/lib_nyqwne.so abtest/r05/lib_nopoll.so"
GOMEL_RUNS=1 GOMEL_FP64_ITERS=-1 timeout -k 10 900 python3 scripts/ab_time.py 7 T $L > $OUT/ab_fp64_T.txt 2>&1
cat $OUT/ab_fp64_T.txt
GOMEL_RUNS=1 GOMEL_FP64_ITERS=-1 timeout -k 10 900 python3 scripts/ab_time.py 5 D abtest/r05/lib_base.so abtest/r05/lib_nyq.so > $OUT/ab_fp64_D.txt 2>&1
cat $OUT/ab_fp64_D.txt
