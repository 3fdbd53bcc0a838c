#!/bin/bash
# Round 5, GPU call 1: the new full-size tests, the MFMA issue-cost probe, A/B of the first four fp64-kernel knobs.
OUT=gpurun_out/r05c1; mkdir -p $OUT
hipcc -O3 --offload-arch=gfx950 -w -o $OUT/mfma_f64_probe scripts/mfma_f64_probe.hip && timeout -k 10 120 $OUT/mfma_f64_probe > $OUT/mfma_f64_probe.txt 2>&1
echo "probe done" ; cat $OUT/mfma_f64_probe.txt
for v in paired wne oneb magse all4; do
  echo "== cmp base vs $v" >> $OUT/cmp.txt
  timeout -k 10 300 python3 scripts/cmp_builds.py abtest/r05/lib_base.so abtest/r05/lib_$v.so >> $OUT/cmp.txt 2>&1 || echo "CMP FAILED $v" >> $OUT/cmp.txt
done
cat $OUT/cmp.txt
L="abtest/r05/lib_base.so abtest/r05/lib_paired.so abtest/r05/lib_wne.so abtest/r05/lib_oneb.so abtest/r05/lib_magse.so abtest/r05/lib_all4.so"
GOMEL_RUNS=1 GOMEL_FP64_ITERS=-1 timeout -k 10 900 python3 scripts/ab_time.py 5 T $L > $OUT/ab_fp64_T.txt 2>&1
cat $OUT/ab_fp64_T.txt
GOMEL_RUNS=1 GOMEL_FP64_ITERS=0 timeout -k 10 600 python3 scripts/ab_time.py 5 T abtest/r05/lib_base.so abtest/r05/lib_paired.so > $OUT/ab_fp32_T.txt 2>&1
cat $OUT/ab_fp32_T.txt
GOMEL_RUNS=1 GOMEL_FP64_ITERS=0 timeout -k 10 600 python3 scripts/ab_time.py 5 S abtest/r05/lib_base.so abtest/r05/lib_paired.so > $OUT/ab_fp32_S.txt 2>&1
cat $OUT/ab_fp32_S.txt
timeout -k 10 900 python3 -m pytest tests/test_gpu_round5.py -x -q -s > $OUT/tests_round5.log 2>&1; echo "tests rc $?"; tail -8 $OUT/tests_round5.log
