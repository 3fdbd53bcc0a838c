#!/bin/bash
# Round 5, GPU call 7: fp64 instruction issue costs; FMA-factored constant twiddles (GL_D_FMATW) on top of RSQ3=2
OUT=gpurun_out/r05c7; mkdir -p $OUT
hipcc -O3 --offload-arch=gfx950 -w -o $OUT/mfma_f64_probe scripts/mfma_f64_probe.hip && timeout -k 10 120 $OUT/mfma_f64_probe > $OUT/mfma_f64_probe.txt 2>&1; cat $OUT/mfma_f64_probe.txt
echo "== cmp fin vs fmatw" >> $OUT/cmp.txt; GOMEL_PRECISION=64 timeout -k 10 300 python3 scripts/cmp_builds.py abtest/r05/lib_fin.so abtest/r05/lib_fmatw.so >> $OUT/cmp.txt 2>&1; cat $OUT/cmp.txt
for g in T D S; do GOMEL_RUNS=1 GOMEL_FP64_ITERS=-1 timeout -k 10 600 python3 scripts/ab_time.py 7 $g abtest/r05/lib_fin.so abtest/r05/lib_rsq3b.so abtest/r05/lib_fmatw.so > $OUT/ab_fp64_fmatw_$g.txt 2>&1; echo "fp64 $g"; cat $OUT/ab_fp64_fmatw_$g.txt; done
