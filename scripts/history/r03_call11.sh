#!/bin/bash
OUT=gpurun_out/r03n; mkdir -p $OUT; export TMPDIR=/tmp
# correctness first: fused fp64 (flag barrier) against the generic fp64 kernels, under a timeout (a barrier that never completes must not hang the box)
GOMEL_LIB=abtest/d_flag.so timeout -k 5 120 python3 scripts/precise_probe.py T 64 10 2>&1 | grep -v amdgpu.ids | tee $OUT/probe_flag.log
grep -q "max rms difference" $OUT/probe_flag.log || { echo "flag build failed or hung: stop"; exit 1; }
for r in 1 2 3; do for l in d_base d_flag; do
  echo "$l fp64 T: $(GOMEL_RUNS=1 GOMEL_FP64_ITERS=-1 timeout -k 5 120 python3 scripts/prof_lib.py abtest/$l.so 256 400 30 T | tail -1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('%.4f ms/launch' % (d['dom64_ms']/d['iter64_launches']))")"
done; done 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_fp64_T.log
echo done
