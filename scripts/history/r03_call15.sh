#!/bin/bash
OUT=gpurun_out/r03x; mkdir -p $OUT; export TMPDIR=/tmp
GOMEL_LIB=abtest/s_spec.so timeout -k 5 120 python3 scripts/precise_probe.py T 64 10 2>&1 | grep -v amdgpu.ids | tee $OUT/probe_spec.log
grep -q "max rms difference" $OUT/probe_spec.log || { echo "spec build failed or hung: stop"; exit 1; }
for r in 1 2 3; do for l in s_prev s_base s_spec; do
  echo "$l fp64 T: $(GOMEL_RUNS=1 GOMEL_FP64_ITERS=-1 timeout -k 5 120 python3 scripts/prof_lib.py abtest/$l.so 256 400 30 T | tail -1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('%.4f ms/launch' % (d['dom64_ms']/d['iter64_launches']))")"
done; done 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_fp64_T.log
for r in 1 2; do for l in s_prev s_base; do
  echo "$l fp32 T steady: $(GOMEL_RUNS=2 timeout -k 5 120 python3 scripts/prof_lib.py abtest/$l.so 256 400 60 T | tail -1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('%.4f ms/launch' % (d['dom_ms']/d['iter_launches']))")"
done; done 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_fp32_T.log
echo done
