#!/bin/bash
OUT=gpurun_out/r03l; mkdir -p $OUT; export TMPDIR=/tmp
for r in 1 2 3; do for l in d_base d_w1carry; do
  echo "$l fp64 T: $(GOMEL_RUNS=1 GOMEL_FP64_ITERS=-1 python3 scripts/prof_lib.py abtest/$l.so 256 400 30 T | tail -1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('%.4f ms/launch' % (d['dom64_ms']/d['iter64_launches']))")"
done; done 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_fp64_T.log
for r in 1 2; do for l in d_base d_w1carry; do
  echo "$l fp64 S: $(GOMEL_RUNS=1 GOMEL_FP64_ITERS=-1 python3 scripts/prof_lib.py abtest/$l.so 256 400 30 S | tail -1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('%.4f ms/launch' % (d['dom64_ms']/d['iter64_launches']))")"
done; done 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_fp64_S.log
echo done
