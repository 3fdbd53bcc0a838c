#!/bin/bash
OUT=gpurun_out/r03o; mkdir -p $OUT; export TMPDIR=/tmp
# fp64 default build: correctness of the flag sync across geometries (subset of the suite), then timing
timeout -k 10 300 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round3.py -q -x -k "fp64 or precision or 64_bit or schedule" 2>&1 | tail -3
# fp32 kernel with the flag sync: correctness vs base (bitwise), timing (steady state)
timeout -k 10 200 python3 scripts/cmp_builds.py abtest/f_base.so abtest/f_flag.so 2>&1 | grep -v amdgpu.ids | tee $OUT/cmp.log
grep -q "^T run-to-run" $OUT/cmp.log || { echo "fp32 flag build failed or hung: stop"; exit 1; }
for r in 1 2 3; do for l in f_base f_flag; do
  echo "$l fp32 T steady: $(GOMEL_RUNS=2 timeout -k 5 120 python3 scripts/prof_lib.py abtest/$l.so 256 400 60 T | tail -1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('%.4f ms/launch' % (d['dom_ms']/d['iter_launches']))")"
done; done 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_fp32_T.log
for l in f_base f_flag; do
  echo "$l fp64 T: $(GOMEL_RUNS=1 GOMEL_FP64_ITERS=-1 timeout -k 5 120 python3 scripts/prof_lib.py abtest/$l.so 256 400 30 T | tail -1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('%.4f ms/launch' % (d['dom64_ms']/d['iter64_launches']))")"
done 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_fp64_T.log
echo done
