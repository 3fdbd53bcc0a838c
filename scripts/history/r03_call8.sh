#!/bin/bash
# round 3, GPU call 8: new tests; the asm blocks without their leading s_nop: bitwise + timing (steady state: 3 calls back to back)
OUT=gpurun_out/r03i; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_round3.py tests/test_gpu_persist.py -q -x -k "host_to_host or escalation" > $OUT/t.log 2>&1; echo "rc $?" >> $OUT/t.log; tail -6 $OUT/t.log
python3 scripts/cmp_builds.py abtest/r3_cur.so abtest/r3_nonop.so > $OUT/cmp.log 2>&1; cat $OUT/cmp.log
for r in 1 2 3 4; do for l in r3_cur r3_nonop; do
  echo "$l T steady: $(GOMEL_RUNS=2 python3 scripts/prof_lib.py abtest/$l.so 256 400 60 T | tail -1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('%.4f ms/launch' % (d['dom_ms']/d['iter_launches']))")"
done; done > $OUT/ab_T.log 2>&1; cat $OUT/ab_T.log
for r in 1 2 3; do for l in r3_cur r3_nonop; do
  echo "$l S steady: $(GOMEL_RUNS=2 python3 scripts/prof_lib.py abtest/$l.so 256 400 60 S | tail -1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('%.4f ms/launch' % (d['dom_ms']/d['iter_launches']))")"
done; done > $OUT/ab_S.log 2>&1; cat $OUT/ab_S.log
echo done
