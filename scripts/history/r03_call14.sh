#!/bin/bash
OUT=gpurun_out/r03v; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round3.py -q -k "any_hop" 2>&1 | tail -2
for l in any_base any_nobar any_base any_nobar; do
  echo "== $l fp32"; GOMEL_LIB=abtest/$l.so timeout -k 5 200 python3 scripts/hop_probe.py 2>&1 | grep "any hop" | grep "N 4096\|N 2048 hop  276\|N 1024 hop  300"
done 2>&1 | tee $OUT/ab_any.txt
