#!/bin/bash
# Round 5, GPU call 14: the persistent kernels with the paired FFT form -- their tests (bitwise against the generic pair), the whole GPU suite, configs[1] timing
OUT=gpurun_out/r05c14; mkdir -p $OUT
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $OUT/tests_gpu.log 2>&1; rc=$?; echo "gpu tests rc $rc"; tail -4 $OUT/tests_gpu.log
[ $rc -eq 0 ] && for i in 1 2 3; do GOMEL_RUNS=2 python3 scripts/prof_run.py 1 342 60 T | tail -1; done > $OUT/cfg2_default.txt 2>&1; cat $OUT/cfg2_default.txt
[ $rc -eq 0 ] && python3 bench.py --steps 10 --warmup 2 --no-cpu --no-modes > $OUT/bench_cfg3_nocpu.json 2> $OUT/bench.err; python3 -c "
import json; d=json.loads(open('$OUT/bench_cfg3_nocpu.json').read().strip().splitlines()[-1]); print(d['value'], d['single_utterance'])"
