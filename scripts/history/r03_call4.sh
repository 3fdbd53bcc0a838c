#!/bin/bash
# round 3, GPU call 4: new tests (schedule exactness, cfg5 shard), cfg5 N=1 anchor line
OUT=gpurun_out/r03d; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_round3.py -q -x > $OUT/gputests_r3.log 2>&1; echo "pytest rc $?" >> $OUT/gputests_r3.log; tail -5 $OUT/gputests_r3.log
python3 bench.py --workload cfg5 --gpus 1 --utts-per-gpu 1024 --steps 3 --warmup 1 > $OUT/bench_cfg5_n1.json 2> $OUT/bench_cfg5_n1.err; tail -c 1500 $OUT/bench_cfg5_n1.json; tail -3 $OUT/bench_cfg5_n1.err
echo done
