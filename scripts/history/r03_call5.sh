#!/bin/bash
# round 3, GPU call 5: odd ISTFT + fp64 any-hop tests, then the whole suite; any-hop fp64 rates
OUT=gpurun_out/r03e; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_round3.py -q -x -k "any_hop" > $OUT/t1.log 2>&1; echo "rc $?" >> $OUT/t1.log; tail -15 $OUT/t1.log
python -m pytest tests/test_gpu_round2.py -q -x -k "any_frame_length" > $OUT/t2.log 2>&1; echo "rc $?" >> $OUT/t2.log; tail -15 $OUT/t2.log
python -m pytest tests -m gpu -q > $OUT/gputests.log 2>&1; echo "pytest rc $?" >> $OUT/gputests.log; tail -4 $OUT/gputests.log
GOMEL_PRECISION=64 python3 scripts/hop_probe.py > $OUT/hop_probe_fp64.txt 2>&1; GOMEL_PRECISION=64 GOMEL_PATH=1 python3 scripts/hop_probe.py >> $OUT/hop_probe_fp64.txt 2>&1; tail -30 $OUT/hop_probe_fp64.txt
echo done
