#!/bin/bash
# round 3, GPU call 2: full GPU suite on the new fp64 spectral step + schedule mode, fp64 kernel timing, one vs two compute streams, a full bench line
OUT=gpurun_out/r03b; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests -m gpu -q > $OUT/gputests.log 2>&1; echo "pytest rc $?" >> $OUT/gputests.log; tail -4 $OUT/gputests.log
python3 scripts/sched_probe.py T 30 > $OUT/sched_T.log 2>&1; cat $OUT/sched_T.log
python3 scripts/sched_probe.py S 30 > $OUT/sched_S.log 2>&1; cat $OUT/sched_S.log
for cs in 1 2 1 2; do
  GOMEL_COMPUTE_STREAMS=$cs python3 bench.py --steps 10 --warmup 3 --no-cpu --no-single --no-precise > $OUT/bench_cs$cs.json 2>$OUT/bench_cs$cs.err
  python3 -c "import json,sys; d=json.load(open('$OUT/bench_cs$cs.json')); print('compute streams $cs: value %.0f ms/step %.2f resident %.0f launch_ms %.4f per-call %s' % (d['value'], d['ms_per_step'], d['resident_frames_per_s'], d['roofline']['launch_ms'], d['roofline']['launch_ms_per_timed_call']))"
done
python3 bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; tail -c 3000 $OUT/bench_full.json
echo done
