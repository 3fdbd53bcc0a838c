#!/bin/bash
# round 3, GPU call 1: full GPU test suite, the precision-schedule probe, and the first profile of k_gl_fused_d
OUT=gpurun_out/r03a; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
python -m pytest tests -m gpu -q > $OUT/gputests.log 2>&1; echo "pytest rc $?" >> $OUT/gputests.log; tail -3 $OUT/gputests.log
python3 scripts/sched_probe.py T 0 10 20 30 40 50 > $OUT/sched_T.log 2>&1; cat $OUT/sched_T.log
python3 scripts/sched_probe.py S 0 10 20 30 40 50 > $OUT/sched_S.log 2>&1; cat $OUT/sched_S.log
export GOMEL_FP64_ITERS=-1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/kt_precise_cfg3" -- python3 $ROOT/scripts/prof_run.py 256 400 12 T > "$ROOT/$OUT/kt_precise_cfg3.log" 2>&1)
cp $OUT/kt_precise_cfg3/*/*kernel_stats.csv $OUT/kt_precise_cfg3_kernel_stats.csv
scripts/pmc_passes.sh $OUT/pmc_precise_cfg3 256 400 6 T > $OUT/pmc_precise_cfg3.log 2>&1
tail -5 $OUT/pmc_precise_cfg3.log
echo done
