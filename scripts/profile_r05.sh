#!/bin/bash
# Everything profiles/r05/ holds for the FINAL round-5 build, in one GPU call.  usage: scripts/profile_r05.sh gpurun_out/r05p
OUT=$1; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
kt() { name=$1; shift; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/$name" -- "$@" > "$ROOT/$OUT/$name.log" 2>&1); cp $OUT/$name/*/*kernel_stats.csv $OUT/${name}_kernel_stats.csv 2>/dev/null; }
# the DEFAULT mode (35 fp64 + 25 fp32 launches per call), whole batch, three calls back to back
GOMEL_RUNS=2 kt kt_default_cfg3 python3 $ROOT/scripts/prof_run.py 256 400 60 T
python3 scripts/launch_index.py $OUT/kt_default_cfg3/*/*kernel_trace.csv k_gl_fused_d > $OUT/launch_index_default_cfg3_fused_d.txt
GOMEL_RUNS=2 kt kt_default_cfg4 python3 $ROOT/scripts/prof_run.py 256 400 60 S
GOMEL_RUNS=2 GOMEL_FP64_ITERS=0 kt kt_fp32_cfg3 python3 $ROOT/scripts/prof_run.py 256 400 60 T
GOMEL_RUNS=1 GOMEL_FP64_ITERS=-1 kt kt_fp64_cfg3 python3 $ROOT/scripts/prof_run.py 256 400 60 T
GOMEL_RUNS=2 kt kt_default_cfg2 python3 $ROOT/scripts/prof_run.py 1 342 60 T
# the bench command itself (the default line's timed region: two calls in flight, whole-batch launches)
kt kt_bench python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu --no-single --no-modes
echo "traces done"
# counters: separate passes, never combined with a trace
GOMEL_FP64_ITERS=-1 scripts/pmc_passes.sh $OUT/pmc_fp64_cfg3 256 400 6 T > $OUT/pmc_fp64_cfg3.log 2>&1
GOMEL_FP64_ITERS=0 scripts/pmc_passes.sh $OUT/pmc_fp32_cfg3 256 400 6 T > $OUT/pmc_fp32_cfg3.log 2>&1
for d in pmc_fp64_cfg3 pmc_fp32_cfg3; do cp $OUT/$d/summary.txt $OUT/${d}_summary.txt; done
echo "pmc done"
scripts/traffic_refresh.sh $OUT/traffic r05 > $OUT/traffic.log 2>&1; cp $OUT/traffic/traffic.json $OUT/traffic.json
echo "traffic done"
# unprofiled bench lines
python3 bench.py --steps 20 --warmup 3 > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
python3 bench.py --workload cfg4 --no-single --steps 20 --warmup 3 > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err
echo done
