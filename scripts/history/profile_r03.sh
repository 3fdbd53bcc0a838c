#!/bin/bash
# Everything profiles/r03/ is made of, in one GPU call (final round-3 build).  usage: scripts/profile_r03.sh gpurun_out/r03p
OUT=$1; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
kt() { name=$1; shift; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/$name" -- "$@" > "$ROOT/$OUT/$name.log" 2>&1); cp $OUT/$name/*/*kernel_stats.csv $OUT/${name}_kernel_stats.csv 2>/dev/null; }
# fp32 kernel, whole batch: three calls back to back (the first runs through the clock transient after idle)
GOMEL_RUNS=2 kt kt_resident_cfg3 python3 $ROOT/scripts/prof_run.py 256 400 60 T
python3 scripts/launch_index.py $OUT/kt_resident_cfg3/*/*kernel_trace.csv k_gl_fused > $OUT/launch_index_cfg3.txt
GOMEL_RUNS=2 kt kt_resident_cfg4 python3 $ROOT/scripts/prof_run.py 256 400 60 S
# the parity modes: fp64 throughout, and the schedule
GOMEL_RUNS=1 GOMEL_FP64_ITERS=-1 kt kt_precise_cfg3 python3 $ROOT/scripts/prof_run.py 256 400 60 T
GOMEL_RUNS=1 GOMEL_FP64_ITERS=30 kt kt_schedule_cfg3 python3 $ROOT/scripts/prof_run.py 256 400 60 T
GOMEL_RUNS=1 GOMEL_FP64_ITERS=-1 kt kt_precise_cfg4 python3 $ROOT/scripts/prof_run.py 256 400 60 S
kt kt_bench python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-single --no-precise
scripts/pmc_passes.sh $OUT/pmc_cfg3 256 400 6 T > $OUT/pmc_cfg3.log 2>&1
scripts/pmc_passes.sh $OUT/pmc_cfg4 256 400 6 S > $OUT/pmc_cfg4.log 2>&1
GOMEL_FP64_ITERS=-1 scripts/pmc_passes.sh $OUT/pmc_precise_cfg3 256 400 6 T > $OUT/pmc_precise_cfg3.log 2>&1
python3 bench.py > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
python3 bench.py --workload cfg4 --no-single > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err
echo done
