#!/bin/bash
# Everything profiles/r04/ is made of besides the calibration, in one GPU call (final round-4 build).  usage: scripts/profile_r04.sh gpurun_out/r04p
OUT=$1; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
kt() { name=$1; shift; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/$name" -- "$@" > "$ROOT/$OUT/$name.log" 2>&1); cp $OUT/$name/*/*kernel_stats.csv $OUT/${name}_kernel_stats.csv 2>/dev/null; }
# the DEFAULT mode (automatic schedule: 35 fp64 + 25 fp32 launches per call), whole batch, three calls back to back
GOMEL_RUNS=2 kt kt_default_cfg3 python3 $ROOT/scripts/prof_run.py 256 400 60 T
python3 scripts/launch_index.py $OUT/kt_default_cfg3/*/*kernel_trace.csv k_gl_fused_d > $OUT/launch_index_default_cfg3_fused_d.txt
GOMEL_RUNS=2 kt kt_default_cfg4 python3 $ROOT/scripts/prof_run.py 256 400 60 S
# the opt-in fast mode (fp32 throughout) and fp64 throughout
GOMEL_RUNS=2 GOMEL_FP64_ITERS=0 kt kt_fp32_cfg3 python3 $ROOT/scripts/prof_run.py 256 400 60 T
python3 scripts/launch_index.py $OUT/kt_fp32_cfg3/*/*kernel_trace.csv "k_gl_fused<" > $OUT/launch_index_fp32_cfg3.txt
GOMEL_RUNS=1 GOMEL_FP64_ITERS=-1 kt kt_fp64_cfg3 python3 $ROOT/scripts/prof_run.py 256 400 60 T
# configs[1]: one utterance, the default schedule on the persistent kernel (one fp64 launch + one fp32 launch)
GOMEL_RUNS=2 kt kt_default_cfg2 python3 $ROOT/scripts/prof_run.py 1 342 60 T
# the bench command itself (the default line's timed region: two calls in flight, whole-batch launches)
kt kt_bench python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu --no-single --no-modes
# counters: separate passes, never combined with a trace
GOMEL_FP64_ITERS=-1 scripts/pmc_passes.sh $OUT/pmc_fp64_cfg3 256 400 6 T > $OUT/pmc_fp64_cfg3.log 2>&1
GOMEL_FP64_ITERS=0 scripts/pmc_passes.sh $OUT/pmc_fp32_cfg3 256 400 6 T > $OUT/pmc_fp32_cfg3.log 2>&1
for d in pmc_fp64_cfg3 pmc_fp32_cfg3; do cp $OUT/$d/summary.txt $OUT/${d}_summary.txt; done
# unprofiled bench lines
python3 bench.py --steps 20 --warmup 3 > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
python3 bench.py --workload cfg4 --no-single --steps 20 --warmup 3 > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err
python3 bench.py --workload cfg5 --gpus 1 --utts-per-gpu 1024 --steps 3 --warmup 1 --no-single --no-modes --no-cpu > $OUT/bench_cfg5_n1.json 2> $OUT/bench_cfg5_n1.err
# 2- and 4-rank REHEARSALS on the one GPU (gloo collectives on CPU tensors, ranks share the device: what they show is in DESIGN.md section 5)
GOMEL_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --utts-per-gpu 256 --steps 3 --warmup 1 > $OUT/bench_2rank_gloo_one_gpu.json 2> $OUT/bench_2rank_gloo_one_gpu.err
GOMEL_BENCH_BACKEND=gloo python3 bench.py --gpus 4 --utts-per-gpu 128 --steps 3 --warmup 1 > $OUT/bench_4rank_gloo_one_gpu.json 2> $OUT/bench_4rank_gloo_one_gpu.err
echo done
