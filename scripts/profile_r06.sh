#!/bin/bash
# Everything profiles/r06/ holds for the FINAL round-6 build.  usage: scripts/profile_r06.sh gpurun_out/r06p [tests|traces|pmc|bench|extras ...]  (default: all)
# Every stage appends to $OUT/progress.log (a GPU call that is silent for seven minutes is taken to be hung).
OUT=$1; shift; STAGES=${*:-tests traces pmc bench extras}; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
say() { echo "$(date +%T) $*" >> $OUT/progress.log; }
kt() { name=$1; shift; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/$name" -- "$@" > "$ROOT/$OUT/$name.log" 2>&1); cp $OUT/$name/*/*kernel_stats.csv $OUT/${name}_kernel_stats.csv 2>/dev/null; say "trace $name"; }
for st in $STAGES; do case $st in
tests)
  python3 -m pytest tests -q -m gpu --durations=15 -rs > $OUT/tests_gpu.log 2>&1; say "tests rc $?" ;;
traces)
  # the DEFAULT mode (35 fp64 + 25 fp32 launches per call), whole batch, three calls back to back
  GOMEL_RUNS=2 kt kt_default_cfg3 python3 $ROOT/scripts/prof_run.py 256 400 60 T
  python3 scripts/launch_index.py $OUT/kt_default_cfg3/*/*kernel_trace.csv k_gl_fused_d > $OUT/launch_index_default_cfg3_fused_d.txt
  GOMEL_RUNS=2 kt kt_default_cfg4 python3 $ROOT/scripts/prof_run.py 256 400 60 S
  GOMEL_RUNS=2 GOMEL_FP64_ITERS=0 kt kt_fp32_cfg3 python3 $ROOT/scripts/prof_run.py 256 400 60 T
  GOMEL_RUNS=1 GOMEL_FP64_ITERS=-1 kt kt_fp64_cfg3 python3 $ROOT/scripts/prof_run.py 256 400 60 T
  GOMEL_RUNS=2 kt kt_default_cfg2 python3 $ROOT/scripts/prof_run.py 1 342 60 T
  # the bench command itself (the default line's timed region: two calls in flight, whole-batch launches)
  kt kt_bench python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu --no-single --no-modes ;;
pmc)
  # counters: separate passes, never combined with a trace
  GOMEL_FP64_ITERS=-1 scripts/pmc_passes.sh $OUT/pmc_fp64_cfg3 256 400 6 T > $OUT/pmc_fp64_cfg3.log 2>&1; say "pmc fp64"
  GOMEL_FP64_ITERS=0 scripts/pmc_passes.sh $OUT/pmc_fp32_cfg3 256 400 6 T > $OUT/pmc_fp32_cfg3.log 2>&1; say "pmc fp32"
  for d in pmc_fp64_cfg3 pmc_fp32_cfg3; do cp $OUT/$d/summary.txt $OUT/${d}_summary.txt; done
  python3 scripts/issue_json.py r06 cfg3:fused_d=$OUT/pmc_fp64_cfg3_summary.txt:k_gl_fused_d cfg3:fused=$OUT/pmc_fp32_cfg3_summary.txt:k_gl_fused > $OUT/issue.json
  scripts/traffic_refresh.sh $OUT/traffic r06 > $OUT/traffic.log 2>&1; cp $OUT/traffic/traffic.json $OUT/traffic.json; say "traffic" ;;
bench)
  python3 bench.py --steps 20 --warmup 3 > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err; say "bench cfg3 rc $?"
  python3 bench.py --workload cfg4 --no-single --steps 20 --warmup 3 > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err; say "bench cfg4 rc $?" ;;
extras)
  # two ranks on this one GPU, gloo for the collectives (the N > 1 code path of bench.py: torch first, table broadcast, sharded corpus, MAX over ranks)
  GOMEL_BENCH_BACKEND=gloo GOMEL_BENCH_NO_BIND=1 python3 bench.py --gpus 2 --steps 2 --warmup 1 --utts-per-gpu 64 --no-cpu > $OUT/bench_2rank_gloo_one_gpu.json 2> $OUT/bench_2rank_gloo_one_gpu.err; say "2-rank rehearsal rc $?"
  python3 scripts/fuzz_paths.py 300 6 > $OUT/fuzz_paths.txt 2>&1; say "fuzz rc $?"
  python3 scripts/soak.py 90 > $OUT/soak.txt 2>&1; say "soak rc $?" ;;
esac; done
say done
