#!/bin/bash
# Everything profiles/<round>/ is made of, in one GPU call.  usage: scripts/profile_all.sh gpurun_out/r02
OUT=$1; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
kt() { name=$1; shift; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/$name" -- "$@" > "$ROOT/$OUT/$name.log" 2>&1); cp $OUT/$name/*/*kernel_stats.csv $OUT/${name}_kernel_stats.csv 2>/dev/null; }
kt kt_resident_cfg3 python3 $ROOT/scripts/prof_run.py 256 400 60 T
kt kt_resident_cfg4 python3 $ROOT/scripts/prof_run.py 256 400 60 S
kt kt_bench python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-single
kt kt_tomel_phase python3 $ROOT/scripts/prof_tomel.py 64
scripts/pmc_passes.sh $OUT/pmc_cfg3 256 400 6 T > $OUT/pmc_cfg3.log 2>&1
scripts/pmc_passes.sh $OUT/pmc_cfg4 256 400 6 S > $OUT/pmc_cfg4.log 2>&1
python3 bench.py > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
python3 bench.py --workload cfg4 --no-single > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err
echo done
