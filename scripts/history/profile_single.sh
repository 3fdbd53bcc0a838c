#!/bin/bash
# The single-utterance path (BASELINE configs[1], k_gl_persist) for profiles/<round>/.  usage: scripts/profile_single.sh gpurun_out/r02s
OUT=$1; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
kt() { name=$1; shift; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/$name" -- "$@" > "$ROOT/$OUT/$name.log" 2>&1); cp $OUT/$name/*/*kernel_stats.csv $OUT/${name}_kernel_stats.csv 2>/dev/null; }
kt kt_single_cfg2 python3 $ROOT/scripts/prof_run.py 1 342 60 T
GOMEL_PATH=1 kt kt_single_cfg2_generic python3 $ROOT/scripts/prof_run.py 1 342 60 T
kt kt_single_h2h python3 $ROOT/scripts/single_h2h.py 100 pinned
scripts/pmc_passes.sh $OUT/pmc_cfg2 1 342 60 T > $OUT/pmc_cfg2.log 2>&1
python3 scripts/single_h2h.py 300 > $OUT/single_h2h.txt 2>&1
python3 scripts/single_h2h.py 300 pinned >> $OUT/single_h2h.txt 2>&1
python3 scripts/persist_sweep.py > $OUT/persist_sweep.txt 2>&1
python3 scripts/persist_probe.py > $OUT/persist_probe.txt 2>&1
python3 bench.py > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
echo done
