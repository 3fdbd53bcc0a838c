#!/bin/bash
# round 3, GPU call 6: wide table fill vs before: bitwise, whole-batch and chunk-sized launches, host-to-host
OUT=gpurun_out/r03g; mkdir -p $OUT; export TMPDIR=/tmp
python3 scripts/cmp_builds.py abtest/r3_rot.so abtest/r3_fill.so > $OUT/cmp.log 2>&1; cat $OUT/cmp.log
for r in 1 2 3; do for l in r3_rot r3_fill; do
  echo "$l chunk-sized (32 x 400 frames): $(python3 scripts/prof_lib.py abtest/$l.so 32 400 60 T | tail -1)"
done; done > $OUT/ab_chunk.log 2>&1; cat $OUT/ab_chunk.log
python3 scripts/ab_time.py 3 T abtest/r3_rot.so abtest/r3_fill.so > $OUT/ab_T.log 2>&1; cat $OUT/ab_T.log
python3 scripts/ab_time.py 3 S abtest/r3_rot.so abtest/r3_fill.so > $OUT/ab_S.log 2>&1; cat $OUT/ab_S.log
R=$(pwd); B=$R/abtest/lib_h2h; rm -f $B
hipcc -O2 -I$R/include -o $B $R/scripts/lib_h2h.cpp -L$R/gomel_amd -lgomel_hip -Wl,-rpath,$R/gomel_amd || exit 1
for i in 1 2 3; do echo "h2h C caller, current build: $($B 256 0 0 | tail -1)"; done
python3 bench.py --steps 10 --warmup 3 --no-cpu --no-single --no-precise > $OUT/bench.json 2>$OUT/bench.err
python3 -c "import json; d=json.load(open('$OUT/bench.json')); print('bench: value %.0f ms/step %.2f resident %.0f launch_ms %.4f chunk launch %.4f frac %.4f' % (d['value'], d['ms_per_step'], d['resident_frames_per_s'], d['roofline']['launch_ms'], d['roofline']['timed_region_launches']['launch_ms'], d['roofline']['timed_region_launches']['frac']))"
echo done
