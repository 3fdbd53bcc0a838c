#!/bin/bash
# round 3, GPU call 3: one-window-read build vs base (bitwise + timing), chunk plan sweep
OUT=gpurun_out/r03c; mkdir -p $OUT; export TMPDIR=/tmp
python3 scripts/cmp_builds.py abtest/r3_base.so abtest/r3_rot.so > $OUT/cmp.log 2>&1; cat $OUT/cmp.log
python3 scripts/ab_time.py 4 T abtest/r3_base.so abtest/r3_rot.so > $OUT/ab_T.log 2>&1; cat $OUT/ab_T.log
python3 scripts/ab_time.py 3 S abtest/r3_base.so abtest/r3_rot.so > $OUT/ab_S.log 2>&1; cat $OUT/ab_S.log
python3 scripts/ab_time.py 3 D abtest/r3_base.so abtest/r3_rot.so > $OUT/ab_D.log 2>&1; cat $OUT/ab_D.log
R=$(pwd); B=$R/abtest/lib_h2h; rm -f $B
hipcc -O2 -I$R/include -o $B $R/scripts/lib_h2h.cpp -L$R/gomel_amd -lgomel_hip -Wl,-rpath,$R/gomel_amd || exit 1
( echo "default plan: $($B 256 0 0 | tail -1)"
for sh in "1,2,2,2,1" "1,2,4,4,4,1" "1,2,3,4,3,2,1" "1,2,2,2,2,2,2,2,1" "1,1,2,2,2,2,2,2,1,1" "1,2,4,4,2,2,1" "1,3,3,3,3,2,1" "2,4,4,4,2" "1,2,4,6,2,1" "1,4,4,4,2,1" "1,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,1"; do echo "shares $sh: $(GOMEL_CHUNK_SHARES=$sh $B 256 0 0 | tail -1)"; done
echo "default plan again: $($B 256 0 0 | tail -1)" ) > $OUT/chunk_plans_T.log 2>&1; cat $OUT/chunk_plans_T.log
echo done
