#!/bin/bash
OUT=gpurun_out/r03j; mkdir -p $OUT; export TMPDIR=/tmp
R=$(pwd); B=$R/abtest/lib_h2h; rm -f $B
hipcc -O2 -I$R/include -o $B $R/scripts/lib_h2h.cpp -L$R/gomel_amd -lgomel_hip -Wl,-rpath,$R/gomel_amd || exit 1
( echo "default plan: $($B 256 0 1 | tail -1)"
for n in 1 2 3 4 6 8; do echo "equal $n: $(GOMEL_CHUNKS=$n $B 256 0 1 | tail -1)"; done
for sh in "1,2,1" "1,2,2,1" "1,2,2,2,1" "1,1,1" "2,2,1" "3,2,1" "1,3,2,1" "1,2,2,2,2,2,1"; do echo "shares $sh: $(GOMEL_CHUNK_SHARES=$sh $B 256 0 1 | tail -1)"; done
echo "default plan again: $($B 256 0 1 | tail -1)" ) > $OUT/chunk_plans_S.log 2>&1; cat $OUT/chunk_plans_S.log
echo done
