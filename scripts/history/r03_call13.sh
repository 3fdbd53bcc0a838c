#!/bin/bash
OUT=gpurun_out/r03u; mkdir -p $OUT; export TMPDIR=/tmp
R=$(pwd); B=$R/abtest/lib_h2h; rm -f $B
hipcc -O2 -I$R/include -o $B $R/scripts/lib_h2h.cpp -L$R/gomel_amd -lgomel_hip -Wl,-rpath,$R/gomel_amd || exit 1
( echo "default: $($B 256 0 0 | tail -1)"
for sf in 8 10 12 16 20 25 32; do echo "GOMEL_SEG_FRAMES=$sf: $(GOMEL_SEG_FRAMES=$sf $B 256 0 0 | tail -1)"; done
echo "default again: $($B 256 0 0 | tail -1)" ) 2>&1 | tee $OUT/seg_frames_h2h.txt
