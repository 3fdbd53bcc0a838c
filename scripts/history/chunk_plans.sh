#!/bin/bash
# Host-to-host time of one 256 x 400-frame gomel_from_mel_batch call from a C caller (scripts/lib_h2h.cpp, pinned buffers) under
# different chunk plans of the library's upload / compute / download pipeline.  usage: scripts/chunk_plans.sh [geometry 0=T 1=S 2=D 3=E]
R=${GRAFT_REPO_ROOT:-$(pwd)}; g=${1:-0}
B=$R/abtest/lib_h2h
if [ ! -x $B ]; then mkdir -p $R/abtest; hipcc -O2 -I$R/include -o $B $R/scripts/lib_h2h.cpp -L$R/gomel_amd -lgomel_hip -Wl,-rpath,$R/gomel_amd || exit 1; fi
echo "default plan: $($B 256 0 $g | tail -1)"
for n in 4 8 16; do echo "equal $n: $(GOMEL_CHUNKS=$n $B 256 0 $g | tail -1)"; done
for sh in "1,2,1" "1,2,2,1" "1,2,2,2,1" "1,2,2,2,2,2,2,2,1"; do echo "shares $sh: $(GOMEL_CHUNK_SHARES=$sh $B 256 0 $g | tail -1)"; done
