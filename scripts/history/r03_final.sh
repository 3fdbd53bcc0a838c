#!/bin/bash
# round 3: final build -- any-hop fp64 tests after the revert, fp64 hop probe, then everything profiles/r03/ is made of
OUT=gpurun_out/r03r; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_round3.py -q -k "any_hop or precision" 2>&1 | tail -2
(GOMEL_PRECISION=64 python3 scripts/hop_probe.py; GOMEL_PRECISION=64 GOMEL_PATH=1 python3 scripts/hop_probe.py) 2>&1 | grep -v amdgpu.ids > $OUT/hop_probe_fp64.txt; grep "N 4096" $OUT/hop_probe_fp64.txt
bash scripts/profile_r03.sh $OUT/prof
echo done
