#!/bin/bash
# Round 5: the whole GPU suite on the GPU box, then the round's profiles (scripts/profile_r05.sh)
mkdir -p gpurun_out/r05c4
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05c4/tests_gpu.log 2>&1; rc=$?; echo "gpu tests rc $rc"; tail -5 gpurun_out/r05c4/tests_gpu.log
[ $rc -eq 0 ] && timeout -k 10 1100 bash scripts/profile_r05.sh gpurun_out/r05p
