#!/bin/bash
# Round 5, GPU call 15: the pass-3 pairing in the generic and any-hop kernels: the whole GPU suite (bitwise / 1e-12 cross-checks between kernel paths) and off-grid hop timings
OUT=gpurun_out/r05c15; mkdir -p $OUT
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $OUT/tests_gpu.log 2>&1; rc=$?; echo "gpu tests rc $rc"; tail -4 $OUT/tests_gpu.log
