#!/bin/bash
# Round 5, GPU call 11: the frozen schedule with the round's fp64 arithmetic, held out (seeds from 9 000 000 on): 32 x 256 = 8 192 random spectrograms per
# geometry at 60 iterations, the default's k against 64-bit mode; the Z geometry (N 2048 / hop 400) at the interpolated 44
OUT=gpurun_out/r05c11; mkdir -p $OUT
export CAL_SEED0=9000000
timeout -k 10 280 python3 scripts/calibrate_schedule.py T random 32 60:35 > $OUT/val5_T_random_8192.txt 2>&1; tail -2 $OUT/val5_T_random_8192.txt
timeout -k 10 200 python3 scripts/calibrate_schedule.py S random 32 60:35 > $OUT/val5_S_random_8192.txt 2>&1; tail -2 $OUT/val5_S_random_8192.txt
timeout -k 10 280 python3 scripts/calibrate_schedule.py D random 32 60:50 > $OUT/val5_D_random_8192.txt 2>&1; tail -2 $OUT/val5_D_random_8192.txt
timeout -k 10 280 python3 scripts/calibrate_schedule.py Z random 32 60:44 > $OUT/val5_Z_random_8192.txt 2>&1; tail -2 $OUT/val5_Z_random_8192.txt
