#!/bin/bash
# Round 5, GPU call 13: larger held-out validation of the frozen schedule on the final build (65 536 random spectrograms per geometry, seeds from 11 000 000 on)
# and a 180 s soak
OUT=gpurun_out/r05c13; mkdir -p $OUT
export CAL_SEED0=11000000
timeout -k 10 400 python3 scripts/calibrate_schedule.py T random 256 60:35 > $OUT/val6_T_random_65536.txt 2>&1; tail -2 $OUT/val6_T_random_65536.txt
timeout -k 10 200 python3 scripts/calibrate_schedule.py S random 256 60:35 > $OUT/val6_S_random_65536.txt 2>&1; tail -2 $OUT/val6_S_random_65536.txt
timeout -k 10 300 python3 scripts/calibrate_schedule.py D random 256 60:50 > $OUT/val6_D_random_65536.txt 2>&1; tail -2 $OUT/val6_D_random_65536.txt
timeout -k 10 300 python3 scripts/calibrate_schedule.py Z random 256 60:44 > $OUT/val6_Z_random_65536.txt 2>&1; tail -2 $OUT/val6_Z_random_65536.txt
timeout -k 10 300 python3 scripts/soak.py 180 > $OUT/soak_180.txt 2>&1; tail -2 $OUT/soak_180.txt
