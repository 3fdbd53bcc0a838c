#!/bin/bash
# Round 4: calibration of the automatic precision schedule, 16 blocks x 256 = 4096 utterances per geometry and input class.
# usage (GPU box): bash scripts/r04_calibrate.sh OUTDIR [BLOCKS]
out=${1:-gpurun_out/r04cal}; blocks=${2:-16}
mkdir -p $out
GRID="2:0,1 5:0,1,2,3 10:0,1,2,3,4,6 20:0,4,6,8,10,12,16 30:0,5,10,15,20,25 40:0,10,15,20,25,30 60:0,10,16,20,22,24,26,28,30,32,35,40,45 100:0,20,30,40,50,60,70,80"
for g in T S D; do for c in random signal; do
  python scripts/calibrate_schedule.py $g $c $blocks $GRID > $out/cal_${g}_${c}_$((blocks*256)).txt 2>&1 || exit 1
  echo "$g $c done"; tail -2 $out/cal_${g}_${c}_$((blocks*256)).txt | cut -c1-160
done; done
