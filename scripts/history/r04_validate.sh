#!/bin/bash
# Round 4: the default schedule at 60 iterations on HELD-OUT utterances (seeds from 1 000 000 on; the calibration used 0 ... 4095): how often does
# k = 30 / 35 / 40 leave an utterance beyond 3e-5 / 1e-4?  usage (GPU box): bash scripts/r04_validate.sh OUTDIR
out=${1:-gpurun_out/r04val}; mkdir -p $out
export CAL_SEED0=1000000
for spec in "T random 128" "S random 128" "D random 64" "T signal 48" "S signal 64" "D signal 64"; do
  set -- $spec
  python scripts/calibrate_schedule.py $1 $2 $3 60:30,35,40 > $out/val_${1}_${2}_$(($3*256)).txt 2>&1 || exit 1
  echo "$spec done"; grep " I=" $out/val_${1}_${2}_$(($3*256)).txt | cut -c1-150
done
