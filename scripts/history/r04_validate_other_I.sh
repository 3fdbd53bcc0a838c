#!/bin/bash
# Round 4: the automatic schedule at iteration counts OTHER than 60 (table entries and interpolated ones) on held-out utterances
# (seeds from 2 000 000 on).  usage (GPU box): bash scripts/r04_validate_other_I.sh OUTDIR "GEOM CLASS BLOCKS" ["GEOM CLASS BLOCKS" ...]
out=$1; shift; mkdir -p $out
export CAL_SEED0=2000000
for spec in "$@"; do
  set -- $spec
  f=$out/val_otherI_${1}_${2}_$(($3*256)).txt; : > $f
  for ik in 5:3 10:6 15:8 20:10 25:18 30:25 40:30 50:33 80:58 100:80 150:130; do
    python scripts/calibrate_schedule.py $1 $2 $3 $ik >> $f 2>&1 || exit 1
  done
  echo "$spec done"; grep " I=" $f | cut -c1-150
done
