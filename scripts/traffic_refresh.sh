#!/bin/bash
# HBM traffic of the iteration kernels on the CURRENT build (profiles/traffic.json, which bench.py's roofline.traffic quotes): separate
# rocprofv3 --pmc passes for FETCH_SIZE and WRITE_SIZE (never combined with a trace), corrected as MI355X_MICROARCH.md "HBM" prescribes
# (FETCH_SIZE doubled on gfx950, WRITE_SIZE as read).  usage: scripts/traffic_refresh.sh OUTDIR ROUND   (one GPU call; ~4 minutes)
OUT=$1; RND=${2:-r05}; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
pass() { key=$1; ctr=$2; shift 2; (cd /tmp && rocprofv3 --pmc $ctr --output-format csv -d "$ROOT/$OUT/$key.$ctr" -- python3 "$ROOT/scripts/prof_run.py" "$@" > "$ROOT/$OUT/$key.$ctr.log" 2>&1); }
both() { key=$1; shift; pass $key FETCH_SIZE "$@" && pass $key WRITE_SIZE "$@"; }
GOMEL_FP64_ITERS=0  both cfg3:fused 256 400 6 T &&
GOMEL_FP64_ITERS=-1 both cfg3:fused_d 256 400 6 T &&
GOMEL_FP64_ITERS=0  both cfg4:fused 256 400 6 S &&
GOMEL_FP64_ITERS=-1 both cfg4:fused_d 256 400 6 S &&
GOMEL_FP64_ITERS=0  both cfg2:persistent 1 342 60 T &&
GOMEL_FP64_ITERS=-1 both cfg2:persistent_d 1 342 60 T &&
python3 scripts/traffic_json.py $OUT $RND > $OUT/traffic.json && cat $OUT/traffic.json
