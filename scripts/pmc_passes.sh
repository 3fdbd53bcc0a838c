#!/bin/bash
# rocprofv3 --pmc passes for the profiling target (scripts/prof_run.py: one cfg3-sized batch, 6 iterations); one counter set per run
# (never combined with trace domains).  usage: scripts/pmc_passes.sh OUTDIR [prof_run args...]   -> OUTDIR/<set>/..., OUTDIR/summary.txt
set -o pipefail
OUT=$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$(pwd)
run() { name=$1; shift; (cd /tmp && rocprofv3 --pmc "$@" --output-format csv -d "$ROOT/$OUT/$name" -- python3 "$ROOT/scripts/prof_run.py" $ARGS > "$ROOT/$OUT/$name.log" 2>&1); }
ARGS="$*"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run sq3 SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL
run sq4 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_LDS_UNALIGNED_STALL
run fetch FETCH_SIZE
run write WRITE_SIZE
run clk GRBM_GUI_ACTIVE
run ta TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
python3 scripts/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
grep -h "k_gl_fused" "$OUT/summary.txt"
