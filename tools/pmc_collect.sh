#!/bin/bash
# Collects rocprofv3 PMC counters for the hot-path kernels in separate passes (one counter
# group per run, --pmc never combined with trace domains other than --kernel-trace).
# Usage (on the GPU box, from the repo root): bash tools/pmc_collect.sh <outdir> [prof_step args...]
set -u
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p "$OUT"
cd /tmp
run() {
  name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$R/tools/prof_step.py" ${ARGS} > "$OUT/$name.log" 2>&1 || return 1
}
ARGS="$*"
run sq_busy SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS &&
run sq_insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT &&
run sq_lds SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE &&
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
echo "pmc_collect done rc=$?"
