#!/bin/bash
# Three PMC passes (busy / instruction mix / LDS) for one kernel of the hash-table or non-epipolar mode, printed
# per launch.  usage (GPU box, repo root): GPC_PROF_HASHTABLE=1 bash tools/pmc_quick.sh k_ht_join [steps batch]
K=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcq
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
run() {
  name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$R/tools/prof_step.py" ${ARGS:-3 32} > "$OUT/$name.log" 2>&1 || return 1
}
ARGS="$*"
run sq_busy SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS &&
run sq_insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT &&
run sq_lds SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU
python3 - "$OUT" "$K" <<'PY'
import csv, glob, sys, collections
out, k = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(out + "/*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if k in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for c in acc:
        print("%-26s %14.0f per launch (%d launches)" % (c, acc[c] / n[c], n[c]))
PY
