#!/bin/bash
# rocprofv3 kernel stats of the device-wide matcher modes (global sort / hash table), batch of 32.
# usage (GPU box): bash tools/prof_modes.sh global|hashtable  -> gpurun_out/<mode>_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-$(pwd)}
mode=${1:-global}
cd /tmp && export TMPDIR=/tmp
export GPC_PROF_GLOBAL=1
[ "$mode" = hashtable ] && { unset GPC_PROF_GLOBAL; export GPC_PROF_HASHTABLE=1; }
rm -rf /tmp/prof_$mode
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -- python3 $R/tools/prof_step.py 4 32 > $R/gpurun_out/prof_$mode.log 2>&1 || exit 1
f=$(find /tmp/prof_$mode -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/${mode}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-30s calls %3s avg %9.1f us  %6s%%" % (r['Name'].split('(')[0][-30:], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
