#!/bin/bash
# rocprofv3 evidence for ONE configuration (GPU box, repo root): kernel-trace stats + the FETCH_SIZE / WRITE_SIZE PMC
# passes (each in a run of its own; only --kernel-trace beside --pmc) of tools/prof_step.py, merged into an HBM report.
#   bash tools/config_report.sh <tag> <steps> <batch> <W> <H> <forest> [s D]   ->  gpurun_out/<tag>/{kernel_stats.csv,hbm_report.json}
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
# the forest path (5th argument) relative to the repo root: the profiler runs from /tmp
args=("$@"); if [ ${#args[@]} -ge 5 ] && [ "${args[4]#/}" = "${args[4]}" ]; then args[4]="$R/${args[4]}"; fi
set -- "${args[@]}"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/cr_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cr_$TAG/trace -- python3 "$R/tools/prof_step.py" "$@" > "$O/trace.log" 2>&1 || echo "trace run failed"
cp "$(find /tmp/cr_$TAG/trace -name '*kernel_stats.csv' | head -1)" "$O/kernel_stats.csv" 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/cr_$TAG/$c -- python3 "$R/tools/prof_step.py" "$@" > "$O/$c.log" 2>&1 || echo "$c run failed"
  cp "$(find /tmp/cr_$TAG/$c -name '*counter_collection.csv' | head -1)" "$O/$c.csv" 2>/dev/null
done
cd "$R"
python3 tools/make_hbm_report.py "$O" "$@" > "$O/hbm_report.json" && echo "report: $O/hbm_report.json"
