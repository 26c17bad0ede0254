#!/bin/bash
# Collects the evidence committed under profiles/ for one state of the tree (run on the GPU box from the repo root):
#   bash tools/collect_profiles.sh <tag>      ->  gpurun_out/<tag>/...
# bench line, rocprofv3 kernel stats of the same command, PMC passes (one counter group per run, only --kernel-trace
# beside --pmc), HBM reports of BASELINE configs[2] / configs[4], per-configuration timings, the secondary matcher
# modes, the drop-in's printout, phase stamps.
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_bench && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 $R/bench.py --no-cpu-baseline --no-extras --windows 20 > $O/prof_bench.json 2> $O/prof_bench.err; cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv ); echo "rocprof rc=$?"
( cd /tmp && export TMPDIR=/tmp && bash $R/tools/pmc_collect.sh $O/pmc 3 256 > $O/pmc.log 2>&1 ); tail -1 $O/pmc.log
python tools/make_traffic.py $O/pmc ${TAG} 1024 436 256 > $O/traffic.txt 2>&1
cp profiles/traffic.json $O/traffic.json; cp profiles/${TAG%%_*}_pmc/${TAG}_pmc_summary.json $O/ 2>/dev/null
bash tools/config_report.sh ${TAG}_c3 10 8 1920 1080 forests/defaultTauForest.txt 1 40 > $O/c3.log 2>&1; cp -r gpurun_out/${TAG}_c3 $O/c3 2>/dev/null
bash tools/config_report.sh ${TAG}_c5 10 1 3840 2160 forests/stress16x20Forest.txt 2 64 > $O/c5.log 2>&1; cp -r gpurun_out/${TAG}_c5 $O/c5 2>/dev/null
python tools/config_timings.py > $O/config_timings.log 2>&1; cp gpurun_out/config_timings.json $O/ 2>/dev/null; echo "configs rc=$?"
bash tools/prof_modes.sh global > $O/prof_global.txt 2>&1; cp gpurun_out/global_kernel_stats.csv $O/ 2>/dev/null
bash tools/prof_modes.sh hashtable > $O/prof_hashtable.txt 2>&1; cp gpurun_out/hashtable_kernel_stats.csv $O/ 2>/dev/null
python tools/dropin_printout.py > $O/dropin_printout.txt 2>&1
python tools/stamp_profile.py 256 > $O/stamps.txt 2>&1
echo "collect done"
