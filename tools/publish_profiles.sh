#!/bin/bash
# Copies what tools/collect_profiles.sh <tag> left under gpurun_out/<tag>/ (scratch) into profiles/ (tracked) under the
# names profiles/README.md lists.  usage: bash tools/publish_profiles.sh r04_a
set -eu
TAG=$1
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/$TAG
P=$R/profiles
mkdir -p $P/${TAG%%_*}_pmc
cp $O/bench.json $P/${TAG}_bench.json
cp $O/kernel_stats.csv $P/${TAG}_kernel_stats.csv
cp $O/${TAG}_pmc_summary.json $P/${TAG%%_*}_pmc/${TAG}_pmc_summary.json
cp $O/traffic.json $P/traffic.json
cp $O/c3/hbm_report.json $P/${TAG}_c3_1080p_tau_hbm_report.json
cp $O/c3/kernel_stats.csv $P/${TAG}_c3_1080p_tau_kernel_stats.csv
cp $O/c5/hbm_report.json $P/${TAG}_c5_4k_hbm_report.json
cp $O/c5/kernel_stats.csv $P/${TAG}_c5_4k_kernel_stats.csv
cp $O/config_timings.json $P/${TAG}_config_timings.json
cp $O/global_kernel_stats.csv $P/${TAG}_global_kernel_stats.csv
cp $O/hashtable_kernel_stats.csv $P/${TAG}_hashtable_kernel_stats.csv
cp $O/dropin_printout.txt $P/${TAG}_dropin_printout.txt
cp $O/stamps.txt $P/${TAG}_phase_stamps.txt
ls -la $P | grep $TAG
