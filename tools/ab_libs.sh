#!/bin/bash
# A/B of prebuilt library variants on one box: bash tools/ab_libs.sh name1=path1.so name2=path2.so ... (each benched in turn,
# the whole list twice; BENCH_ARGS as in variant_bench.sh).  Build the variants in-tree (opengpc_amd/libab_*.so travels to the
# GPU box; *.so is git-ignored).
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
for spec in "$@"; do
  name=${spec%%=*}; lib=${spec#*=}
  GPC_HIP_LIB=$R/$lib timeout -k 10 200 python "$R/bench.py" --steps 20 --windows 8 --no-cpu-baseline --no-extras --no-verify $BENCH_ARGS > "$R/gpurun_out/ab_$name.json" 2> "$R/gpurun_out/ab_$name.err" || { echo "$name: bench failed"; tail -3 "$R/gpurun_out/ab_$name.err"; continue; }
  python - "$name" "$R/gpurun_out/ab_$name.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
k = d["roofline"]["kernels"]
print("%-14s %8.1f Mpix/s  step %.4f ms  " % (sys.argv[1], d["value"], d["ms_per_step"]) + "  ".join("%s=%.1f" % (n.replace("k_", ""), v["avg_us"]) for n, v in k.items()))
PY
done
done
