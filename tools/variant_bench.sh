#!/bin/bash
# Builds variants of libgpc_hip.so with different -D settings (on the GPU box) and benches each.
# usage: bash tools/variant_bench.sh "name1:-DHT_Y=32 -DHT_THREADS=512" "name2:..." ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out"
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags -o /tmp/libgpc_$name.so "$R/opengpc_amd/csrc/gpc_hip.hip" 2>/dev/null || { echo "$name: build failed"; continue; }
  GPC_HIP_LIB=/tmp/libgpc_$name.so timeout -k 10 200 python "$R/bench.py" --steps 20 --windows 8 --no-cpu-baseline --no-extras --no-verify $BENCH_ARGS > "$R/gpurun_out/variant_$name.json" 2> "$R/gpurun_out/variant_$name.err" || { echo "$name: bench failed"; tail -3 "$R/gpurun_out/variant_$name.err"; continue; }
  python - "$name" "$R/gpurun_out/variant_$name.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
k = d["roofline"]["kernels"]
print("%-14s %8.1f Mpix/s  step %.4f ms  " % (sys.argv[1], d["value"], d["ms_per_step"]) + "  ".join("%s=%.1f" % (n.replace("k_", ""), v["avg_us"]) for n, v in k.items()))
PY
done
