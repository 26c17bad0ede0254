#!/bin/bash
# On the GPU box: benches every tools/variants/libgpc_<name>.so given (built by tools/variant_local.sh).
# usage: bash tools/variant_run.sh name1 name2 ...   (BENCH_ARGS, VARIANT_ENV pass through)
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out"
for name in "$@"; do
  env GPC_HIP_LIB="$R/tools/variants/libgpc_$name.so" $VARIANT_ENV timeout -k 10 200 python "$R/bench.py" --steps 20 --windows 8 --no-cpu-baseline --no-extras ${VERIFY:---no-verify} $BENCH_ARGS > "$R/gpurun_out/variant_$name.json" 2> "$R/gpurun_out/variant_$name.err" || { echo "$name: bench failed"; tail -3 "$R/gpurun_out/variant_$name.err"; continue; }
  python - "$name" "$R/gpurun_out/variant_$name.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
k = d["roofline"]["kernels"]
print("%-16s %8.1f Mpix/s  step %.4f ms  " % (sys.argv[1], d["value"], d["ms_per_step"]) + "  ".join("%s=%.1f" % (n.replace("k_", ""), v["avg_us"]) for n, v in k.items()))
PY
done
