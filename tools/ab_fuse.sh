#!/bin/bash
# A/B of the fused join + output against the two-launch path and of its tuning knobs (one bench line each).
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out"
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 150 python "$R/bench.py" --steps 20 --windows 8 --no-cpu-baseline --no-extras $BENCH_ARGS > "$R/gpurun_out/ab_$name.json" 2> "$R/gpurun_out/ab_$name.err" || { echo "$name: bench failed"; tail -3 "$R/gpurun_out/ab_$name.err"; return; }
  python - "$name" "$R/gpurun_out/ab_$name.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
k = d["roofline"]["kernels"]
print("%-18s %8.1f Mpix/s  step %.4f ms  " % (sys.argv[1], d["value"], d["ms_per_step"]) + "  ".join("%s=%.1f" % (n.replace("k_", ""), v["avg_us"]) for n, v in k.items()))
PY
}
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  run $name $envs
done
