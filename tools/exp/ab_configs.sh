#!/bin/bash
# A/B of opengpc_amd/libab_*.so variants over the BASELINE configurations (tools/config_timings.py) and 64 / 256 pairs
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
for lib in $R/opengpc_amd/libab_*.so; do
  name=$(basename $lib .so); name=${name#libab_}
  GPC_HIP_LIB=$lib python $R/tools/config_timings.py > $R/gpurun_out/ct_ab.log 2>&1
  python - "$name" <<'PY'
import json, sys, os
R = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
out = []
for e in json.load(open(os.path.join(R, "gpurun_out", "config_timings.json"))):
    c = e["config"]
    if "global" in c or "hashtable" in c or "single 1024" in c: continue
    k = e["kernel_us_per_launch"]
    out.append("%s %.4f (h %.1f j %.1f)" % (c.split()[0] + c.split()[1][:6] + ("T" if "Tau" in c else "Z"), e["ms_per_step"], k["k_hash"], k["k_row_join"]))
print("%-8s " % sys.argv[1] + " | ".join(out))
PY
  GPC_HIP_LIB=$lib python $R/tools/batch_sweep.py 64 256 2>/dev/null | python -c "
import sys, json
print('         ' + ' | '.join('%d pairs %.4f (h %.1f j %.1f)' % (d['pairs'], d['ms_per_step'], d['kernel_us']['k_hash'], d['kernel_us']['k_row_join']) for d in map(json.loads, sys.stdin)))"
done
done
