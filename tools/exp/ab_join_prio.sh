#!/bin/bash
# A/B of opengpc_amd/libab_*.so variants on the join's time: python tools/batch_sweep.py 32 256 per variant, the list twice
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
for round in 1 2; do
for lib in $R/opengpc_amd/libab_*.so; do
  name=$(basename $lib .so); name=${name#libab_}
  GPC_HIP_LIB=$lib python $R/tools/batch_sweep.py 32 256 2>/dev/null | python -c "
import sys, json
out = []
for l in sys.stdin:
    d = json.loads(l)
    k = d['kernel_us']
    out.append('%3d pairs: step %.4f ms, pre %.1f hash %.1f join %.1f' % (d['pairs'], d['ms_per_step'], k.get('k_preprocess', 0), k.get('k_hash', 0), k.get('k_row_join', 0)))
print('%-10s ' % '$name' + ' | '.join(out))
" | tee -a $R/gpurun_out/ab_join_prio.txt
done
done
