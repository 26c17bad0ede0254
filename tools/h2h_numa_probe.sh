# h2h_numa_probe.sh -- one 256-pair gpc_hip_match_batch call (page-locked buffers) with the calling thread / the whole process
# on the GPU's NUMA node or on the other one: stage times (gpc_hip_batch_stages), where the pages lie, where the workers ran,
# and whether the call hopped to a feeder thread on the GPU's node (gpc_hip_fed_calls).  usage: bash tools/h2h_numa_probe.sh
set -e
mkdir -p gpurun_out/r05d
N1=$(cat /sys/devices/system/node/node1/cpulist)
N0=$(cat /sys/devices/system/node/node0/cpulist)
echo "node0 $N0 / node1 $N1"
cat > /tmp/h2h.py <<'PY'
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import opengpc_amd as g
from opengpc_amd.synth import synth_batch
from opengpc_amd.hostinfo import cpu_nodes, current_cpu, pages_nodes, stage_summary
B, W, H = 256, 1024, 436
ctx = g.Context(0)          # (created with the process's whole affinity: the workers find the GPU's node)
ctx.load_forest("forests/defaultZeroForest.txt", W, H)
s = g.Settings.sparsematch()
if os.environ.get("GPC_PIN_CALLER"):   # only the calling thread is held on one CPU
    os.sched_setaffinity(0, {int(os.environ["GPC_PIN_CALLER"])})
L, R = synth_batch(W, H, list(range(B)))
cap = 300000
Lp, Rp = ctx.pinned_empty(L.shape, np.uint8), ctx.pinned_empty(R.shape, np.uint8)
Lp[:] = L; Rp[:] = R
out = ctx.pinned_empty((B, cap), g.SUPPORT_DTYPE)
for _ in range(5):
    ctx.match_batch(Lp, Rp, s, cap, out=out)
tt, st = [], []
for _ in range(11):
    t0 = time.perf_counter(); ctx.match_batch(Lp, Rp, s, cap, out=out); tt.append(time.perf_counter() - t0); st.append(ctx.batch_stages())
tt.sort()
print(json.dumps({"tag": sys.argv[1], "ms": round(tt[5] * 1e3, 3), "stages": stage_summary(st), "cpu": current_cpu(), "node": cpu_nodes().get(current_cpu()),
                  "pages": {"out": pages_nodes(out), "images": pages_nodes(Lp)}, "workers": sorted(set(cpu_nodes().get(c) for c in ctx.worker_cpus())), "worker_cpus": sorted(ctx.worker_cpus()),
                  "gpu_node": ctx.L.gpc_hip_host_numa_node(ctx.h), "fed_calls": ctx.L.gpc_hip_fed_calls(ctx.h)}))
ctx.close()
PY
python /tmp/h2h.py unpinned_workers_one_per_ccd | tee -a gpurun_out/r05d/h2h_numa.txt
GPC_HIP_NO_CCD_SPREAD=1 python /tmp/h2h.py unpinned_workers_where_the_scheduler_puts_them | tee -a gpurun_out/r05d/h2h_numa.txt
python /tmp/h2h.py unpinned_workers_one_per_ccd | tee -a gpurun_out/r05d/h2h_numa.txt
GPC_HIP_NO_CCD_SPREAD=1 python /tmp/h2h.py unpinned_workers_where_the_scheduler_puts_them | tee -a gpurun_out/r05d/h2h_numa.txt
taskset -c $N0 python /tmp/h2h.py process_confined_to_node0 | tee -a gpurun_out/r05d/h2h_numa.txt
GPC_PIN_CALLER=$(( $(echo $N0 | cut -d, -f1 | cut -d- -f1) + 9 )) python /tmp/h2h.py caller_thread_on_node0_feeder | tee -a gpurun_out/r05d/h2h_numa.txt
GPC_HIP_NO_FEEDER=1 GPC_PIN_CALLER=$(( $(echo $N0 | cut -d, -f1 | cut -d- -f1) + 9 )) python /tmp/h2h.py caller_thread_on_node0_no_feeder | tee -a gpurun_out/r05d/h2h_numa.txt
GPC_PIN_CALLER=$(( $(echo $N1 | cut -d, -f1 | cut -d- -f1) + 9 )) python /tmp/h2h.py caller_thread_on_node1_feeder | tee -a gpurun_out/r05d/h2h_numa.txt
GPC_HIP_NO_FEEDER=1 GPC_PIN_CALLER=$(( $(echo $N1 | cut -d, -f1 | cut -d- -f1) + 9 )) python /tmp/h2h.py caller_thread_on_node1_no_feeder | tee -a gpurun_out/r05d/h2h_numa.txt
taskset -c $N1 python /tmp/h2h.py caller_on_node1 | tee -a gpurun_out/r05d/h2h_numa.txt
GPC_HIP_NO_NUMA_BIND=1 taskset -c $N1 python /tmp/h2h.py caller_and_workers_on_node1 | tee -a gpurun_out/r05d/h2h_numa.txt
