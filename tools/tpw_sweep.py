import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
import opengpc_amd as g
from opengpc_amd.synth import synth_pair
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
dev = torch.device("cuda", 0)
W, H, B, forest = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
ctx = g.Context(0)
ctx.load_forest(os.path.join(ROOT, "forests", forest), W, H)
L, R = synth_pair(W, H, 1, 40)
dL = torch.from_numpy(np.stack([L]*B)).to(dev); dR = torch.from_numpy(np.stack([R]*B)).to(dev)
cap = (W-26)*(H-26)
out = torch.empty((B, cap, 3), dtype=torch.int32, device=dev); cnt = torch.zeros(B, dtype=torch.int32, device=dev); nc = torch.zeros((B,2), dtype=torch.int32, device=dev)
s = g.Settings.sparsematch()
def step(): ctx.match_batch_device(dL.data_ptr(), dR.data_ptr(), W, H, B, s, out.data_ptr(), cap, cnt.data_ptr(), nc.data_ptr())
for _ in range(3): step()
ctx.synchronize()
ctx.enable_kernel_timing(True); ctx.reset_kernel_timing()
for _ in range(8): step()
kt = {k: round(1e3*ms/n,1) for k,(ms,n) in ctx.kernel_times().items() if n}
print(os.environ.get("GPC_HIP_HASH_TPW","auto"), kt)
