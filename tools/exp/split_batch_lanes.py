#!/usr/bin/env python3
"""Experiment: a SMALL batch (32 pairs: a GPU's share of BASELINE configs[3] at 8 GPUs) as ONE call against the same batch
split into 2 / 4 calls over the library's two lanes (gpc_hip_set_pipeline(2)): do the halves' kernels fill each other's tails?
usage: python tools/exp/split_batch_lanes.py [pairs]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import opengpc_amd as g  # noqa: E402
from opengpc_amd.synth import synth_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
W, H = 1024, 436
dev = torch.device("cuda", 0)
L, R = synth_batch(W, H, list(range(B)))
dL, dR = torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)
cap = (W - 26) * (H - 26)
s = g.Settings.sparsematch()
ref = None
for lanes, parts in ((1, 1), (2, 2), (2, 4), (1, 2)):
    c = g.Context(0)
    c.load_forest(os.path.join(ROOT, "forests", "defaultZeroForest.txt"), W, H)
    c.set_pipeline(lanes)
    out = torch.zeros((B, cap, 3), dtype=torch.int32, device=dev)
    cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    nc = torch.zeros((B, 2), dtype=torch.int32, device=dev)
    per = B // parts

    def step():
        for i in range(parts):
            o = i * per
            c.match_batch_device(dL[o:].data_ptr(), dR[o:].data_ptr(), W, H, per, s, out[o:].data_ptr(), cap, cnt[o:].data_ptr(), nc[o:].data_ptr())
    for _ in range(10):
        step()
    c.synchronize()
    tt = []
    for _ in range(15):
        t0 = time.perf_counter()
        for _ in range(20):
            step()
        c.synchronize()
        tt.append((time.perf_counter() - t0) / 20)
    dt = sorted(tt)[len(tt) // 2]
    res = (out.cpu().numpy(), cnt.cpu().numpy())
    if ref is None:
        ref = res
    same = bool(np.array_equal(res[0], ref[0]) and np.array_equal(res[1], ref[1]))
    print("%d pairs as %d call(s) of %d, %d lane(s): %.4f ms  %.1f Gpix/s  identical: %s" % (B, parts, per, lanes, dt * 1e3, 2.0 * W * H * B / dt / 1e9, same), flush=True)
    c.close()
