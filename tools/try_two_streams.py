#!/usr/bin/env python3
"""Experiment: does splitting a 32-pair batch over two contexts/streams (so the HBM-bound and the
compute/LDS-bound kernels of different halves overlap) beat one stream?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import opengpc_amd as g
from opengpc_amd.synth import synth_batch

W, H, B = 1024, 436, 32
dev = torch.device("cuda", 0)
L, R = synth_batch(W, H, list(range(B)))
dL, dR = torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)
cap = (W - 26) * (H - 26)
out = torch.empty((B, cap, 3), dtype=torch.int32, device=dev)
cnt = torch.zeros(B, dtype=torch.int32, device=dev)
nc = torch.zeros((B, 2), dtype=torch.int32, device=dev)
s = g.Settings.sparsematch()
for nsplit in (1, 2, 4):
    ctxs = []
    for i in range(nsplit):
        c = g.Context(0)
        c.load_forest(os.path.join(ROOT, "forests", "defaultZeroForest.txt"), W, H)
        c.reserve(W, H, B // nsplit)
        ctxs.append(c)
    per = B // nsplit
    def step():
        for i, c in enumerate(ctxs):
            o = i * per
            c.match_batch_device(dL[o:].data_ptr(), dR[o:].data_ptr(), W, H, per, s, out[o:].data_ptr(), cap,
                                 cnt[o:].data_ptr(), nc[o:].data_ptr())
    for _ in range(5): step()
    for c in ctxs: c.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n): step()
    for c in ctxs: c.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("streams %d: %.4f ms/step  %.1f Gpix/s  (checksum %d)" % (nsplit, dt * 1e3, 2.0 * W * H * B / dt / 1e9, int(cnt.sum().item())))
    for c in ctxs: c.close()

print("-- alternating full-batch steps over k contexts (pipelining across steps)")
outs = [torch.empty((B, cap, 3), dtype=torch.int32, device=dev) for _ in range(3)]
for k in (1, 2, 3):
    ctxs = []
    for i in range(k):
        c = g.Context(0)
        c.load_forest(os.path.join(ROOT, "forests", "defaultZeroForest.txt"), W, H)
        c.reserve(W, H, B)
        ctxs.append(c)
    def step(i):
        c = ctxs[i % k]
        c.match_batch_device(dL.data_ptr(), dR.data_ptr(), W, H, B, s, outs[i % k].data_ptr(), cap, cnt.data_ptr(), nc.data_ptr())
    for i in range(6): step(i)
    for c in ctxs: c.synchronize()
    n = 60
    t0 = time.perf_counter()
    for i in range(n): step(i)
    for c in ctxs: c.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("contexts %d: %.4f ms/step  %.1f Gpix/s" % (k, dt * 1e3, 2.0 * W * H * B / dt / 1e9))
    for c in ctxs: c.close()
