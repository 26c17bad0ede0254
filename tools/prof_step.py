#!/usr/bin/env python3
"""Runs a few batched steps of the hot path without torch (fast start-up) -- the target of
rocprofv3 kernel-trace / PMC runs.  Usage: python3 tools/prof_step.py [steps] [batch] [W] [H] [forest] [s D]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import opengpc_amd as g  # noqa: E402
from opengpc_amd.synth import synth_batch  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    H = int(sys.argv[4]) if len(sys.argv) > 4 else 436
    forest = sys.argv[5] if len(sys.argv) > 5 else os.path.join(ROOT, "forests", "defaultZeroForest.txt")
    epipolar = os.environ.get("GPC_PROF_GLOBAL") is None
    ctx = g.Context(0)
    ctx.load_forest(forest, W, H)
    if len(sys.argv) > 7:   # BASELINE configurations with their own (s, D): pair i uses s + i, D
        from opengpc_amd.synth import synth_pair
        s0, D = int(sys.argv[6]), int(sys.argv[7])
        prs = [synth_pair(W, H, s0 + i, D) for i in range(B)]
        L, R = np.stack([p[0] for p in prs]), np.stack([p[1] for p in prs])
    else:
        L, R = synth_batch(W, H, list(range(B)))
    s = g.Settings.sparsematch()
    s.epipolar_mode = int(epipolar)
    s.use_hashtable = int(os.environ.get("GPC_PROF_HASHTABLE") is not None)
    cap = (W - 26) * (H - 26)
    # device-resident launches exactly like bench.py's (whole batch per launch, 12-byte supports left in HBM);
    # device memory through the HIP runtime the library itself is linked against
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")

    def dmalloc(nbytes):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0
        return p
    d_L, d_R = dmalloc(L.nbytes), dmalloc(R.nbytes)
    d_out, d_cnt, d_nc = dmalloc(B * cap * 12), dmalloc(B * 4), dmalloc(B * 8)
    assert hip.hipMemcpy(d_L, C.c_void_p(L.ctypes.data), C.c_size_t(L.nbytes), 1) == 0
    assert hip.hipMemcpy(d_R, C.c_void_p(R.ctypes.data), C.c_size_t(R.nbytes), 1) == 0
    for _ in range(steps):
        ctx.match_batch_device(d_L.value, d_R.value, W, H, B, s, d_out.value, cap, d_cnt.value, d_nc.value)
    ctx.synchronize()
    counts, ncand = np.empty(B, np.int32), np.empty((B, 2), np.int32)
    assert hip.hipMemcpy(C.c_void_p(counts.ctypes.data), d_cnt, C.c_size_t(B * 4), 2) == 0
    assert hip.hipMemcpy(C.c_void_p(ncand.ctypes.data), d_nc, C.c_size_t(B * 8), 2) == 0
    print("steps", steps, "pairs", B, "supports/pair", counts.mean(), "cand/pair", ncand.sum(1).mean())
    ctx.close()


if __name__ == "__main__":
    main()
