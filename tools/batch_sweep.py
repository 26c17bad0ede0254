#!/usr/bin/env python3
"""Step time and kernel split of the device-resident hot path over batch sizes (1024x436, defaultZeroForest, sparsematch
settings): what a rank's share of BASELINE configs[3] costs at N GPUs (256 / N pairs).  usage: python tools/batch_sweep.py
[batch sizes ...]  (environment knobs of the library apply: GPC_HIP_PRE_ROWS, GPC_HIP_HASH_TPW, ...)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import opengpc_amd as g  # noqa: E402
from opengpc_amd.synth import synth_batch  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 32, 64, 128, 256]
    W, H = 1024, 436
    dev = torch.device("cuda", 0)
    B = max(sizes)
    L, R = synth_batch(W, H, list(range(B)))
    d_L, d_R = torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)
    cap = (W - 26) * (H - 26)
    d_out = torch.empty((B, cap, 3), dtype=torch.int32, device=dev)
    d_counts = torch.zeros(B, dtype=torch.int32, device=dev)
    d_ncand = torch.zeros((B, 2), dtype=torch.int32, device=dev)
    ctx = g.Context(0)
    stream = torch.cuda.Stream(device=dev)
    ctx.set_stream(stream.cuda_stream)
    ctx.load_forest(os.path.join(ROOT, "forests", "defaultZeroForest.txt"), W, H)
    ctx.reserve(W, H, B)
    st = g.Settings.sparsematch()
    full = None
    for n in sizes:
        def step():
            ctx.match_batch_device(d_L.data_ptr(), d_R.data_ptr(), W, H, n, st, d_out.data_ptr(), cap, d_counts.data_ptr(),
                                   d_ncand.data_ptr())
        for _ in range(10):
            step()
        ctx.synchronize()
        reps = []
        for _ in range(15):
            t0 = time.perf_counter()
            for _ in range(20):
                step()
            ctx.synchronize()
            reps.append((time.perf_counter() - t0) / 20)
        dt = sorted(reps)[len(reps) // 2]
        ctx.enable_kernel_timing(True)
        ctx.reset_kernel_timing()
        for _ in range(10):
            step()
        kt = {k: round(1e3 * ms / c, 1) for k, (ms, c) in ctx.kernel_times().items() if c}
        ctx.enable_kernel_timing(False)
        rate = 2.0 * W * H * n / dt / 1e6
        if n == 256:
            full = rate
        print(json.dumps({"pairs": n, "ms_per_step": round(dt * 1e3, 4), "Mpix_per_s": round(rate, 1), "kernel_us": kt,
                          "kernels": {k: v for k, v in ctx.kernel_launch_names().items() if k in kt}}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
