#!/usr/bin/env python3
"""Times the hot path on every BASELINE.json configuration (device-resident inputs, HIP-event kernel
split), for profiles/.  Not the headline bench (bench.py); parity for these configs is in tests/."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import opengpc_amd as g  # noqa: E402
from opengpc_amd.synth import synth_pair  # noqa: E402

CONFIGS = [
    ("C2 single 1024x436 Zero", 1024, 436, "defaultZeroForest.txt", 0, 24, 1, True, False),
    ("C2 batch32 1024x436 Zero", 1024, 436, "defaultZeroForest.txt", 0, 24, 32, True, False),
    ("C2 batch32 1024x436 Tau", 1024, 436, "defaultTauForest.txt", 0, 24, 32, True, False),
    ("C3 single 1920x1080 Tau", 1920, 1080, "defaultTauForest.txt", 1, 40, 1, True, False),
    ("C3 batch8 1920x1080 Tau", 1920, 1080, "defaultTauForest.txt", 1, 40, 8, True, False),
    ("C5 single 3840x2160 stress16x20 (first 32 tests)", 3840, 2160, "stress16x20Forest.txt", 2, 64, 1, True, False),
    ("C2 single 1024x436 Zero global (non-epipolar)", 1024, 436, "defaultZeroForest.txt", 0, 24, 1, False, False),
    ("C2 single 1024x436 Zero epipolar hashtable", 1024, 436, "defaultZeroForest.txt", 0, 24, 1, True, True),
    ("C2 batch32 1024x436 Zero global (non-epipolar)", 1024, 436, "defaultZeroForest.txt", 0, 24, 32, False, False),
    ("C2 batch32 1024x436 Zero epipolar hashtable", 1024, 436, "defaultZeroForest.txt", 0, 24, 32, True, True),
    ("C3 batch8 1920x1080 Tau global (non-epipolar)", 1920, 1080, "defaultTauForest.txt", 1, 40, 8, False, False),
    ("C3 batch8 1920x1080 Tau epipolar hashtable", 1920, 1080, "defaultTauForest.txt", 1, 40, 8, True, True),
]


def main():
    dev = torch.device("cuda", 0)
    out = []
    only = os.environ.get("GPC_CONFIGS")   # substring filter (GPC_CONFIGS="C5 single"): the other configurations are skipped
    for name, W, H, forest, s, D, B, epi, ht in CONFIGS:
        if only and only not in name:
            continue
        ctx = g.Context(0)
        stream = torch.cuda.Stream(device=dev)
        ctx.set_stream(stream.cuda_stream)
        ctx.load_forest(os.path.join(ROOT, "forests", forest), W, H)
        L, R = synth_pair(W, H, s, D)
        d_L = torch.from_numpy(np.stack([L] * B)).to(dev)
        d_R = torch.from_numpy(np.stack([R] * B)).to(dev)
        cap = (W - 26) * (H - 26)
        d_out = torch.empty((B, cap, 3), dtype=torch.int32, device=dev)
        d_counts = torch.zeros(B, dtype=torch.int32, device=dev)
        d_ncand = torch.zeros((B, 2), dtype=torch.int32, device=dev)
        st = g.Settings(5, 128, 0, epi, ht, 1)

        def step():
            ctx.match_batch_device(d_L.data_ptr(), d_R.data_ptr(), W, H, B, st, d_out.data_ptr(), cap,
                                   d_counts.data_ptr(), d_ncand.data_ptr())
        for _ in range(3):
            step()
        ctx.synchronize()
        steps = 20 if W * H * B < 3e7 else 8
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ctx.enable_kernel_timing(True)
        ctx.reset_kernel_timing()
        for _ in range(5):
            step()
        kt = {k: round(1e3 * ms / n, 1) for k, (ms, n) in ctx.kernel_times().items() if n}
        ctx.enable_kernel_timing(False)
        rec = {"config": name, "pairs": B, "ms_per_step": round(dt * 1e3, 4),
               "Mpix_per_s": round(2.0 * W * H * B / dt / 1e6, 1), "supports_per_pair": int(d_counts[0].item()),
               "candidates_per_pair": int(d_ncand[0].sum().item()), "kernel_us_per_launch": kt}
        out.append(rec)
        print(json.dumps(rec), flush=True)
        ctx.close()
        del d_L, d_R, d_out
        torch.cuda.empty_cache()
    with open(os.path.join(ROOT, "gpurun_out", "config_timings.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
