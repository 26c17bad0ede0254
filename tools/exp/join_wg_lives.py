#!/usr/bin/env python3
"""Experiment: every workgroup of ONE k_row_join_fused launch -- start, end (s_memrealtime, 100 MHz), rows it took, the CU
it ran on (-DGPC_WGLIFE build): how the persistent launch fills and drains.  The three live scalars cost this kernel registers:
the instrumented launch takes 677 us per 256 pairs where the product takes 521 -- read the SHAPE.  Round 5: all 2048 workgroups
start within 1 us; the ones that started first take 59 rows, the last 41 (oldest wave first; tickets are drawn, so nothing is
lost); the launch drains over its last 30 us (25 us before the end 1513 workgroups run, 15: 866, 10: 337, 6: 65) = ~2 % of
the launch at falling occupancy; at 32 pairs (6.4 rows per workgroup) the ends spread over 70 .. 90 us, ~5 %.  Nothing a
priority ladder like k_hash's would recover more than 2-4 us of.
usage: python tools/exp/join_wg_lives.py pairs [W H forest]"""
import ctypes as C
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "gpurun_out", "libgpc_hip_wglife.so")
import numpy as np  # noqa: E402


def main():
    B = int(sys.argv[1])
    W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1024, 436)
    forest = sys.argv[4] if len(sys.argv) > 4 else "defaultZeroForest.txt"
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DGPC_WGLIFE",
                           "-o", LIB, os.path.join(ROOT, "opengpc_amd", "csrc", "gpc_hip.hip")])
    import opengpc_amd.capi as capi
    capi.LIB_PATH = LIB
    import opengpc_amd as g
    from opengpc_amd.synth import synth_batch
    ctx = g.Context(0)
    ctx.load_forest(os.path.join(ROOT, "forests", forest), W, H)
    L, R = synth_batch(W, H, list(range(B)))
    s = g.Settings.sparsematch()
    cap = (W - 26) * (H - 26)
    hip = C.CDLL("libamdhip64.so")

    def dmalloc(nbytes):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0
        return p
    d_L, d_R = dmalloc(L.nbytes), dmalloc(R.nbytes)
    d_out, d_cnt, d_nc = dmalloc(B * cap * 12), dmalloc(B * 4), dmalloc(B * 8)
    assert hip.hipMemcpy(d_L, C.c_void_p(L.ctypes.data), C.c_size_t(L.nbytes), 1) == 0
    assert hip.hipMemcpy(d_R, C.c_void_p(R.ctypes.data), C.c_size_t(R.nbytes), 1) == 0
    for _ in range(4):
        ctx.match_batch_device(d_L.value, d_R.value, W, H, B, s, d_out.value, cap, d_cnt.value, d_nc.value)
        ctx.synchronize()
    NWG = 4096
    buf = np.zeros(3 * NWG, np.uint64)
    ctx.L.gpc_hip_debug_join_workgroups.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    assert ctx.L.gpc_hip_debug_join_workgroups(ctx.h, buf.ctypes.data, NWG) == 0
    rec = buf.reshape(NWG, 3)
    rec = rec[rec[:, 1] > 0]
    n = len(rec)
    t0 = int(rec[:, 0].min())
    start = (rec[:, 0].astype(np.int64) - t0) / 100.0
    end = (rec[:, 1].astype(np.int64) - t0) / 100.0
    rows = (rec[:, 2] >> np.uint64(40)).astype(np.int64)
    print("%d workgroups, %d rows; starts within %.2f us; ends %.2f .. %.2f us (p10 %.2f, p50 %.2f, p90 %.2f); rows per workgroup %d .. %d (mean %.1f)"
          % (n, int(rows.sum()), start.max(), end.min(), end.max(), np.percentile(end, 10), np.percentile(end, 50), np.percentile(end, 90),
             rows.min(), rows.max(), rows.mean()))
    for back in (40, 30, 25, 20, 15, 12, 10, 8, 6, 4, 3, 2, 1):
        t = end.max() - back
        if t > 0:
            print("  %5.1f us before the last end: %4d workgroups running" % (back, int(((start <= t) & (end > t)).sum())))
    # rows against the order the workgroups started in (the older a wave, the sooner it is served)
    order = np.argsort(start, kind="stable")
    q = max(n // 8, 1)
    print("rows taken, by start order in eighths: " + ", ".join("%.1f" % rows[order[i * q:(i + 1) * q]].mean() for i in range(8)))


if __name__ == "__main__":
    main()
