#!/usr/bin/env python3
"""Experiment: every workgroup of ONE k_hash launch -- start, end (s_memrealtime, 100 MHz), the XCD / SE / CU it ran on
(-DGPC_STAMPS build): who are the workgroups a one-round launch waits for?  Round 5:
  * 32 pairs of 1024x436 (512 workgroups of 6 / 7 tiles, two on every CU, all started within 0.5 us): ends 28.9 .. 51.2 us,
    lives p50 38.6 / p90 49.2 / max 51.0 us.  Read as: the older workgroup of a CU gets the issue slots first and ends at
    29-39 us, its younger mate at ~50: a CU works its 13 tiles in 50 us = 3.85 us per tile where the 256-pair launch (13-tile workgroups
    replaced as they end) runs 3.13 -- the start of a workgroup and its last tiles alone on the CU (8 waves) are what a
    one-round launch pays; every XCD within 2 %.
  * one 1920x1080 Tau pair (432 one-tile workgroups): alone on a CU 7.1 us, two on a CU 9.8 us on average -- and two XCDs of
    the eight (1 and 6 in the launch recorded) with lives of 18-21 us: the launch's extent (21 us) is theirs.
usage: python tools/exp/hash_wg_lives.py W H forest pairs"""
import ctypes as C
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "gpurun_out", "libgpc_hip_stamps.so")
import numpy as np  # noqa: E402


def main():
    W, H, forest, B = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DGPC_STAMPS",
                           "-o", LIB, os.path.join(ROOT, "opengpc_amd", "csrc", "gpc_hip.hip")])
    import opengpc_amd.capi as capi
    capi.LIB_PATH = LIB
    import opengpc_amd as g
    from opengpc_amd.synth import synth_batch
    os.environ["GPC_HIP_DEBUG_PLAN"] = "1"
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

    def dev_step():
        ctx.match_batch_device(d_L.value, d_R.value, W, H, B, s, d_out.value, cap, d_cnt.value, d_nc.value)
        ctx.synchronize()
    for _ in range(4):
        dev_step()
    NWG = 8192
    buf = np.zeros(3 * NWG, np.uint64)
    ctx.L.gpc_hip_debug_hash_workgroups.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    assert ctx.L.gpc_hip_debug_hash_workgroups(ctx.h, buf.ctypes.data, NWG) == 0
    rec = buf.reshape(NWG, 3)
    used = rec[:, 1] > 0
    rec = rec[used]
    n = len(rec)
    t0 = int(rec[:, 0].min())
    start = (rec[:, 0].astype(np.int64) - t0) / 100.0
    end = (rec[:, 1].astype(np.int64) - t0) / 100.0
    life = end - start
    hw = rec[:, 2] & np.uint64(0xFFFFFFFF)
    xcc = (rec[:, 2] >> np.uint64(32)) & np.uint64(0xF)
    cu = (hw >> np.uint64(8)) & np.uint64(0xF)
    sh = (hw >> np.uint64(12)) & np.uint64(0x1)
    se = (hw >> np.uint64(13)) & np.uint64(0x7)
    print("%d workgroups recorded (the last launch that wrote each slot); starts within %.2f us, ends %.2f .. %.2f us, life mean %.2f  p50 %.2f  p90 %.2f  max %.2f us"
          % (n, start.max(), end.min(), end.max(), life.mean(), np.percentile(life, 50), np.percentile(life, 90), life.max()))
    # how many workgroups share a CU, and the lives by that
    place = defaultdict(list)
    for i in range(n):
        place[(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]))].append(i)
    by_share = defaultdict(list)
    for k, v in place.items():
        for i in v:
            by_share[len(v)].append(life[i])
    print("distinct (XCD, SE, SH, CU) places: %d" % len(place))
    for k in sorted(by_share):
        a = np.array(by_share[k])
        print("  workgroups on a CU that ran %d of them: %4d, life mean %.2f  max %.2f us" % (k, len(a), a.mean(), a.max()))
    for x in range(8):
        m = xcc == x
        if m.any():
            print("  XCD %d: %3d workgroups, life mean %.2f  max %.2f, last end %.2f us" % (x, int(m.sum()), life[m].mean(), life[m].max(), end[m].max()))
    # when the places run dry: how many workgroups are still running t us before the launch's last end
    for back in (40, 30, 20, 15, 10, 7, 5, 3, 2, 1):
        t = end.max() - back
        if t > 0:
            print("  %5.1f us before the last end: %4d workgroups running, %4d not yet started" % (back, int(((start <= t) & (end > t)).sum()), int((start > t).sum())))
    order = np.argsort(-life)[:12]
    print("longest: " + ", ".join("#%d %.1f us (XCD %d SE %d CU %d)" % (int(np.flatnonzero(used)[i]), life[i], int(xcc[i]), int(se[i]), int(cu[i])) for i in order))


if __name__ == "__main__":
    main()
