#!/usr/bin/env python3
"""Experiment: absolute phase times of k_hash per workgroup (s_memtime ticks = shader clocks on gfx950, ~2.1 per ns: a 13-tile
workgroup of the 256-pair launch reports 169.5 k ticks for its ~81 us; -DGPC_STAMPS build) for any shape.  Result (round 5): the one-tile
workgroups of a single 1920x1080 Tau pair run 14.2 k ticks = 6.8 us and of a single 1024x436 pair 9.3 k = 4.4 us, where the launches
take 29.8 / 13.3 us between HIP events: a one-round launch is launch latency, the end-of-kernel write-back and event overhead, not
tile time -- nothing for the tile schedule to win there.
usage: python tools/exp/hash_phase_ticks.py W H forest pairs"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "gpurun_out", "libgpc_hip_stamps.so")
HASH_PHASES = ["prologue / wait for other waves", "window arrives + staged + barrier", "next window's loads issued", "candidate flags",
               "tests + transposes", "code stores issued"]


def main():
    W, H, forest, B = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DGPC_STAMPS",
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
    buf = (C.c_ulonglong * 16)()
    ctx.L.gpc_hip_debug_hash_stamps.argtypes = [C.c_void_p, C.c_void_p]

    def dev_step():
        ctx.match_batch_device(d_L.value, d_R.value, W, H, B, s, d_out.value, cap, d_cnt.value, d_nc.value)
        ctx.synchronize()
    for _ in range(3):
        dev_step()
    ctx.L.gpc_hip_debug_hash_stamps(ctx.h, buf)
    for _ in range(8):
        dev_step()
    ctx.L.gpc_hip_debug_hash_stamps(ctx.h, buf)
    n = max(int(buf[7]), 1)
    tot = sum(buf[i] for i in range(6))
    print("k_hash, %dx%d %s, %d pair(s): %d workgroup reports; shader-clock ticks per workgroup" % (W, H, forest, B, n))
    for i, name in enumerate(HASH_PHASES):
        print("  %-40s %8.1f ticks  %5.1f %%" % (name, buf[i] / n, 100.0 * buf[i] / max(tot, 1)))
    print("  %-40s %8.1f ticks ~ %.2f us at 2.1 GHz" % ("total", tot / n, tot / n / 2100.0))


if __name__ == "__main__":
    main()
