#!/usr/bin/env python3
"""Experiment: k_hash's phase times per workgroup (s_memtime ticks = shader clocks on gfx950, ~2.0 per ns), a workgroup's life and
the launch's extent on the 100 MHz clock (s_memrealtime), for any shape (-DGPC_STAMPS build).  Round 5, what it showed:
  * one 1920x1080 Tau pair (432 one-tile workgroups on 512 places): every workgroup starts within 0.7 us, a workgroup lives
    7.4 us on average (10 us where two share a CU), yet the launch takes 29 us between HIP events (21 us with neither tests nor
    stores, -DHT_EXP_NOCOMPUTE -DHT_EXP_NOSTORE): a one-round launch is launch latency, event overhead, the end-of-kernel
    write-back and the statistics' atomics -- 216 workgroups per image add to the SAME three words, ~8 ns each in turn (this
    script's own three atomicMax per workgroup stretch the extent by ~10 us the same way);
  * one 1024x436 pair (104 workgroups): lives 4.3 us, extent 4.6 us, 13 us between events;
  * 32 pairs (512 workgroups of 6 / 7 tiles): mean life 40 us, the last ends at 52 us = the launch.
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
    # launches on their own: the spread of the workgroups' starts and the launch's extent on the 100 MHz clock
    extent = []
    for _ in range(5):
        dev_step()
        ctx.L.gpc_hip_debug_hash_stamps(ctx.h, buf)
        first = (1 << 62) - int(buf[8])
        extent.append(((int(buf[9]) - first) / 100.0, (int(buf[10]) - first) / 100.0))
    for _ in range(8):
        dev_step()
    ctx.L.gpc_hip_debug_hash_stamps(ctx.h, buf)
    print("per launch: last workgroup start / last workgroup end after the first workgroup's start, us:", extent)
    n = max(int(buf[7]), 1)
    tot = sum(buf[i] for i in range(6))
    print("k_hash, %dx%d %s, %d pair(s): %d workgroup reports; shader-clock ticks per workgroup" % (W, H, forest, B, n))
    for i, name in enumerate(HASH_PHASES):
        print("  %-40s %8.1f ticks  %5.1f %%" % (name, buf[i] / n, 100.0 * buf[i] / max(tot, 1)))
    print("  %-40s %8.1f ticks; the workgroup's life by s_memrealtime (100 MHz): %.2f us => %.2f s_memtime ticks per ns"
          % ("total", tot / n, buf[6] / n / 100.0, (tot / n) / max(buf[6] / n * 10.0, 1e-9)))


if __name__ == "__main__":
    main()
