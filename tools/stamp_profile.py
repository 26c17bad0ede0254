#!/usr/bin/env python3
"""Diagnostic: builds libgpc_hip_stamps.so with -DGPC_STAMPS (s_memtime stamps at the phase
boundaries of k_row_join) and prints each phase's SHARE of the workgroup's cycles.  The stamped
build's run time is not representative (its fences forbid overlap); read shares only."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "gpurun_out", "libgpc_hip_stamps.so")
PHASES = ["loads+init", "insert left", "lookups+adds", "decide", "rank count", "rank scan+scatter", "rank walk+store"]
# k_row_join_fused (round 4): stamps at its barriers B0, B1, B2, B3, before the scan, B6, end of the row
FUSED_PHASES = ["codes arrive, keys, clears (-> B0)", "insert left + flag clear + ticket draw (-> B1)", "lookups + marks (-> B2)",
                "pending row's place, decide, rank count (-> B3)", "count out, bucket counts read, pending row's records out",
                "scan + starts + shared buckets' codes (-> B6)", "walk, ranked words, next row's loads asked for"]
HASH_PHASES = ["wait for other waves", "window arrives + staged + barrier", "next window's loads issued", "candidate flags",
               "tests + transposes", "code stores issued"]


def main():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DGPC_STAMPS",
                           "-o", LIB, os.path.join(ROOT, "opengpc_amd", "csrc", "gpc_hip.hip")])
    import opengpc_amd.capi as capi
    capi.LIB_PATH = LIB
    import opengpc_amd as g
    from opengpc_amd.synth import synth_batch
    W, H, B = 1024, 436, int(sys.argv[1]) if len(sys.argv) > 1 else 32
    ctx = g.Context(0)
    ctx.load_forest(os.path.join(ROOT, "forests", "defaultZeroForest.txt"), W, H)
    L, R = synth_batch(W, H, list(range(B)))
    s = g.Settings.sparsematch()
    cap = (W - 26) * (H - 26)
    buf = (C.c_ulonglong * 16)()
    # device-resident launches like bench.py's (the fused join + output runs there; the host entry point packs the
    # pairs' records gap-free, which takes the two-launch path)
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
    dev_step()
    ctx.L.gpc_hip_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
    ctx.L.gpc_hip_debug_stamps(ctx.h, buf)
    for _ in range(3):
        dev_step()
    ctx.L.gpc_hip_debug_stamps(ctx.h, buf)
    fused = os.environ.get("GPC_HIP_NO_FUSE") is None
    names = FUSED_PHASES if fused else PHASES
    tot = sum(buf[i] for i in range(len(names)))
    print("k_row_join (%s) phase shares (s_memtime ticks summed over the sampled workgroups' rows):" % ("fused output" if fused else "two launches"))
    nrows_sampled = 4 * B * (H - 26) / 64.0   # one workgroup in 64 reports, 4 launches since the last read
    for i, name in enumerate(names):
        print("  %-62s %6.1f %%  %7.0f cycles per row" % (name, 100.0 * buf[i] / tot, buf[i] / nrows_sampled))
    print("  total %.0f ticks = %.0f cycles per row of a sampled workgroup (%s workgroups)" % (tot, tot / nrows_sampled, os.environ.get("GPC_HIP_FUSE_WGS", "all resident")))
    HJ = ["bin bounds (scalar loads)", "records arrive", "buckets + counts", "scan", "placed", "10-cap", "ranks",
          "list order + links", "walk + scan", "output"]
    hs = g.Settings(5, 128, 1, True, True, 1)
    Bh = min(B, 32)
    ctx.match_batch(L[:Bh], R[:Bh], hs, cap)
    ctx.L.gpc_hip_debug_htjoin_stamps.argtypes = [C.c_void_p, C.c_void_p]
    ctx.L.gpc_hip_debug_htjoin_stamps(ctx.h, buf)
    ctx.match_batch(L[:Bh], R[:Bh], hs, cap)
    ctx.L.gpc_hip_debug_htjoin_stamps(ctx.h, buf)
    nwg = Bh * len([b for b in range(210) if (b & 15) == 5])
    print("k_ht_join phases (s_memtime ticks of thread 0 -- shader clocks on gfx950, ~2.1 per ns -- per sampled workgroup):")
    for i, name in enumerate(HJ):
        print("  %-28s %8.1f ticks" % (name, buf[i] / nwg))
    print("  total %.1f ticks" % (sum(buf[i] for i in range(len(HJ))) / nwg))
    ctx.L.gpc_hip_debug_hash_stamps.argtypes = [C.c_void_p, C.c_void_p]
    ctx.L.gpc_hip_debug_hash_stamps(ctx.h, buf)
    tot = sum(buf[i] for i in range(len(HASH_PHASES)))
    print("k_hash phase shares (s_memtime ticks of thread 0 of the sampled workgroups, summed over their tiles):")
    for i, name in enumerate(HASH_PHASES):
        print("  %-36s %6.1f %%" % (name, 100.0 * buf[i] / max(tot, 1)))


if __name__ == "__main__":
    main()
