#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer batch entry point (gpc_hip_match_batch): raw pairs in
host memory -> supports in host memory, synchronous.  Never used as bench.py's `value`."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import opengpc_amd as g  # noqa: E402
from opengpc_amd.synth import synth_batch  # noqa: E402
from opengpc_amd.hostinfo import cpu_nodes, current_cpu, pages_nodes, stage_summary  # noqa: E402


def one(B, W, H, forest):
    """One record for bench.py: B pairs, page-locked buffers, median of 15 synchronous calls after 5 untimed ones (the
    first calls of a process touch the page-locked buffers and find clocks and worker threads cold); counts and checksums
    of three pairs so that the caller can hold the result against its device-resident one."""
    import zlib
    ctx = g.Context(0)
    ctx.load_forest(forest, W, H)
    s = g.Settings.sparsematch()
    L, R = synth_batch(W, H, list(range(B)))
    cap = 300000
    Lp, Rp = ctx.pinned_empty(L.shape, np.uint8), ctx.pinned_empty(R.shape, np.uint8)
    Lp[:] = L
    Rp[:] = R
    out = ctx.pinned_empty((B, cap), g.SUPPORT_DTYPE)
    for _ in range(5):
        o, counts, ncand, st = ctx.match_batch(Lp, Rp, s, cap, out=out)
    tt = []
    for _ in range(15):
        t0 = time.perf_counter()
        o, counts, ncand, st = ctx.match_batch(Lp, Rp, s, cap, out=out)
        tt.append(time.perf_counter() - t0)
    tt.sort()
    dt = tt[len(tt) // 2]
    rec = {"value": round(2.0 * W * H * B / dt / 1e6, 1), "unit": "Mpix/s", "ms_per_call": round(dt * 1e3, 3),
           "ms_per_call_min": round(tt[0] * 1e3, 3), "ms_per_call_max": round(tt[-1] * 1e3, 3),
           "pairs_per_call": B, "host_buffers": "page-locked (gpc_hip_host_alloc)", "status": int(st),
           "bytes_in": int(L.nbytes + R.nbytes), "bytes_over_the_link_out": int(counts.sum()) * 4 + B * H * 4,
           "bytes_delivered": int(counts.sum()) * 12, "counts": [int(v) for v in counts],
           "crc32": {str(j): zlib.crc32(o[j, : int(counts[j])].tobytes()) for j in (0, B // 2, B - 1)}}
    rec["single_pair_host_to_host"] = single_pair(ctx, W, H, s)
    rec["host"] = host_info(ctx)
    print(json.dumps(rec))
    ctx.close()


def serve(B, W, H, forest, dev_index, warm=5):
    """One rank's torch-free host-to-host leg, driven by bench.py over a pipe (bench.py: HostChild): this process holds the
    library's own ROCm runtime -- the situation of a C++ caller -- while the bench process (torch's bundled runtime) carries
    the barriers between the ranks.  Protocol, one line each way: after `warm` untimed calls "ready"; per "go" one synchronous
    gpc_hip_match_batch call, answered with its seconds; "done" -> one JSON record (counts, checksums of three pairs, host
    description; rank 0 also the single-pair leg) and exit."""
    import zlib
    ctx = g.Context(dev_index)
    ctx.load_forest(forest, W, H)
    s = g.Settings.sparsematch()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    L, R = synth_batch(W, H, [rank + world * j for j in range(B)])   # this rank's shard: pair i -> rank i mod N
    cap = 300000
    Lp, Rp = ctx.pinned_empty(L.shape, np.uint8), ctx.pinned_empty(R.shape, np.uint8)
    Lp[:] = L
    Rp[:] = R
    out = ctx.pinned_empty((B, cap), g.SUPPORT_DTYPE)
    st, counts, o = 0, None, None
    for _ in range(warm):
        o, counts, ncand, st = ctx.match_batch(Lp, Rp, s, cap, out=out)
    # the same call with the records LEFT packed (gpc_hip_match_batch_packed): 4 bytes per support + row counts
    pk, prow = ctx.pinned_empty((B, cap), np.uint32), ctx.pinned_empty((B, H), np.int32)
    pst, pcounts = 0, None
    for _ in range(2):
        pk, prow, pcounts, pncand, pst = ctx.match_batch_packed(Lp, Rp, s, cap, packed=pk, rows=prow)
    sys.stdout.write("ready\n")
    sys.stdout.flush()
    tt, tp = [], []
    st_rows, sp_rows = [], []
    for line in sys.stdin:
        cmd = line.strip()
        if cmd == "go":
            t0 = time.perf_counter()
            o, counts, ncand, st = ctx.match_batch(Lp, Rp, s, cap, out=out)
            dt = time.perf_counter() - t0
            tt.append(dt)
            st_rows.append(ctx.batch_stages())
            sys.stdout.write("%.9f\n" % dt)
            sys.stdout.flush()
        elif cmd == "gop":
            t0 = time.perf_counter()
            pk, prow, pcounts, pncand, pst = ctx.match_batch_packed(Lp, Rp, s, cap, packed=pk, rows=prow)
            dt = time.perf_counter() - t0
            tp.append(dt)
            sp_rows.append(ctx.batch_stages())
            sys.stdout.write("%.9f\n" % dt)
            sys.stdout.flush()
        elif cmd == "done":
            break
    rec = {"status": int(st), "calls": len(tt), "pairs_per_call": B, "host_buffers": "page-locked (gpc_hip_host_alloc)",
           "bytes_in": int(L.nbytes + R.nbytes), "bytes_over_the_link_out": int(counts.sum()) * 4 + B * H * 4,
           "bytes_delivered": int(counts.sum()) * 12, "counts": [int(v) for v in counts],
           "crc32": {str(j): zlib.crc32(o[j, : int(counts[j])].tobytes()) for j in sorted(set((0, B // 2, B - 1)))},
           "host": host_info(ctx)}
    # the packed call against the expanded one: same counts, and three pairs expanded on the host record for record
    ok = bool(pst == 0 and np.array_equal(pcounts, counts))
    for j in sorted(set((0, B // 2, B - 1))):
        ok = ok and np.array_equal(g.capi.expand_packed(pk[j], prow[j], int(counts[j])), o[j, : int(counts[j])])
    rec["packed"] = {"status": int(pst), "calls": len(tp), "identical_to_expanded": ok,
                     "bytes_delivered": int(counts.sum()) * 4 + B * H * 4, "stages_ms": stage_summary(sp_rows)}
    rec["stages_ms"] = stage_summary(st_rows)
    rec["pages_on_node"] = {"out": pages_nodes(out), "images": pages_nodes(Lp), "packed_out": pages_nodes(pk)}
    if rank == 0 and not os.environ.get("GPC_BENCH_NO_SINGLE"):
        rec["single_pair_host_to_host"] = single_pair(ctx, W, H, s)
    sys.stdout.write(json.dumps(rec) + "\n")
    sys.stdout.flush()
    ctx.close()


def host_info(ctx):
    """What the host side of the call had to work with (the expansion of packed records is CPU work)."""
    model = ""
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    info = {"cpu_model": model, "cpus_visible": len(os.sched_getaffinity(0)), "local_world_size": int(os.environ.get("LOCAL_WORLD_SIZE", "1"))}
    try:
        info["expand_threads"] = int(ctx.L.gpc_hip_host_threads(ctx.h))
        info["gpu_numa_node"] = int(ctx.L.gpc_hip_host_numa_node(ctx.h))
        info["workers_bound_to_gpu_node"] = info["gpu_numa_node"] >= 0
        nodes = cpu_nodes()
        cpus = ctx.worker_cpus()
        info["worker_cpus"] = cpus
        info["worker_nodes"] = sorted(set(nodes.get(c, -1) for c in cpus))
        me = current_cpu()
        info["calling_thread_cpu"] = me
        info["calling_thread_node"] = nodes.get(me, -1)
        info["numa_nodes"] = len(set(nodes.values()))
    except Exception:
        pass
    return info


def single_pair(ctx, W, H, s):
    """BASELINE configs[1] taken literally: ONE pair, host images -> host supports (the reference's t0..t2), page-locked
    and pageable buffers; first call and median of 50."""
    L, R = synth_batch(W, H, [0])
    cap = 300000
    res = {}
    for kind in ("pinned", "pageable"):
        if kind == "pinned":
            Lb, Rb = ctx.pinned_empty(L.shape, np.uint8), ctx.pinned_empty(R.shape, np.uint8)
            Lb[:] = L
            Rb[:] = R
            out = ctx.pinned_empty((1, cap), g.SUPPORT_DTYPE)
        else:
            Lb, Rb, out = L.copy(), R.copy(), np.empty((1, cap), g.SUPPORT_DTYPE)
        tt = []
        for _ in range(60):
            t0 = time.perf_counter()
            o, counts, ncand, st = ctx.match_batch(Lb, Rb, s, cap, out=out)
            tt.append(time.perf_counter() - t0)
        first, rest = tt[0], sorted(tt[10:])
        res[kind] = {"ms_first_call": round(first * 1e3, 4), "ms_median": round(rest[len(rest) // 2] * 1e3, 4),
                     "ms_min": round(rest[0] * 1e3, 4), "Mpix_per_s": round(2.0 * W * H / rest[len(rest) // 2] / 1e6, 1),
                     "supports": int(counts[0]), "status": int(st),
                     "crc32": __import__("zlib").crc32(o[0, : int(counts[0])].tobytes())}
    res["timed_region"] = "one synchronous gpc_hip_match_batch call with one pair: host images -> host gpc_support array (sparsematch.cpp:45-52)"
    return res


def main():
    if len(sys.argv) >= 2 and sys.argv[1] == "--one":
        return one(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
    if len(sys.argv) >= 2 and sys.argv[1] == "--serve":
        return serve(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], int(sys.argv[6]))
    W, H = 1024, 436
    ctx = g.Context(0)
    ctx.load_forest(os.path.join(ROOT, "forests", "defaultZeroForest.txt"), W, H)
    s = g.Settings.sparsematch()
    res = []
    for B in (1, 32, 256):
        for pinned in (False, True):
            if B == 256 and not pinned:
                continue
            L, R = synth_batch(W, H, list(range(B)))
            cap = 300000
            # (the output array is allocated and touched ONCE either way: a fresh np.empty per call is 115 MB of first-touch
            # page faults at 32 pairs -- 6 of the 7.5 ms "pageable" calls of earlier records were that, not the copies)
            out = np.zeros((B, cap), g.SUPPORT_DTYPE)
            if pinned:
                Lp, Rp = ctx.pinned_empty(L.shape, np.uint8), ctx.pinned_empty(R.shape, np.uint8)
                Lp[:] = L
                Rp[:] = R
                L, R = Lp, Rp
                out = ctx.pinned_empty((B, cap), g.SUPPORT_DTYPE)
            for _ in range(4):  # the first calls touch the page-locked buffers and find clocks and worker threads cold
                o, counts, ncand, st = ctx.match_batch(L, R, s, cap, out=out)
            n = 10
            t0 = time.perf_counter()
            for _ in range(n):
                o, counts, ncand, st = ctx.match_batch(L, R, s, cap, out=out)
            dt = (time.perf_counter() - t0) / n
            rec = {"pairs": B, "host_buffers": "pinned (gpc_hip_host_alloc)" if pinned else "pageable",
                   "ms_per_call": round(dt * 1e3, 3), "Mpix_per_s": round(2.0 * W * H * B / dt / 1e6, 1),
                   "bytes_in": int(L.nbytes + R.nbytes), "bytes_out": int(counts.sum()) * 12,
                   "note": "one synchronous call = H2D + pipeline + D2H of the valid supports"}
            res.append(rec)
            print(json.dumps(rec))
    # double-buffered feed: T host threads, one context (stream + workspaces + pinned
    # staging) each, every thread issuing synchronous calls -- PCIe is full duplex, so one thread's
    # D2H overlaps another's H2D and kernels
    import threading
    B = 32
    for T in (2, 3):
        ctxs, bufs = [], []
        for t in range(T):
            c = g.Context(0)
            c.load_forest(os.path.join(ROOT, "forests", "defaultZeroForest.txt"), W, H)
            L, R = synth_batch(W, H, list(range(B)))
            Lp, Rp = c.pinned_empty(L.shape, np.uint8), c.pinned_empty(R.shape, np.uint8)
            Lp[:] = L
            Rp[:] = R
            out = c.pinned_empty((B, 300000), g.SUPPORT_DTYPE)
            c.match_batch(Lp, Rp, s, 300000, out=out)
            ctxs.append(c)
            bufs.append((Lp, Rp, out))
        n = 10
        tot = [0] * T

        def work(t):
            Lp, Rp, out = bufs[t]
            for _ in range(n):
                o, counts, ncand, st = ctxs[t].match_batch(Lp, Rp, s, 300000, out=out)
            tot[t] = int(counts.sum())
        th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        dt = (time.perf_counter() - t0) / (n * T)
        rec = {"pairs": B, "host_buffers": "pinned, %d host threads x 1 context each" % T,
               "ms_per_call": round(dt * 1e3, 3), "Mpix_per_s": round(2.0 * W * H * B / dt / 1e6, 1),
               "bytes_in": int(bufs[0][0].nbytes * 2), "bytes_out": tot[0] * 12,
               "note": "amortised time per 32-pair call with calls of different threads overlapping"}
        res.append(rec)
        print(json.dumps(rec))
        for c in ctxs:
            c.close()
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "pcie_inclusive.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
