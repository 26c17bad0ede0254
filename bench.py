#!/usr/bin/env python3
"""bench.py -- throughput of the openGPC hot path (preprocess + hash + match) on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the whole timed region of the reference's sparsematch
(samples/sparsematch.cpp:45-52: preprocessImage x2 + rectifiedMatch) over one batch of
synthetic 1024x436 pairs per GPU, inputs already resident in HBM, supports left in HBM.
Pairs shard embarrassingly (pair i -> rank i mod N); the only collectives are
the timing barrier / MAX and a gather of per-rank counters.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
DOMINANT_KERNEL = "k_row_join"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256,
                    help="pairs per GPU per step: BASELINE configs[3]'s batch of 256 pairs, one such batch per GPU "
                         "(measured: 32 -> 125, 256 -> 150, 384 -> 152, 512 -> 155 Gpix/s; larger batches only amortise "
                         "the four launch gaps further)")
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=436)
    ap.add_argument("--forest", default=os.path.join(ROOT, "forests", "defaultZeroForest.txt"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the single-pair and two-stream side measurements (profiling runs: every launch "
                         "of a kernel then has the same grid, so rocprofv3's averages match the HIP-event ones)")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="contexts (HIP streams + workspaces) the steps alternate over; 1 = strictly serial steps "
                         "(default: clean per-kernel timing); 2 lets step k+1's HBM-bound kernels overlap step k's "
                         "LDS/VALU-bound ones (+9 % throughput, reported as `two_stream_pipeline` at N=1 anyway)")
    return ap.parse_args()


def host_cores():
    """CPU share of this process: cgroup quota if set, else the affinity mask, never more than 16
    (a one-GPU box's share of its host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(args, W, H):
    """Single-thread CPU time of the same timed region on a bounded sample of the workload.
    Uses the reference's own SSE kernels (oracle/_ref) + a C++ port of the inference.hpp glue
    when that build is present, otherwise the plain C oracle (a slower, scalar port)."""
    from oracle.pyoracle import Oracle, Ref, sparsematch_settings
    from opengpc_amd.synth import synth_pair
    o = Oracle(fast=True)
    rc, f = o.read_forest(args.forest, W, H)
    s = sparsematch_settings()
    use_ref = Ref.available()
    ref = Ref() if use_ref else None
    times, idx = [], 0
    t_end = time.time() + args.cpu_seconds
    while time.time() < t_end or len(times) < 3:
        L, R = synth_pair(W, H, idx, 8 + idx % 64)
        t0 = time.perf_counter()
        if use_ref:
            ref.cpu_baseline_pair(L, R, f, s)
        else:
            o.match_pair(L, R, f, s)
        times.append(time.perf_counter() - t0)
        idx += 1
        if len(times) >= 400:
            break
    times.sort()
    med = times[len(times) // 2]
    # all host cores, pairs in parallel: one thread per core, each matching its own
    # pairs; ctypes releases the GIL for the duration of a call
    allc = None
    if use_ref:
        import threading
        ncores = host_cores()
        per = max(2, min(8, int(args.cpu_seconds / 3.0 / med / 1.5)))
        pairs = [synth_pair(W, H, 1000 + i, 8 + i % 64) for i in range(min(ncores, 16))]

        def work(t):
            L, R = pairs[t % len(pairs)]
            for _ in range(per):
                ref.cpu_baseline_pair(L, R, f, s)
        th = [threading.Thread(target=work, args=(t,)) for t in range(ncores)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        allc = {"value": round(2.0 * W * H * per * ncores / dt / 1e6, 3), "unit": "Mpix/s", "cores": ncores,
                "sample": "%d threads x %d pairs, pairs in parallel" % (ncores, per)}
    what = ("reference SSE kernels (filter.hpp via oracle/_ref) + C++ port of the inference.hpp glue "
            "(std::sort on 24-byte descriptors)") if use_ref else "scalar C oracle (oracle/gpc_oracle.c, -O3 -march=native)"
    return {
        "value": round(2.0 * W * H / med / 1e6, 3),
        "unit": "Mpix/s",
        "cores": 1,
        "kind": "port",
        "ms_per_pair": round(med * 1e3, 3),
        "sample": "%d pairs %dx%d (s=i, D=8+i%%64), median; %s; cold first pair excluded by median" % (len(times), W, H, what),
        "all_cores": allc,
    }


def main():
    args = parse_args()
    import numpy as np
    import torch

    from opengpc_amd import dist as gdist

    rank, world, local_rank = gdist.env_world()
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    gdist.init("nccl", dev)  # nccl == RCCL on ROCm; no-op for a single process

    import opengpc_amd as g
    from opengpc_amd.synth import synth_batch

    W, H, B = args.width, args.height, args.batch
    P = max(1, args.pipeline)
    settings = g.Settings.sparsematch()
    ctxs, streams = [], []
    for _ in range(P):  # one context = one HIP stream + one set of workspaces
        c = g.Context(local_rank)
        fm = c.load_forest(args.forest, W, H)
        st = torch.cuda.Stream(device=dev)
        c.set_stream(st.cuda_stream)
        c.reserve(W, H, B)
        ctxs.append(c)
        streams.append(st)
    ctx = ctxs[0]

    # pair i -> rank i mod N  (weak scaling: B pairs per GPU per step)
    indices = gdist.shard_indices(rank, world, B)
    Lh, Rh = synth_batch(W, H, indices)
    d_L = torch.from_numpy(Lh).to(dev)
    d_R = torch.from_numpy(Rh).to(dev)
    cap = (W - 26) * (H - 26)  # a row can emit at most W-26 supports
    # every in-flight step owns its outputs (gpc_support = 12 bytes)
    d_outs = [torch.empty((B, cap, 3), dtype=torch.int32, device=dev) for _ in range(P)]
    d_cnts = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(P)]
    d_ncs = [torch.zeros((B, 2), dtype=torch.int32, device=dev) for _ in range(P)]
    d_out, d_counts, d_ncand = d_outs[0], d_cnts[0], d_ncs[0]
    torch.cuda.synchronize(dev)

    step_no = [0]

    def step():
        i = step_no[0] % P
        step_no[0] += 1
        ctxs[i].match_batch_device(d_L.data_ptr(), d_R.data_ptr(), W, H, B, settings, d_outs[i].data_ptr(), cap,
                                   d_cnts[i].data_ptr(), d_ncs[i].data_ptr())

    def device_sync():
        for c in ctxs:
            c.synchronize()  # the streams the kernels run on
        torch.cuda.synchronize(dev)

    for _ in range(max(args.warmup, P)):
        step()
    device_sync()

    # HIP events around every launch of the dominant kernel, on the stream it runs on; the other
    # kernels are bracketed in a separate pass below so their event records do not sit in the
    # timed region
    for c in ctxs:
        c.enable_kernel_timing(True, only=[DOMINANT_KERNEL])
        c.reset_kernel_timing()
    step_no[0] = 0
    elapsed = gdist.timed_steps(step, args.steps, device_sync)
    dom_ms, dom_n = 0.0, 0
    for c in ctxs:
        ms, n = c.kernel_times()[DOMINANT_KERNEL]
        dom_ms += ms
        dom_n += n
        c.enable_kernel_timing(False)
    # per-kernel split: a few extra, serial steps with every kernel bracketed (not part of `value`)
    ctx.enable_kernel_timing(True)
    ctx.reset_kernel_timing()
    for _ in range(5):
        ctx.match_batch_device(d_L.data_ptr(), d_R.data_ptr(), W, H, B, settings, d_out.data_ptr(), cap,
                               d_counts.data_ptr(), d_ncand.data_ptr())
    ktimes = ctx.kernel_times()
    ctx.enable_kernel_timing(False)

    counts = d_counts.cpu().numpy().astype(np.int64)
    ncand = d_ncand.cpu().numpy().astype(np.int64)
    # O(32 B) per rank over xGMI: timing / counters only, never pixel data
    allr = gdist.gather_stats([elapsed, float(B), float(ncand.sum()), float(counts.sum())], device=dev).numpy()
    job = gdist.reduce_job(torch.from_numpy(allr), args.steps, 2 * W * H)
    t_max = job["t_max"]
    pairs_per_step = job["pairs_per_step"]

    if rank == 0:
        mpix_per_step = 2.0 * W * H * pairs_per_step / 1e6
        value = mpix_per_step * args.steps / t_max

        # ---- parity gate on the bench's own data: pair 0 against the oracle (checker only)
        verified = None
        if not args.no_verify:
            from oracle.pyoracle import Oracle, sparsematch_settings
            o = Oracle()
            rc, f = o.read_forest(args.forest, W, H)
            want, nl, nr = o.match_pair(Lh[0], Rh[0], f, sparsematch_settings())
            got = d_out[0, : int(counts[0])].cpu().numpy()
            verified = bool(len(want) == int(counts[0]) and (nl, nr) == tuple(int(v) for v in ncand[0])
                            and np.array_equal(got[:, 0], want["x"]) and np.array_equal(got[:, 1], want["y"])
                            and np.array_equal(got[:, 2].view(np.float32), want["d"]))
            if not verified:
                raise SystemExit("bench.py: GPU supports differ from the oracle -- refusing to report a number")

        # ---- roofline of the dominant kernel: algorithmic bytes / HIP-event time
        # per pair:  A = 10*W*H + 48*N + 12*M; the row-match launch owns the sort/match share
        # 36*N + 12*M (sort read+write 24, match read 12, supports 12) of every pair it processes.
        N_step = float(ncand.sum())
        M_step = float(counts.sum())
        alg = {
            "k_preprocess": 6.0 * W * H * B,               # raw read + smooth/grad write, both images
            "k_hash": 4.0 * W * H * B + 12.0 * N_step,      # smooth+grad read, key+index write
            "k_row_join": 36.0 * N_step,                   # sort read+write once, match read
            "k_gather_rows": 12.0 * M_step,                 # supports out
        }
        kinfo = {}
        for name, (ms, n) in ktimes.items():
            if n:
                kinfo[name] = {"avg_us": round(1e3 * ms / n, 2), "launches": n}
                if name in alg:
                    kinfo[name]["alg_GBs"] = round(alg[name] / (ms / n * 1e-3) / 1e9, 1)
        dom_name = DOMINANT_KERNEL
        serial_dom = max(ktimes.items(), key=lambda kv: kv[1][0])[0]
        achieved = alg.get(dom_name, 0.0) / (dom_ms / max(dom_n, 1) * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as fh:
                    traffic = json.load(fh).get("%s@%dx%dx%d" % (dom_name, W, H, B))
            except Exception:
                traffic = None
        a_pair = (10.0 * W * H * B + 48.0 * N_step + 12.0 * M_step) / B
        roofline = {
            "bound": "hbm",
            "kernel": dom_name,
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "alg_bytes_per_launch": alg.get(dom_name),
            "avg_launch_us": round(1e3 * dom_ms / max(dom_n, 1), 2),
            "pipeline_alg_bytes_per_pair": a_pair,
            "pipeline_frac": round(a_pair * pairs_per_step * args.steps / t_max / 1e9 / HBM_PEAK_GBS, 4),
            "launches_timed": dom_n,
            "note": "achieved = SURVEY 8d's ALGORITHMIC bytes of a sort-based matcher (36 B per candidate) / launch "
                    "duration: the join keeps that sort on-chip (LDS), so it can exceed what HBM could stream and its "
                    "measured HBM traffic (`traffic`) is ~5x smaller. "
                    "Duration = HIP events around every %s launch inside the timed region (steps of %d "
                    "alternating streams may overlap it with the next step's HBM-bound kernels); `kernels` = "
                    "per-kernel split of 5 extra serial steps; largest there: %s" % (dom_name, P, serial_dom),
            "kernels": kinfo,
        }

        # achievable-copy figure of this box: device-to-device copy of 1 GiB, read + write
        if world == 1 and not args.no_extras:
            nb = 1 << 30
            a = torch.empty(nb, dtype=torch.uint8, device=dev)
            b = torch.empty(nb, dtype=torch.uint8, device=dev)
            b.copy_(a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev)
            e0.record()
            for _ in range(10):
                b.copy_(a)
            e1.record()
            torch.cuda.synchronize(dev)
            copy_gbs = 2.0 * nb * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del a, b
            roofline["copy_measured_GBs"] = round(copy_gbs, 1)
            roofline["frac_of_copy"] = round(achieved / copy_gbs, 4)

        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args, W, H)

        # the same steps alternating over TWO streams/workspaces (step k+1 overlaps step k); reported
        # beside the serial headline because its per-kernel event times are no longer clean
        two = None
        if world == 1 and P == 1 and not args.no_extras:
            c2 = g.Context(local_rank)
            c2.load_forest(args.forest, W, H)
            st2 = torch.cuda.Stream(device=dev)
            c2.set_stream(st2.cuda_stream)
            c2.reserve(W, H, B)
            o2 = torch.empty((B, cap, 3), dtype=torch.int32, device=dev)
            n2 = torch.zeros(B, dtype=torch.int32, device=dev)
            m2 = torch.zeros((B, 2), dtype=torch.int32, device=dev)
            pair = [(ctx, d_out, d_counts, d_ncand), (c2, o2, n2, m2)]

            def step2(i):
                c, o, n, m = pair[i & 1]
                c.match_batch_device(d_L.data_ptr(), d_R.data_ptr(), W, H, B, settings, o.data_ptr(), cap,
                                     n.data_ptr(), m.data_ptr())
            for i in range(4):
                step2(i)
            device_sync(); c2.synchronize()
            t2 = time.perf_counter()
            for i in range(args.steps):
                step2(i)
            device_sync(); c2.synchronize()
            dt2 = (time.perf_counter() - t2) / args.steps
            same = bool(torch.equal(n2, d_counts) and torch.equal(o2[0, : int(counts[0])], d_out[0, : int(counts[0])]))
            two = {"streams": 2, "ms_per_step": round(dt2 * 1e3, 4), "value": round(mpix_per_step / dt2, 1),
                   "unit": "Mpix/s", "identical_outputs": same}
            c2.close()

        # BASELINE configs[1] taken literally: ONE pair per step (launch/occupancy-bound, reported
        # beside the batched headline, never instead of it)
        single = None
        if world == 1 and not args.no_extras:
            def step1():
                ctx.match_batch_device(d_L.data_ptr(), d_R.data_ptr(), W, H, 1, settings, d_outs[-1].data_ptr(), cap,
                                       d_cnts[-1].data_ptr(), d_ncs[-1].data_ptr())
            for _ in range(5):
                step1()
            device_sync()
            n1 = 200
            t1 = time.perf_counter()
            for _ in range(n1):
                step1()
            device_sync()
            dt1 = (time.perf_counter() - t1) / n1
            single = {"ms_per_pair": round(dt1 * 1e3, 4), "Mpix_per_s": round(2.0 * W * H / dt1 / 1e6, 1),
                      "note": "one 1024x436 pair per step, back-to-back steps, inputs/outputs in HBM"}

        line = {
            "metric": "Mpix/s hashed+matched (1024x436 pair)",
            "value": round(value, 1),
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * t_max / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1] pairs (%dx%d, %s, sparsematch settings: thr 5, epipolar, sort-match) "
                            "in batches of %d pairs per GPU per step (configs[3] sharding: pair i -> GPU i mod N)"
                            % (W, H, os.path.basename(args.forest), B),
                "width": W, "height": H, "pairs_per_gpu_per_step": B, "tests": fm.num_tests,
                "forest_type": fm.type, "parallelism": "pairs-dp%d" % world, "streams_per_gpu": P,
                "timed_region": "raw pairs in HBM -> supports in HBM (no PCIe)",
            },
            "pairs_per_s": round(pairs_per_step * args.steps / t_max, 1),
            "candidates_per_pair": round(N_step / B, 1),
            "supports_per_pair": round(M_step / B, 1),
            "verified_vs_oracle": verified,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "single_pair": single,
            "two_stream_pipeline": two,
        }
        if cpu:
            line["speedup_vs_cpu_1thread"] = round(value / cpu["value"], 1)
        print(json.dumps(line), flush=True)

    for c in ctxs:
        c.close()
    gdist.finalize()


if __name__ == "__main__":
    main()
