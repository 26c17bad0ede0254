#!/usr/bin/env python3
"""bench.py -- throughput of the openGPC hot path (preprocess + hash + match) on MI355X.

    python bench.py --gpus N --steps K --warmup W

is the ONE command at any N: with N > 1 and no WORLD_SIZE in the environment the script starts its own N rank processes
(one per GPU, RCCL between them) before anything touches a GPU, relays rank 0's JSON line and exits with the ranks' status;
it fails loudly when fewer than N devices are visible (GPC_DIST_BACKEND=gloo: rehearsal with the ranks sharing the devices
there are).  Under a launcher that exports RANK / WORLD_SIZE itself --
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
-- every process is one rank, as before.

One step = one pass of the whole timed region of the reference's sparsematch
(samples/sparsematch.cpp:45-52: preprocessImage x2 + rectifiedMatch) over one batch of
synthetic 1024x436 pairs per GPU, inputs already resident in HBM, supports left in HBM.
Pairs shard embarrassingly (pair i -> rank i mod N); the only collectives are
the timing barrier / MAX and a gather of per-rank counters.  Prints ONE JSON line on rank 0.

A timed WINDOW is exactly K steps bracketed by barrier + device sync on both sides (the driver's
contract).  K = 20 steps last ~23 ms, too short to be seen from outside, so the window is repeated
(--windows, default 260 => the timed windows keep the GPU busy for ~6 s) and the MEDIAN window is
reported; min / max are in the line.  The reference's own timed region is host-to-host: `pcie_inclusive` measures that
in the same run and `speedup_vs_cpu_1thread` compares like with like.
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


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--windows", type=int, default=260,
                    help="timed windows of --steps steps each (every one bracketed by barrier + sync); the median is reported")
    ap.add_argument("--batch", type=int, default=256,
                    help="pairs per GPU per step: BASELINE configs[3]'s batch of 256 pairs, one such batch per GPU")
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=436)
    ap.add_argument("--forest", default=os.path.join(ROOT, "forests", "defaultZeroForest.txt"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--verify-pairs", type=int, default=16, help="pairs per rank checked against the oracle (spread over the batch)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the single-pair, two-stream and PCIe-inclusive side measurements (profiling runs: every "
                         "launch of a kernel then has the same grid, so rocprofv3's averages match the HIP-event ones)")
    ap.add_argument("--host-path", action="store_true",
                    help="measure the host-to-host call on every rank even with --no-extras (rehearsals of the N > 1 line)")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="contexts (HIP streams + workspaces) the steps alternate over; 1 = strictly serial steps")
    return ap.parse_args()


class HostChild:
    """This rank's torch-free child for the host-to-host leg (tools/pcie_inclusive.py --serve): the child makes the
    synchronous gpc_hip_match_batch calls on the library's own ROCm runtime, this process tells it when (after the
    barrier between the ranks) and collects the child's own timing of each call."""

    def __init__(self, B, W, H, forest, dev_index):
        import subprocess
        import queue
        import threading
        self.p = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "pcie_inclusive.py"), "--serve", str(B), str(W),
                                   str(H), forest, str(dev_index)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, bufsize=1)
        self.q = queue.Queue()

        def pump(stream, q):
            for line in stream:
                q.put(line)
            q.put(None)
        threading.Thread(target=pump, args=(self.p.stdout, self.q), daemon=True).start()

    def _line(self, timeout=float(os.environ.get("GPC_BENCH_CHILD_TIMEOUT_S", "600"))):
        # (a child that never answers must not park this rank in a barrier until somebody's outer time limit: its lines
        #  come through a reader thread and a queue with a deadline)
        import queue
        try:
            line = self.q.get(timeout=timeout)
        except queue.Empty:
            self.kill()
            raise RuntimeError("host-to-host child did not answer within %.0f s" % timeout)
        if line is None:
            raise RuntimeError("host-to-host child ended early (status %r)" % (self.p.poll(),))
        return line.strip()

    def wait_ready(self):
        while self._line() != "ready":
            pass

    def call(self, cmd="go"):
        self.p.stdin.write(cmd + "\n")
        self.p.stdin.flush()
        return float(self._line())

    def finish(self):
        self.p.stdin.write("done\n")
        self.p.stdin.flush()
        rec = json.loads(self._line())
        self.p.stdin.close()
        self.p.wait(timeout=120)
        return rec

    def kill(self):
        if self.p.poll() is None:
            self.p.kill()


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
    """Single-thread CPU time of the same timed region (host images -> host supports) on a bounded
    sample of the workload.  Uses the reference's own SSE kernels (oracle/_ref) + a C++ port of the
    inference.hpp glue when that build is present, otherwise the plain C oracle (a slower, scalar port)."""
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
    what = ("the reference's SSE kernels (lib/gpc/filter.hpp compiled in place: oracle/_ref) + a C++ port of the "
            "inference.hpp glue (24-byte descriptors, std::sort, zero-filled code image); the port omits the reference's "
            "by-value deep copies, so it is a little FASTER than the reference binary") if use_ref else \
        "scalar C restatement (oracle/gpc_oracle.c, -O3 build); the reference tree is not on this box"
    return {
        "value": round(2.0 * W * H / med / 1e6, 3),
        "unit": "Mpix/s",
        "cores": 1,
        "kind": "port",
        "ms_per_pair": round(med * 1e3, 3),
        "timed_region": "host images -> host supports (the reference's t0..t2, sparsematch.cpp:45-52)",
        "sample": "%d pairs %dx%d (s=i, D=8+i%%64), median; %s; cold first pair excluded by median" % (len(times), W, H, what),
        "all_cores": allc,
    }


def compulsory_bytes(W, H, B, M_step, fused=True):
    """HBM bytes each kernel must move per launch given its input / output formats:
    what the roofline of that kernel is priced on.  M = supports of the step.  `fused`: the join writes the 12-byte
    supports itself (one launch); otherwise it stages 4-byte words and k_gather_rows expands them."""
    rows = H - 26
    # the gradient image between the two is one BIT per pixel in the batched pipelines (GPC_HIP_NO_GRAD_BITS keeps the bytes)
    g = 1.0 if os.environ.get("GPC_HIP_NO_GRAD_BITS") else 0.125
    d = {
        "k_preprocess": 2.0 * (2.0 + g) * W * H * B,              # raw read (1 B/px), smooth + grad written, both images
        "k_hash": 2.0 * (5.0 + g) * W * H * B,                    # smooth + grad read, dense 4-byte code image written, both images
    }
    if fused:
        d["k_row_join"] = 8.0 * W * rows * B + 12.0 * M_step      # both code rows read, 12-byte supports written
    else:
        d["k_row_join"] = 8.0 * W * rows * B + 4.0 * M_step + 4.0 * rows * B   # both code rows read, packed supports + row counts written
        d["k_gather_rows"] = 16.0 * M_step + 4.0 * rows * B       # packed supports read, 12-byte supports written
    return d


def bytes_text(fused):
    g = 1.0 if os.environ.get("GPC_HIP_NO_GRAD_BITS") else 0.125
    return {"k_row_join": "8 B per pixel of the joined rows + %d B per support" % (12 if fused else 4),
            "k_hash": "%.2f B per pixel of a pair (smooth 1 + gradient %.3f read, code 4 written, two images)" % (2 * (5 + g), g),
            "k_preprocess": "%.2f B per pixel of a pair (raw 1 read, smooth 1 + gradient %.3f written, two images)" % (2 * (2 + g), g),
            "k_gather_rows": "16 B per support"}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, BEFORE anything initialises a GPU here
        from opengpc_amd.launch import launch_local_ranks
        raise SystemExit(launch_local_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                            backend=os.environ.get("GPC_DIST_BACKEND", "nccl")))
    import numpy as np
    import torch

    from opengpc_amd import dist as gdist

    rank, world, local_rank = gdist.env_world()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE %d: the launcher's rank count and --gpus must agree "
                         "(run `python bench.py --gpus N` plainly and it starts its own N ranks)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # GPC_DIST_BACKEND=gloo: rehearsal of the N > 1 line on a box with fewer GPUs than ranks (the ranks share the devices
    # there are, rank r on device r mod count; RCCL refuses two ranks on one device, gloo carries the barriers instead)
    backend = os.environ.get("GPC_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    gdist.init(backend, dev)  # nccl == RCCL on ROCm; no-op for a single process
    stat_dev = dev if backend == "nccl" else "cpu"

    import opengpc_amd as g
    from opengpc_amd.synth import synth_batch

    W, H, B = args.width, args.height, args.batch
    P = max(1, args.pipeline)
    settings = g.Settings.sparsematch()
    ctxs, streams = [], []
    for _ in range(P):  # one context = one HIP stream + one set of workspaces
        c = g.Context(dev_index)
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
    # every in-flight step owns its outputs (gpc_support = 12 bytes); zero-filled so whole arrays can be compared
    d_outs = [torch.zeros((B, cap, 3), dtype=torch.int32, device=dev) for _ in range(P)]
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

    # ---- which kernel dominates?  a few serial steps with every kernel bracketed by HIP events on the stream
    #      the kernels run on (not part of `value`); the slowest one is then the only kernel bracketed inside
    #      the timed windows, so the other launches carry no event records
    def bracket_all(nsteps):
        ctx.enable_kernel_timing(True)
        ctx.reset_kernel_timing()
        for _ in range(nsteps):
            ctx.match_batch_device(d_L.data_ptr(), d_R.data_ptr(), W, H, B, settings, d_out.data_ptr(), cap,
                                   d_counts.data_ptr(), d_ncand.data_ptr())
        kt = {k: v for k, v in ctx.kernel_times().items() if v[1]}
        ctx.enable_kernel_timing(False)
        return kt

    ktimes = bracket_all(4)
    launch_names = ctx.kernel_launch_names()
    dom_slot = max(ktimes.items(), key=lambda kv: kv[1][0] / kv[1][1])[0]

    for c in ctxs:
        c.enable_kernel_timing(True, only=[dom_slot])
        c.reset_kernel_timing()
    step_no[0] = 0
    windows = [gdist.timed_steps(step, args.steps, device_sync) for _ in range(max(1, args.windows))]
    dom_ms, dom_n = 0.0, 0
    for c in ctxs:
        ms, n = c.kernel_times()[dom_slot]
        dom_ms += ms
        dom_n += n
        c.enable_kernel_timing(False)

    # per-kernel table: serial steps with every kernel bracketed, run right behind the timed windows (the clocks are
    # where the windows left them: the first launches after an idle second run several per cent slower)
    bracket_all(5)            # (the first launches with every kernel bracketed run a few per cent slower: not counted)
    ktimes = bracket_all(20)

    counts = d_counts.cpu().numpy().astype(np.int64)
    ncand = d_ncand.cpu().numpy().astype(np.int64)

    # ---- BASELINE configs[3] AS WRITTEN: ONE batch of 256 pairs shared by the N GPUs, 256 / N pairs per rank and step
    #      (the headline above scales weakly: 256 pairs per rank at any N).  The same barrier-bracketed windows, fewer of
    #      them; at one rank also the 32-pair step that is a rank's share at N = 8.  Not part of `value`.
    strong_windows, share8_windows = [], []
    Bs = max(1, min(B, 256 // world))
    if not args.no_extras or args.host_path:
        def step_n(npairs):
            def f():
                ctx.match_batch_device(d_L.data_ptr(), d_R.data_ptr(), W, H, npairs, settings, d_outs[-1].data_ptr(), cap,
                                       d_cnts[-1].data_ptr(), d_ncs[-1].data_ptr())
            return f
        for _ in range(5):
            step_n(Bs)()
        device_sync()
        strong_windows = [gdist.timed_steps(step_n(Bs), args.steps, device_sync) for _ in range(40)]
        if world == 1 and B >= 32:
            for _ in range(5):
                step_n(32)()
            device_sync()
            share8_windows = [gdist.timed_steps(step_n(32), args.steps, device_sync) for _ in range(40)]
        # (d_outs[-1] was overwritten by shorter batches: the arrays compared below are d_out's, written by the timed windows)
        if P == 1:
            ctx.match_batch_device(d_L.data_ptr(), d_R.data_ptr(), W, H, B, settings, d_out.data_ptr(), cap,
                                   d_counts.data_ptr(), d_ncand.data_ptr())
            device_sync()

    # ---- parity gate on the bench's own data: pairs spread over this rank's batch against the oracle
    #      (checker only; the -O3 build of the C restatement, held equal to the plain build by tests/)
    n_verified, verified = 0, None
    if not args.no_verify:
        from oracle.pyoracle import Oracle, sparsematch_settings
        o = Oracle(fast=True)
        rc, f = o.read_forest(args.forest, W, H)
        nv = max(1, min(B, args.verify_pairs))
        pick = sorted(set([int(round(j * (B - 1) / max(nv - 1, 1))) for j in range(nv)]))
        verified = True
        for j in pick:
            want, nl, nr = o.match_pair(Lh[j], Rh[j], f, sparsematch_settings())
            got = d_out[j, : int(counts[j])].cpu().numpy()
            ok = (len(want) == int(counts[j]) and (nl, nr) == tuple(int(v) for v in ncand[j])
                  and np.array_equal(got[:, 0], want["x"]) and np.array_equal(got[:, 1], want["y"])
                  and np.array_equal(got[:, 2].view(np.float32), want["d"]))
            verified = verified and bool(ok)
            n_verified += 1
        if not verified:
            print("bench.py: rank %d: GPU supports differ from the oracle" % rank, file=sys.stderr)

    # ---- the reference's own timed region on EVERY rank at once: raw pairs in page-locked host memory -> gpc_support
    #      arrays in host memory, one synchronous gpc_hip_match_batch call per rank and repetition, barrier-bracketed
    #      (the ranks of a node share the host's memory bandwidth and CPUs, so they are timed together); the slowest
    #      rank of a repetition counts.  The host API has no torch dependency, and a process that has imported torch runs
    #      every HIP library on the ROCm runtime bundled with the torch wheel (an older one than the /opt/rocm this library
    #      is built against).  Which of the two legs is the slower one has swapped between rounds (r03: this process 8.7 ms,
    #      child 5.6; r04: child 8.5, this process 6.6), so the line carries the stages of every leg, the CPUs the workers
    #      ran on and the NUMA node the buffers' pages lie on (pcie_inclusive.stages / .host / .pages_on_node).  Every rank makes the calls in a
    #      CHILD process without torch -- the C++ caller's situation -- and this process only carries the barriers and
    #      tells the child when to go.  At one rank the same call is also timed inside this process, for the record.
    host_reps, host_times, host_ok, host_threads, child_rec = 15, [], True, 0, None
    packed_reps, packed_times = 9, []
    inproc_times = []
    if not args.no_extras or args.host_path:
        child = HostChild(B, W, H, args.forest, dev_index)
        try:
            child.wait_ready()                    # 5 untimed calls: page-locked buffers touched, workers and clocks warm

            def child_call():
                res_t.append(child.call())
            res_t = []
            gdist.timed_calls(child_call, host_reps)   # barrier before each repetition; the child times its own call
            host_times = res_t
            # and the packed variant of the same call (gpc_hip_match_batch_packed: the records stay 4 bytes each in host
            # memory -- what a node of 8 ranks has the memory bandwidth for)
            gdist.timed_calls(lambda: packed_times.append(child.call("gop")), packed_reps)
            child_rec = child.finish()
        finally:
            child.kill()
        ch_counts = np.asarray(child_rec.pop("counts"), np.int64)
        crc = child_rec.pop("crc32")
        import zlib
        host_ok = bool(child_rec["status"] == 0 and np.array_equal(ch_counts, counts))
        for j in sorted(set((0, B // 2, B - 1))):   # the device path's 12-byte records of three pairs, byte for byte
            host_ok = host_ok and zlib.crc32(d_out[j, : int(counts[j])].cpu().numpy().tobytes()) == crc[str(j)]
        host_threads = int(child_rec.get("host", {}).get("expand_threads", 0))
        host_ok = host_ok and bool(child_rec.get("packed", {}).get("identical_to_expanded", False))
        if world == 1:
            capi_cap = 300000
            Lp, Rp = ctx.pinned_empty(Lh.shape, np.uint8), ctx.pinned_empty(Rh.shape, np.uint8)
            Lp[:] = Lh
            Rp[:] = Rh
            outb = ctx.pinned_empty((B, capi_cap), g.SUPPORT_DTYPE)
            res = {}

            inproc_stages = []

            def host_call():
                res["r"] = ctx.match_batch(Lp, Rp, settings, capi_cap, out=outb)
                inproc_stages.append(ctx.batch_stages())
            for _ in range(3):
                host_call()
            del inproc_stages[:]
            inproc_times = sorted(gdist.timed_calls(host_call, 9))
            from opengpc_amd.hostinfo import cpu_nodes, current_cpu, pages_nodes, stage_summary
            inproc_info = {"stages_ms": stage_summary(inproc_stages), "pages_on_node": {"out": pages_nodes(outb), "images": pages_nodes(Lp)},
                           "worker_cpus": ctx.worker_cpus(), "calling_thread_cpu": current_cpu(),
                           "calling_thread_node": cpu_nodes().get(current_cpu(), -1)}
            o_, c_, n_, st_ = res["r"]
            host_ok = host_ok and bool(st_ == 0 and np.array_equal(c_.astype(np.int64), counts))
            del Lp, Rp, outb

    # O(100 B) per rank over xGMI: timing / counters only, never pixel data
    row = [float(B), float(ncand.sum()), float(counts.sum()), 1.0 if verified in (True, None) else 0.0,
           float(n_verified), 1.0 if host_ok else 0.0, float(host_threads)] + [float(t) for t in host_times] + \
          [float(t) for t in packed_times] + [float(t) for t in windows]
    allr = gdist.gather_stats(row, device=stat_dev).numpy()
    strong = None
    if strong_windows:
        sw = np.sort(gdist.gather_stats(strong_windows, device=stat_dev).numpy().max(axis=0))   # per window: the slowest rank
        ts = float(sw[len(sw) // 2]) / args.steps
        strong = {"pairs_per_rank_per_step": Bs, "pairs_per_step": Bs * world, "ms_per_step": round(ts * 1e3, 4),
                  "value": round(2.0 * W * H * Bs * world / ts / 1e6, 1), "unit": "Mpix/s",
                  "ms_per_step_min": round(float(sw[0]) / args.steps * 1e3, 4), "ms_per_step_max": round(float(sw[-1]) / args.steps * 1e3, 4),
                  "note": "BASELINE configs[3] as written: one batch of 256 pairs over the %d GPU(s), %d pairs per rank and step, "
                          "device-resident like `value`; median of %d windows of %d steps" % (world, Bs, len(sw), args.steps)}
    if float(allr[:, 3].min()) < 1.0:
        raise SystemExit("bench.py: GPU supports differ from the oracle on some rank -- refusing to report a number")
    nh, npk = len(host_times), len(packed_times)
    win = np.sort(allr[:, 7 + nh + npk:].max(axis=0))     # per window: the slowest rank
    host_packed = None
    if npk:
        hp = np.sort(allr[:, 7 + nh:7 + nh + npk].max(axis=0))
        tpk = float(hp[len(hp) // 2])
        host_packed = {"ms_per_call": round(tpk * 1e3, 3), "value": round(2.0 * W * H * float(allr[:, 0].sum()) / tpk / 1e6, 1),
                       "unit": "Mpix/s", "pairs_per_call_per_rank": B, "ranks": int(world),
                       "ms_per_call_min": round(float(hp[0]) * 1e3, 3), "ms_per_call_max": round(float(hp[-1]) * 1e3, 3),
                       "note": "gpc_hip_match_batch_packed on every rank at the same moment: host images -> packed records in host "
                               "memory (4 bytes per support + row counts: what crosses the link), the same pipeline without the "
                               "expansion to 12-byte ndb::Support records; median of %d repetitions, same children" % npk}
    host_all = None
    if nh:
        hs = np.sort(allr[:, 7:7 + nh].max(axis=0))  # per repetition: the slowest rank
        th = float(hs[len(hs) // 2])
        host_all = {"ms_per_call": round(th * 1e3, 3), "value": round(2.0 * W * H * float(allr[:, 0].sum()) / th / 1e6, 1),
                    "unit": "Mpix/s", "pairs_per_call_per_rank": B, "ranks": int(world),
                    "ms_per_call_min": round(float(hs[0]) * 1e3, 3), "ms_per_call_max": round(float(hs[-1]) * 1e3, 3),
                    "identical_to_device_path": bool(float(allr[:, 5].min()) >= 1.0),
                    "expand_threads_per_rank": int(allr[:, 6].min()),
                    "measured_in": "one child process without torch per rank (the library's own ROCm runtime), told when to "
                                   "go by its rank between the ranks' barriers; each call timed by the child itself",
                    "note": "every rank's synchronous gpc_hip_match_batch call at the same moment (barrier before each "
                            "repetition), all pairs of all ranks / the slowest rank's time, median of %d repetitions after 5 "
                            "untimed" % nh}
    t_med = float(win[len(win) // 2])
    pairs_per_step = float(allr[:, 0].sum())

    if rank == 0:
        mpix_per_step = 2.0 * W * H * pairs_per_step / 1e6
        value = mpix_per_step * args.steps / t_med

        # ---- roofline of the dominant kernel: HBM bytes it must move (its formats) / HIP-event time
        N_step = float(ncand.sum())
        M_step = float(counts.sum())
        fused = "k_gather_rows" not in ktimes   # the join wrote the supports itself (k_rowjoin.h, FUSE)
        alg = compulsory_bytes(W, H, B, M_step, fused)
        BYTES_TEXT = bytes_text(fused)
        kinfo = {}
        for name, (ms, n) in ktimes.items():
            kinfo[name] = {"kernel": launch_names.get(name, name), "avg_us": round(1e3 * ms / n, 2), "launches": n}
            if name in alg:
                kinfo[name]["hbm_GBs"] = round(alg[name] / (ms / n * 1e-3) / 1e9, 1)
                kinfo[name]["hbm_frac"] = round(alg[name] / (ms / n * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        dom_us = 1e3 * dom_ms / max(dom_n, 1)
        achieved = alg.get(dom_slot, 0.0) / (dom_us * 1e-6) / 1e9 if dom_ms > 0 else 0.0
        # counters of the same workload from the committed rocprofv3 --pmc passes (never measured here)
        prof, valu_prof = None, None
        ppath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(ppath):
            try:
                with open(ppath) as fh:
                    tj = json.load(fh)
                prof = tj.get("%s@%dx%dx%d" % (dom_slot, W, H, B))
                # (the kernels are bound by vector-instruction issue, not by HBM: the committed PMC passes say how busy the VALUs were)
                valu_prof = {k.split("@")[1]: v for k, v in tj.items() if k.startswith("valu_issue@") and k.endswith("@%dx%dx%d" % (W, H, B))} or None
            except Exception:
                prof = None
        # SURVEY.md 8d's whole-pipeline figure: A = 10*W*H + 48*N + 12*M bytes per pair of a sort-based
        # matcher whose sort streams through HBM (this build keeps it in LDS: comparable across builds,
        # not a statement about traffic)
        a_pair = (10.0 * W * H * B + 48.0 * N_step + 12.0 * M_step) / B
        roofline = {
            "bound": "hbm",
            "kernel": launch_names.get(dom_slot, dom_slot),
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": None,
            "traffic_from_profiles": prof,
            "valu_issue_from_profiles": valu_prof,
            "alg_bytes_per_launch": alg.get(dom_slot),
            "avg_launch_us": round(dom_us, 2),
            "launches_timed": dom_n,
            "pipeline_alg_bytes_per_pair": a_pair,
            "pipeline_alg_GBs_equivalent": round(a_pair * pairs_per_step * args.steps / t_med / 1e9, 1),
            "pipeline_compulsory_frac": round(sum(alg.values()) * world * args.steps / t_med / 1e9 / HBM_PEAK_GBS, 4),
            "note": "achieved = bytes the dominant kernel must read + write per launch given its formats (DESIGN.md 4: "
                    "%s) / mean HIP-event duration of its %d launches inside the timed windows, on the stream it runs on.  "
                    "The kernel keeps the reference's sort in LDS and is bound by LDS / issue, not by HBM: `frac` says how "
                    "far below the HBM roof that leaves it.  `traffic` is not measured by this script; "
                    "`traffic_from_profiles` = 2*FETCH_SIZE + WRITE_SIZE of the committed rocprofv3 --pmc passes.  "
                    "pipeline_alg_GBs_equivalent = SURVEY 8d's A*pairs/t: the rate a matcher whose sort streams through HBM "
                    "would need for this throughput -- NOT traffic and not a fraction of anything (this build's sort never "
                    "leaves LDS, so it can exceed the peak); pipeline_compulsory_frac = sum of the kernels' compulsory bytes / "
                    "step time / peak." % (BYTES_TEXT.get(dom_slot, ""), dom_n),
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

        # ---- `pcie_inclusive`: the reference's own timed region (host images -> host gpc_support arrays,
        #      sparsematch.cpp:45-52) -- the all-ranks record measured above in torch-free children; at one rank with
        #      the same call inside this (torch) process beside it
        pcie, single_h2h = None, None
        if host_all:
            pcie = dict(host_all)
            pcie["timed_region"] = "host images -> host gpc_support arrays (sparsematch.cpp:45-52), median of %d calls after 5 untimed" % nh
            if child_rec:
                single_h2h = child_rec.pop("single_pair_host_to_host", None)
                for k in ("host_buffers", "bytes_in", "bytes_over_the_link_out", "bytes_delivered", "host", "pages_on_node"):
                    if k in child_rec:
                        pcie[k + ("_rank0" if world > 1 and k != "host_buffers" else "")] = child_rec[k]
                # where a call's time goes (host clock, ms since entry, medians): the stages of the expanded call, and of
                # the packed call of the same child beside it (the same pipeline without the expansion)
                pcie["stages"] = {"expanded": child_rec.get("stages_ms"), "packed": child_rec.get("packed", {}).get("stages_ms"),
                                  "note": "ms since the call's entry when the host saw: the last chunk's upload complete, the "
                                          "last chunk's kernels done, the last chunk of packed records landed in host memory, "
                                          "the delivery done (gpc_hip_batch_stages); rank 0's child"}
            if inproc_times:
                ti = inproc_times[len(inproc_times) // 2]
                pcie["stages"]["in_this_process"] = inproc_info.get("stages_ms")
                pcie["in_this_process_host"] = {k: v for k, v in inproc_info.items() if k != "stages_ms"}
                pcie["in_this_process"] = {"ms_per_call": round(ti * 1e3, 3), "value": round(2.0 * W * H * B / ti / 1e6, 1),
                                           "unit": "Mpix/s", "ms_per_call_min": round(inproc_times[0] * 1e3, 3),
                                           "ms_per_call_max": round(inproc_times[-1] * 1e3, 3),
                                           "note": "the same call from this process, whose HIP runtime is the one bundled with torch"}

        # the CPU leg: rank 0 only, at any N (after the last collective: the other ranks are done and idle)
        cpu = None
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(args, W, H)

        # A STREAM of batches through the library's two lanes (gpc_hip_set_pipeline(2): batch k+1's k_preprocess and k_hash
        # run beside batch k's join); reported beside the strictly serial headline, whose per-kernel event times and
        # roofline stay clean.  Two output sets alternate (two calls are in flight at a time).
        two = None
        if world == 1 and P == 1 and not args.no_extras:
            c2 = g.Context(dev_index)
            c2.load_forest(args.forest, W, H)
            st2 = torch.cuda.Stream(device=dev)
            c2.set_stream(st2.cuda_stream)
            c2.set_pipeline(2)
            o2 = torch.zeros((B, cap, 3), dtype=torch.int32, device=dev)
            n2 = torch.zeros(B, dtype=torch.int32, device=dev)
            m2 = torch.zeros((B, 2), dtype=torch.int32, device=dev)
            o3 = torch.zeros((B, cap, 3), dtype=torch.int32, device=dev)
            n3 = torch.zeros(B, dtype=torch.int32, device=dev)
            m3 = torch.zeros((B, 2), dtype=torch.int32, device=dev)
            sets = [(o2, n2, m2), (o3, n3, m3)]

            def step2(i):
                o, n, m = sets[i & 1]
                c2.match_batch_device(d_L.data_ptr(), d_R.data_ptr(), W, H, B, settings, o.data_ptr(), cap, n.data_ptr(), m.data_ptr())
            for i in range(6):
                step2(i)
            c2.synchronize()
            lane_t = []
            for _ in range(15):
                device_sync(); c2.synchronize()
                t2 = time.perf_counter()
                for i in range(args.steps):
                    step2(i)
                c2.synchronize()
                lane_t.append((time.perf_counter() - t2) / args.steps)
            dt2 = sorted(lane_t)[len(lane_t) // 2]
            # every pair, every support, both output sets (the arrays were zero-filled; only valid entries are ever written)
            same = bool(all(torch.equal(n, d_counts) and torch.equal(m, d_ncand) and torch.equal(o, d_out) for o, n, m in sets))
            two = {"lanes": 2, "ms_per_step": round(dt2 * 1e3, 4), "value": round(mpix_per_step / dt2, 1),
                   "unit": "Mpix/s", "identical_outputs_all_pairs": same,
                   "note": "gpc_hip_set_pipeline(ctx, 2): consecutive batches alternate between two lanes of ONE context; "
                           "median of 15 windows of %d steps; not the headline (its kernels overlap, so no clean per-kernel times)" % args.steps}
            # the same at 32 pairs per step (a rank's share of configs[3] at 8 GPUs): there the join's fill and drain are a
            # quarter of its launch, which is what the neighbouring batch's kernels fill
            if B >= 32:
                for o, n, m in sets:
                    n.zero_(); m.zero_()

                def step32(i):
                    o, n, m = sets[i & 1]
                    c2.match_batch_device(d_L.data_ptr(), d_R.data_ptr(), W, H, 32, settings, o.data_ptr(), cap, n.data_ptr(), m.data_ptr())
                for i in range(6):
                    step32(i)
                c2.synchronize()
                lane_t = []
                for _ in range(15):
                    device_sync(); c2.synchronize()
                    t2 = time.perf_counter()
                    for i in range(args.steps):
                        step32(i)
                    c2.synchronize()
                    lane_t.append((time.perf_counter() - t2) / args.steps)
                dt32 = sorted(lane_t)[len(lane_t) // 2]
                same32 = bool(all(torch.equal(n[:32], d_counts[:32]) and torch.equal(m[:32], d_ncand[:32]) and
                                  torch.equal(o[:32], d_out[:32]) and int(n[32:].abs().sum().item()) == 0 for o, n, m in sets))
                two["share_at_8_gpus"] = {"pairs_per_step": 32, "ms_per_step": round(dt32 * 1e3, 4),
                                          "value": round(2.0 * W * H * 32 / dt32 / 1e6, 1), "unit": "Mpix/s",
                                          "identical_outputs_all_pairs": same32}
            c2.close()
            del o2, o3

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
            "ms_per_step": round(1e3 * t_med / args.steps, 4),
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
                "timed_region": "raw pairs in HBM -> supports in HBM (no PCIe); pcie_inclusive = the reference's host-to-host region",
            },
            "windows": {"count": len(win), "steps_each": args.steps, "reported": "median",
                        "ms_per_step_min": round(1e3 * float(win[0]) / args.steps, 4),
                        "ms_per_step_max": round(1e3 * float(win[-1]) / args.steps, 4),
                        "gpu_busy_s": round(float(win.sum()), 3)},
            "pairs_per_s": round(pairs_per_step * args.steps / t_med, 1),
            "candidates_per_pair": round(float(allr[:, 1].sum()) / pairs_per_step, 1),
            "supports_per_pair": round(float(allr[:, 2].sum()) / pairs_per_step, 1),
            "verified_vs_oracle": None if args.no_verify else True,
            "verified_pairs": int(allr[:, 4].sum()),
            "roofline": roofline,
            "cpu_baseline": cpu,
            "pcie_inclusive": pcie,
            "single_pair": single,
            "single_pair_host_to_host": single_h2h,
            "host_to_host_all_ranks": host_all,
            "host_to_host_packed_all_ranks": host_packed,
            "two_lane_pipeline": two,
            "strong_256": strong,
        }
        if strong:
            # against N times what ONE of these GPUs does on 256 pairs (the weak-scaling value / N)
            strong["efficiency_vs_n_times_one_gpu_256_pairs"] = round(strong["value"] / value, 4)
        if share8_windows:
            s8 = sorted(share8_windows)
            t8 = s8[len(s8) // 2] / args.steps
            line["strong_256_share_at_8_gpus"] = {
                "pairs_per_step": 32, "ms_per_step": round(t8 * 1e3, 4), "value": round(2.0 * W * H * 32 / t8 / 1e6, 1), "unit": "Mpix/s",
                "fraction_of_the_256_pair_rate": round(2.0 * W * H * 32 / t8 / 1e6 / value, 4),
                "note": "a rank's share of configs[3] at 8 GPUs (32 pairs per step) on this one GPU: what the strong-scaling "
                        "leg of an 8-GPU run costs per rank"}
        if cpu:
            # like with like: the CPU leg is host -> host, so is pcie_inclusive
            if pcie:
                line["speedup_vs_cpu_1thread"] = round(pcie["value"] / cpu["value"], 1)
            line["speedup_device_resident_vs_cpu_1thread"] = round(value / cpu["value"], 1)
        print(json.dumps(line), flush=True)

    for c in ctxs:
        c.close()
    gdist.finalize()


if __name__ == "__main__":
    main()
