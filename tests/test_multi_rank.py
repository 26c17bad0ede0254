"""The N>1 path on CPU: two gloo ranks shard a batch of pairs exactly like bench.py does
(pair i -> rank i mod N, no data-path collective), each rank matches its shard (with the CPU
oracle standing in for the GPU in this test only), and the gathered counters / timing
reduction equal the single-process result."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from opengpc_amd import dist as gdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, PER_RANK = 96, 64, 3


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def match_shard(indices):
    from oracle.pyoracle import Oracle, sparsematch_settings
    from opengpc_amd.synth import synth_batch
    o = Oracle()
    rc, f = o.read_forest(os.path.join(ROOT, "forests", "defaultZeroForest.txt"), W, H)
    L, R = synth_batch(W, H, indices)
    out = []
    for j in range(len(indices)):
        supp, nl, nr = o.match_pair(L[j], R[j], f, sparsematch_settings())
        out.append((len(supp), nl + nr, o.fnv(np.ascontiguousarray(supp["x"]))))
    return out


def worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w = gdist.init("gloo")
    assert (r, w) == (rank, world)
    idx = gdist.shard_indices(rank, world, PER_RANK)
    res = {}

    def step():
        res["out"] = match_shard(idx)

    elapsed = gdist.timed_steps(step, 2, lambda: None)
    n_supp = sum(v[0] for v in res["out"])
    n_cand = sum(v[1] for v in res["out"])
    stats = gdist.gather_stats([elapsed, len(idx), n_cand, n_supp])
    if rank == 0:
        q.put((stats.numpy().tolist(), gdist.reduce_job(stats, 2, 2 * W * H)))
    q.put((rank, idx, res["out"]))
    gdist.finalize()


def test_two_ranks_shard_pairs_without_overlap():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world + 1)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    summary = [g for g in got if isinstance(g[1], dict)][0]
    per_rank = sorted([g for g in got if not isinstance(g[1], dict)], key=lambda g: g[0])
    # every global pair is owned by exactly one rank
    all_idx = sorted(i for _, idx, _ in per_rank for i in idx)
    assert all_idx == list(range(world * PER_RANK))
    for rank, idx, _ in per_rank:
        assert all(gdist.owner_of(i, world) == rank for i in idx)
    # sharded results == the same pairs matched in one process
    single = match_shard(list(range(world * PER_RANK)))
    for rank, idx, out in per_rank:
        assert out == [single[i] for i in idx]
    stats, job = summary
    assert len(stats) == world and all(s[1] == PER_RANK for s in stats)
    assert sum(s[3] for s in stats) == sum(v[0] for v in single)
    assert job["pairs_per_step"] == world * PER_RANK
    assert job["t_max"] == max(s[0] for s in stats)
    assert job["mpix_per_s"] == pytest.approx(2 * W * H * world * PER_RANK * 2 / job["t_max"] / 1e6)


RANK_PROG = """
import json, os, sys
r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == str(r) and os.environ["LOCAL_WORLD_SIZE"] == str(w)
assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
if len(sys.argv) > 1 and sys.argv[1] == "fail" and r == 1:
    sys.exit(7)
if len(sys.argv) > 1 and sys.argv[1] == "fail":
    import time
    time.sleep(60)          # the launcher must stop this rank when rank 1 fails
if len(sys.argv) > 1 and sys.argv[1] == "hang":
    import signal, subprocess, time
    if r == 1:
        signal.signal(signal.SIGTERM, signal.SIG_IGN)     # a rank that does not stop when asked
    kid = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(300)"])    # a rank's own child (bench.py's HostChild)
    print("pids %d %d" % (os.getpid(), kid.pid), file=sys.stderr, flush=True)
    time.sleep(300)
print(json.dumps({"rank": r, "world": w}))
"""


def run_launcher(tmp_path, n, ndev, backend, arg=None):
    """launch_local_ranks in a process of its own (its rank 0 inherits that process's stdout)."""
    prog = tmp_path / "rank_prog.py"
    prog.write_text(RANK_PROG)
    drv = ("import sys; sys.path.insert(0, %r)\n"
           "from opengpc_amd.launch import launch_local_ranks\n"
           "assert 'torch' not in sys.modules\n"
           "rc = launch_local_ranks(%d, [sys.executable, %r] + %r, backend=%r, count_devices=lambda: %d)\n"
           "assert 'torch' not in sys.modules      # the launching process stays torch- and GPU-free\n"
           "sys.exit(rc)\n" % (ROOT, n, str(prog), [arg] if arg else [], backend, ndev))
    return subprocess.run([sys.executable, "-c", drv], capture_output=True, text=True, timeout=120)


def test_launcher_starts_n_ranks_and_relays_rank0(tmp_path):
    import json
    r = run_launcher(tmp_path, 3, 8, "nccl")
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout.strip()) == {"rank": 0, "world": 3}          # rank 0's line alone on stdout
    others = sorted(json.loads(l)["rank"] for l in r.stderr.strip().splitlines() if l.startswith("{"))
    assert others == [1, 2]


def test_launcher_refuses_fewer_devices_than_ranks(tmp_path):
    r = run_launcher(tmp_path, 4, 1, "nccl")
    assert r.returncode == 2 and "refusing" in r.stderr and r.stdout == ""
    r = run_launcher(tmp_path, 2, 0, "gloo")
    assert r.returncode == 2 and "no HIP device" in r.stderr
    r = run_launcher(tmp_path, 2, 1, "gloo")        # rehearsal: the ranks share the one device
    assert r.returncode == 0 and '"world": 2' in r.stdout


def test_launcher_failing_rank_stops_the_job(tmp_path):
    import time
    t0 = time.time()
    r = run_launcher(tmp_path, 3, 8, "nccl", "fail")
    assert r.returncode == 7 and "rank 1 exited with status 7" in r.stderr
    assert time.time() - t0 < 30     # ranks 0 and 2 were stopped, not waited for


def test_launcher_forwards_sigterm_and_leaves_nothing_behind(tmp_path):
    """A driver's `timeout` sends SIGTERM to the launcher: every rank's process GROUP (the rank and its children) is told to
    stop, a rank that ignores that is killed after the grace period, and the launcher returns 128 + 15."""
    import signal
    import time
    prog = tmp_path / "rank_prog.py"
    prog.write_text(RANK_PROG)
    drv = ("import sys; sys.path.insert(0, %r)\n"
           "from opengpc_amd.launch import launch_local_ranks\n"
           "sys.exit(launch_local_ranks(2, [sys.executable, %r, 'hang'], backend='nccl', count_devices=lambda: 8))\n" % (ROOT, str(prog)))
    env = dict(os.environ, GPC_LAUNCH_GRACE_S="2")
    p = subprocess.Popen([sys.executable, "-c", drv], stderr=subprocess.PIPE, text=True, env=env)
    pids = []
    t0 = time.time()
    while len(pids) < 4 and time.time() - t0 < 60:
        line = p.stderr.readline()
        if line.startswith("pids "):
            pids += [int(v) for v in line.split()[1:]]
    assert len(pids) == 4
    p.send_signal(signal.SIGTERM)
    rc = p.wait(timeout=30)
    rest = p.stderr.read()
    assert rc == 128 + signal.SIGTERM, rest
    assert "did not stop within" in rest               # rank 1 ignored SIGTERM and was killed
    time.sleep(0.3)
    for pid in pids:                                    # neither a rank nor a rank's child is left
        alive = True
        try:
            os.kill(pid, 0)
            with open("/proc/%d/stat" % pid) as fh:      # (a zombie waiting for its reaper does not hold anything)
                alive = fh.read().split(") ")[1][0] != "Z"
        except (ProcessLookupError, FileNotFoundError):
            alive = False
        assert not alive, pid


def test_bench_plainly_with_gpus_2_without_a_gpu_fails_loudly():
    """`python bench.py --gpus 2` is the driver's command: without a device it must say so and return non-zero (here, in
    the CPU container) -- never a line with fewer ranks than asked for."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=600)
    import torch
    if torch.cuda.device_count() == 0:
        assert r.returncode == 2 and "no HIP device" in r.stderr and r.stdout == ""


def test_single_process_helpers():
    assert gdist.shard_indices(3, 8, 4) == [3, 11, 19, 27]
    stats = gdist.gather_stats([0.5, 32, 10, 5])
    assert stats.shape == (1, 4)
    job = gdist.reduce_job(stats, 10, 2 * 1024 * 436)
    assert job["pairs_per_s"] == pytest.approx(640.0)


@pytest.mark.gpu
def test_rccl_path_rehearsal_on_one_gpu(tmp_path):
    """bench.py with GPC_FORCE_DIST=1 at world size 1: torch.distributed is initialised with backend "nccl"
    (RCCL), the timed windows run between RCCL barriers and the per-rank row goes through all_gather on the
    device -- the N > 1 code path end to end, on the one GPU a test box has."""
    import json
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, GPC_FORCE_DIST="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--windows", "3", "--batch", "16", "--verify-pairs", "3", "--no-cpu-baseline", "--no-extras",
                          "--host-path"],
                         env=env, check=True, capture_output=True, text=True, timeout=600).stdout
    line = json.loads(out.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["windows"]["count"] == 3
    assert line["verified_vs_oracle"] is True and line["verified_pairs"] == 3
    assert line["value"] > 1000 and line["config"]["parallelism"] == "pairs-dp1"
    assert 0 < line["roofline"]["frac"] <= 1 and line["roofline"]["kernel"].startswith("gpc::k_")
    # the reference's host-to-host region, timed on every rank at once between RCCL barriers (what a SCALE line carries)
    hh = line["host_to_host_all_ranks"]
    assert hh["ranks"] == 1 and hh["pairs_per_call_per_rank"] == 16 and hh["value"] > 100 and hh["identical_to_device_path"] is True
    assert 1 <= hh["expand_threads_per_rank"] <= 32


@pytest.mark.gpu
def test_bench_gpus_2_plainly_starts_two_ranks():
    """The driver's command at N = 2, run plainly (no torchrun, no RANK / WORLD_SIZE): bench.py starts its own two ranks.
    On a one-GPU box the ranks share the device (GPC_DIST_BACKEND=gloo carries the barriers; RCCL refuses two ranks on one
    device); the line must be a complete two-rank line: n_gpus 2, the CPU leg, the host-to-host leg of BOTH ranks measured in
    torch-free children."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT")}
    env.update(GPC_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", GPC_BENCH_NO_SINGLE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--windows", "3", "--batch", "16", "--verify-pairs", "3", "--cpu-seconds", "1"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # ONE JSON line: rank 0's
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "pairs-dp2" and line["scaling"] == "weak"
    assert line["verified_vs_oracle"] is True and line["verified_pairs"] == 6          # 3 per rank
    assert line["pairs_per_s"] > 0 and line["value"] > 1000
    cb = line["cpu_baseline"]
    assert cb and cb["cores"] == 1 and cb["value"] > 0 and cb["kind"] in ("port", "reference")
    hh = line["host_to_host_all_ranks"]
    assert hh["ranks"] == 2 and hh["pairs_per_call_per_rank"] == 16 and hh["identical_to_device_path"] is True
    assert "without torch" in hh["measured_in"] and hh["value"] > 100
    assert line["pcie_inclusive"]["ranks"] == 2 and line["speedup_vs_cpu_1thread"] > 1
    assert "pipeline_frac" not in line["roofline"] and 0 < line["roofline"]["frac"] <= 1
    # the strong-scaling leg (BASELINE configs[3] as written: 256 / N pairs per rank -- here cut to the 16 this run holds)
    sg = line["strong_256"]
    assert sg["pairs_per_rank_per_step"] == 16 and sg["pairs_per_step"] == 32 and sg["value"] > 1000
    assert 0 < sg["efficiency_vs_n_times_one_gpu_256_pairs"] < 4
    # where the host-to-host call's time goes: the stage clock of the expanded and of the packed call (rank 0's child)
    stg = line["pcie_inclusive"]["stages"]
    for leg in ("expanded", "packed"):
        assert 0 < stg[leg]["upload_done"] <= stg[leg]["kernels_done"] <= stg[leg]["last_chunk_landed"] <= stg[leg]["delivered"]
    assert "out" in line["pcie_inclusive"]["pages_on_node_rank0"]
