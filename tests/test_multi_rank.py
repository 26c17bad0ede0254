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
