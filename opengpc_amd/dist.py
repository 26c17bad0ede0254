"""Multi-GPU plumbing: stereo pairs are independent, so the hot path shards with NO data-path
collective.  One process per GPU; torch.distributed (backend "nccl" == RCCL over
xGMI on the GPU box, "gloo" in CPU tests) is used only for the timing barrier and one
all_gather of a few per-rank counters.
"""
import os
import time

import torch
import torch.distributed as dist


def env_world():
    """(rank, world, local_rank) as torchrun exports them; (0, 1, 0) when launched plainly."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend, device=None):
    """Initialises the default process group when WORLD_SIZE > 1.  Returns (rank, world)."""
    rank, world, _ = env_world()
    # GPC_FORCE_DIST=1 initialises the group even for a single rank (rehearsal of the RCCL path on a 1-GPU box)
    if (world > 1 or os.environ.get("GPC_FORCE_DIST")) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_indices(rank, world, per_rank):
    """Weak scaling: every rank owns `per_rank` pairs; global pair i lives on rank i mod world."""
    return [rank + world * j for j in range(per_rank)]


def owner_of(pair_index, world):
    return pair_index % world


def barrier():
    if dist.is_initialized():
        dist.barrier()


def timed_steps(step, steps, device_sync):
    """Times exactly `steps` calls of step(), bracketed by barrier + device sync on both sides.
    Returns this rank's elapsed seconds (take the MAX over ranks with gather_stats)."""
    device_sync()
    barrier()
    device_sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    device_sync()
    barrier()
    return time.perf_counter() - t0


def timed_calls(call, reps):
    """Times `reps` synchronous host calls, every one bracketed by a barrier, so that the ranks of a node run theirs
    at the same moment (they share the host's memory bandwidth and CPUs).  Returns this rank's seconds per call."""
    out = []
    for _ in range(reps):
        barrier()
        t0 = time.perf_counter()
        call()
        out.append(time.perf_counter() - t0)
    barrier()
    return out


def gather_stats(values, device="cpu"):
    """all_gather of a short list of floats; returns a [world][len(values)] float64 tensor on the CPU."""
    local = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    if dist.is_initialized():
        out = [torch.zeros_like(local) for _ in range(dist.get_world_size())]
        dist.all_gather(out, local)
        return torch.stack(out).cpu()
    return local.cpu()[None, :]


def reduce_job(stats, steps, pixels_per_pair):
    """stats: [world][>=2] rows of (elapsed_s, pairs_per_step, ...).  Whole-job throughput:
    all pairs of all ranks over the slowest rank's time."""
    t_max = float(stats[:, 0].max())
    pairs_per_step = float(stats[:, 1].sum())
    return {
        "t_max": t_max,
        "pairs_per_step": pairs_per_step,
        "pairs_per_s": pairs_per_step * steps / t_max,
        "mpix_per_s": pixels_per_pair * pairs_per_step * steps / t_max / 1e6,
    }


def finalize():
    if dist.is_initialized():
        dist.destroy_process_group()
