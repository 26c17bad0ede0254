"""One process per GPU without an external launcher: `python bench.py --gpus N` starts its own N ranks.

Pairs shard embarrassingly (the reference's only split is parFor's row ranges, lib/gpc/filter.hpp:128-145),
so a "job" is N independent rank processes plus the RCCL barriers that time them.  This module is the part of that which
runs in the LAUNCHING process: it must never initialise a GPU (no HIP call, no torch import here) -- it only counts the
devices in a throw-away child, exports what torch.distributed.run would export, starts the ranks, relays rank 0's stdout
and stops the others by exact PID when one fails.
"""
import os
import signal
import socket
import subprocess
import sys
import time


def visible_devices():
    """HIP devices a child of this process would see, counted in a child so that this process stays GPU-free."""
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                             capture_output=True, text=True, timeout=600).stdout.strip().splitlines()
        return int(out[-1])
    except Exception:
        return 0


def free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def rank_env(n, rank, port, base=None):
    """The environment of rank `rank` of an n-rank single-node job (what torch.distributed.run exports)."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    return env


def launch_local_ranks(n, cmd, backend="nccl", count_devices=visible_devices, poll_s=0.05, log=sys.stderr):
    """Starts `cmd` (argv list) n times as ranks 0 .. n-1 of one node and waits for all of them.

    * fewer than n devices visible: refuses (returns 2) unless backend == "gloo" (rehearsal: the ranks share the devices
      there are) -- never a silent run with fewer ranks than asked for;
    * rank 0 inherits this process's stdout (its ONE JSON line is the job's), the other ranks' stdout goes to stderr;
    * returns 0 when every rank returned 0, else the first failing rank's status; a failing rank takes the others down
      (terminate() on the exact processes started here).
    """
    ndev = count_devices()
    if ndev < 1:
        print("launch: no HIP device visible (there is no CPU fallback)", file=log)
        return 2
    if ndev < n and backend != "gloo":
        print("launch: %d ranks asked for but only %d HIP device(s) visible: refusing to run fewer ranks than asked for "
              "(GPC_DIST_BACKEND=gloo rehearses the N-rank line with the ranks sharing the devices there are)" % (n, ndev), file=log)
        return 2
    port = os.environ.get("MASTER_PORT") or free_port()
    # Every rank in a session (process group) of its own: a rank's children (bench.py's host-to-host child) are reached by
    # the group's signal too.  SIGTERM / SIGINT to THIS process (a driver's `timeout`) are forwarded to the groups, so no
    # rank is left holding a GPU; a rank that ignores SIGTERM is killed after `grace_s`.  Exact process groups started
    # here, never a pattern.
    procs = [subprocess.Popen(list(cmd), env=rank_env(n, r, port), stdout=None if r == 0 else log, start_new_session=True)
             for r in range(n)]
    grace_s = float(os.environ.get("GPC_LAUNCH_GRACE_S", "10"))

    def signal_group(p, sig):
        try:
            os.killpg(p.pid, sig)       # (start_new_session: the rank's pid is its group's id)
        except (ProcessLookupError, PermissionError):
            pass

    def stop_all(sig=signal.SIGTERM):
        for p in procs:
            if p.poll() is None:
                signal_group(p, sig)

    got = []

    def on_signal(signum, frame):
        got.append(signum)
        stop_all(signal.SIGTERM)
    old = {}
    for sg in (signal.SIGTERM, signal.SIGINT):
        try:
            old[sg] = signal.signal(sg, on_signal)
        except ValueError:              # not the main thread (tests): no forwarding, the finally below still cleans up
            pass
    rc, alive, deadline = 0, set(range(n)), None
    try:
        while alive:
            for r in sorted(alive):
                st = procs[r].poll()
                if st is None:
                    continue
                alive.discard(r)
                if st != 0 and rc == 0:
                    rc = st if st > 0 else 128 - st
                    print("launch: rank %d exited with status %d; stopping the other ranks" % (r, st), file=log)
                    for q in alive:
                        signal_group(procs[q], signal.SIGTERM)
                    deadline = time.monotonic() + grace_s
            if got and deadline is None:
                deadline = time.monotonic() + grace_s
            if deadline is not None and time.monotonic() > deadline:
                for q in alive:
                    print("launch: rank %d did not stop within %.0f s: killed" % (q, grace_s), file=log)
                    signal_group(procs[q], signal.SIGKILL)
                deadline = time.monotonic() + 3600.0
            time.sleep(poll_s)
    finally:
        for p in procs:                 # whatever ends this function, no rank outlives it
            if p.poll() is None:
                signal_group(p, signal.SIGKILL)
        for sg, h in old.items():
            signal.signal(sg, h)
    if got and rc == 0:
        rc = 128 + got[0]
    return rc
