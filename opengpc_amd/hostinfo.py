"""Diagnostics of the host side of a host-to-host call (bench.py, tools/pcie_inclusive.py): which NUMA node CPUs and pages
belong to, and the medians of gpc_hip_batch_stages.  Plumbing for measurement only."""
import os

import numpy as np


def current_cpu():
    """the CPU the calling thread runs on (libc sched_getcpu), -1 if unknown"""
    import ctypes as C
    try:
        return int(C.CDLL(None).sched_getcpu())
    except Exception:
        return -1


def cpu_nodes():
    """cpu -> NUMA node, from sysfs"""
    m = {}
    base = "/sys/devices/system/node"
    try:
        for d in os.listdir(base):
            if not d.startswith("node") or not d[4:].isdigit():
                continue
            with open(os.path.join(base, d, "cpulist")) as fh:
                for part in fh.read().strip().split(","):
                    if not part:
                        continue
                    a, _, b = part.partition("-")
                    for c in range(int(a), int(b or a) + 1):
                        m[c] = int(d[4:])
    except OSError:
        pass
    return m


def pages_nodes(arr, samples=96):
    """NUMA node of `samples` pages spread over a numpy array (move_pages with no target nodes only asks): {node: pages};
    -2 = not present / not known."""
    import ctypes as C
    try:
        libc = C.CDLL(None, use_errno=True)
        addr = arr.ctypes.data
        nbytes = arr.nbytes
        if nbytes < 4096:
            return {}
        n = min(samples, nbytes // 4096)
        pages = (C.c_void_p * n)(*[((addr + (i * (nbytes - 4096)) // max(n - 1, 1)) & ~4095) for i in range(n)])
        status = (C.c_int * n)(*([-2] * n))
        rc = libc.syscall(279, 0, C.c_ulong(n), pages, None, status, 0)   # SYS_move_pages (x86-64)
        if rc != 0:
            return {"error": C.get_errno()}
        out = {}
        for v in status:
            out[int(v)] = out.get(int(v), 0) + 1
        return {str(k): v for k, v in sorted(out.items())}
    except Exception as e:      # diagnostics only
        return {"error": str(e)}


def stage_summary(rows):
    """median of every stage over the calls: ms since entry when the host saw the last upload complete, the last kernels
    done, the last packed chunk landed, the delivery done"""
    if not rows:
        return None
    a = np.sort(np.asarray(rows, np.float64), axis=0)
    med = a[len(a) // 2]
    return {"upload_done": round(float(med[0]), 3), "kernels_done": round(float(med[1]), 3),
            "last_chunk_landed": round(float(med[2]), 3), "delivered": round(float(med[3]), 3)}
