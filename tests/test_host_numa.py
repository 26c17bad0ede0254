"""gpc_hip_match_batch from a thread on the other socket: the chunk pipeline hops to a thread bound to the GPU's NUMA node
(gpc_hip.hip: on_gpu_node) -- same results, the caller's affinity untouched; the stage clock and the worker CPUs are
reported (gpc_hip_batch_stages, gpc_hip_host_worker_cpus).  Reference region: samples/sparsematch.cpp:45-52."""
import os

import numpy as np
import pytest

from opengpc_amd.hostinfo import cpu_nodes, current_cpu, pages_nodes, stage_summary

pytestmark = pytest.mark.gpu


def test_stages_and_worker_cpus_are_reported(forest_paths):
    import opengpc_amd as g
    from opengpc_amd.synth import synth_batch
    W, H, B = 1024, 436, 32
    c = g.Context(0)
    try:
        c.load_forest(forest_paths["zero"], W, H)
        L, R = synth_batch(W, H, list(range(B)))
        o, counts, ncand, st = c.match_batch(L, R, g.Settings.sparsematch(), 300000)
        s = c.batch_stages()
        assert st == 0 and 0.0 < s[0] <= s[1] <= s[2] <= s[3] < 1000.0, s
        cpus = c.worker_cpus()
        assert len(cpus) == c.L.gpc_hip_host_threads(c.h) >= 2 and all(v >= 0 for v in cpus)
        node = c.L.gpc_hip_host_numa_node(c.h)
        nodes = cpu_nodes()
        if node >= 0 and nodes:          # workers are bound to the GPU's node
            assert {nodes.get(v) for v in cpus} == {node}
        assert stage_summary([s, s])["delivered"] == round(s[3], 3)
        assert pages_nodes(np.ones(1 << 22, np.uint8))     # (the query works in this process)
    finally:
        c.close()


def test_caller_on_the_other_socket_hops_to_the_gpu_node(forest_paths):
    import opengpc_amd as g
    from opengpc_amd.synth import synth_batch
    W, H, B = 1024, 436, 16
    nodes = cpu_nodes()
    mine = os.sched_getaffinity(0)
    c = g.Context(0)
    try:
        node = c.L.gpc_hip_host_numa_node(c.h)
        far = sorted(v for v in mine if nodes.get(v, node) != node)
        if node < 0 or not far:
            pytest.skip("one NUMA node (or the GPU's node is unknown): nothing to hop from")
        c.load_forest(forest_paths["zero"], W, H)
        L, R = synth_batch(W, H, list(range(B)))
        s = g.Settings.sparsematch()
        want = c.match_batch(L, R, s, 300000)
        before = c.L.gpc_hip_fed_calls(c.h)
        os.sched_setaffinity(0, {far[len(far) // 2]})      # this thread only
        try:
            assert nodes[current_cpu()] != node
            got = c.match_batch(L, R, s, 300000)
            assert os.sched_getaffinity(0) == {far[len(far) // 2]}      # the caller stays where it was
        finally:
            os.sched_setaffinity(0, mine)
        assert c.L.gpc_hip_fed_calls(c.h) == before + 1
        assert got[3] == 0 and np.array_equal(got[1], want[1])
        for j in range(B):
            assert np.array_equal(got[0][j, : got[1][j]], want[0][j, : want[1][j]])
    finally:
        c.close()
