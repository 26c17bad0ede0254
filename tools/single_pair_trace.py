#!/usr/bin/env python3
"""Twenty single-pair host-to-host calls (page-locked buffers) with the wall-clock start / end of each printed in ns
(CLOCK_MONOTONIC ... rocprofv3's timestamps are on the same clock): run under
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -- python3 tools/single_pair_trace.py
and read the timeline of one call with tools/single_pair_timeline.py."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import opengpc_amd as g  # noqa: E402
from opengpc_amd.synth import synth_batch  # noqa: E402

W, H = 1024, 436
ctx = g.Context(0)
ctx.load_forest(os.path.join(ROOT, "forests", "defaultZeroForest.txt"), W, H)
s = g.Settings.sparsematch()
L, R = synth_batch(W, H, [0])
cap = 300000
Lb, Rb = ctx.pinned_empty(L.shape, np.uint8), ctx.pinned_empty(R.shape, np.uint8)
Lb[:] = L
Rb[:] = R
out = ctx.pinned_empty((1, cap), g.SUPPORT_DTYPE)
calls = []
for i in range(20):
    t0 = time.clock_gettime_ns(time.CLOCK_MONOTONIC)
    ctx.match_batch(Lb, Rb, s, cap, out=out)
    t1 = time.clock_gettime_ns(time.CLOCK_MONOTONIC)
    calls.append((t0, t1))
    time.sleep(0.002)
print(json.dumps({"calls_ns": calls}))
ctx.close()
