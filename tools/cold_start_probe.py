#!/usr/bin/env python3
"""Where the first call of a process goes (the reference's sample is a run-once CLI): wall time of context creation, forest
upload, the first and the second single-pair call.  usage: python tools/cold_start_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

t0 = time.perf_counter()
import opengpc_amd as g  # noqa: E402  (loads libgpc_hip.so and with it libamdhip64)
from opengpc_amd.synth import synth_batch  # noqa: E402
t1 = time.perf_counter()
W, H = 1024, 436
L, R = synth_batch(W, H, [0])
t2 = time.perf_counter()
ctx = g.Context(0)
t3 = time.perf_counter()
ctx.load_forest(os.path.join(ROOT, "forests", "defaultZeroForest.txt"), W, H)
t4 = time.perf_counter()
s = g.Settings.sparsematch()
o, c, n, st = ctx.match_batch(L, R, s, 300000)
t5 = time.perf_counter()
o, c, n, st = ctx.match_batch(L, R, s, 300000)
t6 = time.perf_counter()
print("import (dlopen) %.1f ms | gpc_hip_create %.1f ms | forest %.1f ms | first call %.1f ms | second call %.2f ms" %
      ((t1 - t0) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t6 - t5) * 1e3))
ctx.close()
