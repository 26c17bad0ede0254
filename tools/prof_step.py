#!/usr/bin/env python3
"""Runs a few batched steps of the hot path without torch (fast start-up) -- the target of
rocprofv3 kernel-trace / PMC runs.  Usage: python3 tools/prof_step.py [steps] [batch] [W] [H] [forest]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import opengpc_amd as g  # noqa: E402
from opengpc_amd.synth import synth_batch  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    H = int(sys.argv[4]) if len(sys.argv) > 4 else 436
    forest = sys.argv[5] if len(sys.argv) > 5 else os.path.join(ROOT, "forests", "defaultZeroForest.txt")
    epipolar = os.environ.get("GPC_PROF_GLOBAL") is None
    ctx = g.Context(0)
    ctx.load_forest(forest, W, H)
    L, R = synth_batch(W, H, list(range(B)))
    s = g.Settings.sparsematch()
    s.epipolar_mode = int(epipolar)
    s.use_hashtable = int(os.environ.get("GPC_PROF_HASHTABLE") is not None)
    cap = (W - 26) * (H - 26)
    for _ in range(steps):
        out, counts, ncand, st = ctx.match_batch(L, R, s, cap)
    print("steps", steps, "pairs", B, "supports/pair", counts.mean(), "cand/pair", ncand.sum(1).mean())
    ctx.close()


if __name__ == "__main__":
    main()
