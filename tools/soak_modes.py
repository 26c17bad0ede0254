#!/usr/bin/env python3
"""One-off soak of the device-wide matcher modes (non-epipolar sort-matcher, hash-table matcher, both with their
partitioned and radix implementations decided by the data) at image sizes around the partition capacity:
random shapes 640..1600 x 200..640, the image kinds of tests/test_gpu_fuzz.py, against the oracle (-O3 build).
usage (GPU box): python tools/soak_modes.py [configurations] [first seed] [all]
"all" adds the default epipolar sort-matcher to the rotation, through the single-pair call and through gpc_hip_match_batch
(packed results over the link + host expansion) with the pair and its mirror."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import opengpc_amd as g  # noqa: E402
from oracle.pyoracle import Oracle, sparsematch_settings  # noqa: E402
from test_gpu_fuzz import draw_pair  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    modes = [(False, False), (True, True), (False, True)] + ([(True, False)] if len(sys.argv) > 3 and sys.argv[3] == "all" else [])
    oracle = Oracle(fast=True)
    ctx = g.Context(0)
    forests = {k: os.path.join(ROOT, "forests", "default%sForest.txt" % k.capitalize()) for k in ("zero", "tau")}
    t0 = time.time()
    done = 0
    for seed in range(s0, s0 + n):
        rng = np.random.default_rng(50000 + seed)
        W = 16 * int(rng.integers(40, 101))
        H = int(rng.integers(200, 641))
        fo = "tau" if seed % 2 else "zero"
        epi, ht = modes[seed % len(modes)]
        thr = int(rng.choice([0, 5, 5, 5, 40]))
        disp = int(rng.choice([64, 128, 4000]))
        vtol = int(rng.choice([0, 1, 3]))
        L, R = draw_pair(rng, W, H)
        rc, f = oracle.read_forest(forests[fo], W, H)
        ctx.load_forest(forests[fo], W, H)
        want, nl, nr = oracle.match_pair(L, R, f, sparsematch_settings(thr, disp, vtol, epi, ht))
        got, cnt, ncand, st = ctx.match_pair(L, R, g.Settings(thr, disp, vtol, epi, ht, 1))
        ok = st == 0 and (nl, nr) == tuple(ncand) and cnt == len(want) and np.array_equal(got, want.astype(got.dtype))
        if ok and epi and not ht:
            sset = g.Settings(thr, disp, vtol, epi, ht, 1)
            out, counts, nc, st2 = ctx.match_batch(np.stack([L, R, L]), np.stack([R, L, R]), sset, max(cnt, 1) * 2 + W * H)
            w2, l2, r2 = oracle.match_pair(R, L, f, sparsematch_settings(thr, disp, vtol, epi, ht))
            ok = (st2 == 0 and counts[0] == cnt and counts[2] == cnt and np.array_equal(out[0, :cnt], got)
                  and np.array_equal(out[2, :cnt], got) and counts[1] == len(w2)
                  and np.array_equal(out[1, :counts[1]], w2.astype(got.dtype)) and tuple(nc[1]) == (l2, r2))
        if not ok:
            print("MISMATCH seed", seed, W, H, fo, epi, ht, thr, disp, vtol, "cand", nl, nr, "n", cnt, len(want))
            sys.exit(1)
        done += 1
        if done % 20 == 0:
            print("%d ok (last: %dx%d %s epi=%d ht=%d cand %d+%d -> %d) %.0f s" % (done, W, H, fo, epi, ht, nl, nr, cnt, time.time() - t0), flush=True)
    print("soak_modes: %d configurations identical to the oracle in %.0f s" % (done, time.time() - t0))
    ctx.close()


if __name__ == "__main__":
    main()
