#!/usr/bin/env python3
"""Times the fern-training scoring loop on the GPU for the reference's own settings
(samples/train.cpp: depth 5, 10 resamples, zero / tau optimizer) and for wider searches, with the
candidate-evaluation kernel's HIP-event time and its algorithmic HBM rate; the CPU oracle is timed
on a bounded sample beside it.  Output: gpurun_out/train_bench.json."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import opengpc_amd as g  # noqa: E402
from oracle.pyoracle import Oracle, SPLIT_DTYPE  # noqa: E402  (CPU leg only)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    rng = np.random.default_rng(7)
    t = rng.integers(0, 256, (n, 3, 729), dtype=np.uint8)
    t[:, 1] = t[:, 0] ^ rng.integers(0, 4, (n, 729), dtype=np.uint8)  # pos ~ ref
    ctx = g.Context(0)
    t0 = time.perf_counter()
    ts = ctx.train_set(t)
    upload_s = time.perf_counter() - t0
    depth = 5
    res = {"triplets": n, "bytes": int(t.nbytes), "upload_and_transpose_s": round(upload_s, 3), "runs": []}
    for name, nres, taulo, tauhi, only in (("train.cpp zero optimizer", 10, 0, 1, False),
                                           ("train.cpp tau optimizer", 10, -10, 10, False),
                                           ("zero, 1000 resamples", 1000, 0, 1, True),
                                           ("tau (-10..10), 200 resamples", 200, -10, 10, True)):
        cand = np.zeros(depth * nres, SPLIT_DTYPE)
        cand["i"] = rng.integers(0, 729, len(cand))
        cand["j"] = (cand["i"] + rng.integers(1, 729, len(cand))) % 729
        ts.train_fern(depth, cand, nres, taulo, tauhi, only, 0.5)  # warm-up
        ctx.enable_kernel_timing(True, only=["k_train_eval"])
        ctx.reset_kernel_timing()
        t0 = time.perf_counter()
        fp, st = ts.train_fern(depth, cand, nres, taulo, tauhi, only, 0.5)
        dt = time.perf_counter() - t0
        ms, launches = ctx.kernel_times()["k_train_eval"]
        ctx.enable_kernel_timing(False)
        alg = 7.0 * n * nres  # 6 patch bytes + 1 flag byte per triplet and candidate, per launch (= level)
        res["runs"].append({
            "config": name, "depth": depth, "resamples": nres, "taus": tauhi - taulo,
            "train_fern_ms": round(dt * 1e3, 3),
            "eval_kernel_ms_per_level": round(ms / launches, 4),
            "eval_alg_GBs": round(alg / (ms / launches * 1e-3) / 1e9, 1),
            "candidate_evaluations_per_s": round(depth * nres * (tauhi - taulo) * n / dt, 1),
            "last_level": {k: (float(st[-1][k]) if k in ("prec", "rec", "hmean") else int(st[-1][k]))
                           for k in ("prec", "rec", "hmean", "tp", "fp", "fn", "tot")},
        })
        print(json.dumps(res["runs"][-1]))
    # CPU leg: the oracle's evalSplit, one thread, on a bounded sample
    o = Oracle(fast=True)
    m = min(n, 200000)
    marks = np.zeros(m, np.uint8)
    p = np.zeros(5, SPLIT_DTYPE)
    p["i"] = rng.integers(0, 729, 5)
    p["j"] = (p["i"] + 1) % 729
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        o.eval_split(t[:m], marks, p, 4, 0.5)
    cpu = (time.perf_counter() - t0) / reps
    # Fern::train evaluates levels 0..L for every candidate of level L: 1+2+..+5 = 15 level-tests per resample and tau
    res["cpu_oracle"] = {"kind": "port", "cores": 1, "sample": "%d triplets, evalSplit over 5 levels, %d reps" % (m, reps),
                         "level_tests_per_s": round(5 * m / cpu, 1)}
    print(json.dumps(res["cpu_oracle"]))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "train_bench.json"), "w"), indent=1)
    ts.close()
    ctx.close()


if __name__ == "__main__":
    main()
