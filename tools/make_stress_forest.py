#!/usr/bin/env python3
"""Writes forests/stress16x20Forest.txt (BASELINE.json config 5): 16 ferns x 20 tests, offsets uniform in
[-13, 13], tau in [-10, 10], from a fixed LCG.  readForest keeps only the first 32 tests
(reference inference.hpp:425-432) and reports 288 discarded ones."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lcg(seed):
    state = seed
    while True:
        state = (state * 1664525 + 1013904223) & 0xFFFFFFFF
        yield state >> 8


def main():
    g = lcg(20181)
    lines = ["16 "]
    for fern in range(16):
        lines.append("%d l 20" % fern)
        for t in range(20):
            v = [next(g) % 27 - 13 for _ in range(4)]
            tau = next(g) % 21 - 10
            lines.append("%d %d %d %d %d %d" % (t, v[0], v[1], v[2], v[3], tau))
    with open(os.path.join(ROOT, "forests", "stress16x20Forest.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
