#!/usr/bin/env python3
"""What the host link of this box gives: page-locked H2D, D2H and both at once (two streams), 256 MiB each way.
Context for pcie_inclusive: a 256-pair batch moves 229 MB in and (packed) 209 MB out."""
import json
import time

import torch


def main():
    dev = torch.device("cuda", 0)
    nb = 256 << 20
    h_in = torch.empty(nb, dtype=torch.uint8).pin_memory()
    h_out = torch.empty(nb, dtype=torch.uint8).pin_memory()
    d_a = torch.empty(nb, dtype=torch.uint8, device=dev)
    d_b = torch.empty(nb, dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    res = {}

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / reps

    def h2d():
        with torch.cuda.stream(s1):
            d_a.copy_(h_in, non_blocking=True)

    def d2h():
        with torch.cuda.stream(s2):
            h_out.copy_(d_b, non_blocking=True)

    def both():
        h2d()
        d2h()
    piece = 7 << 20

    def h2d_pieces():
        with torch.cuda.stream(s1):
            for o in range(0, nb - piece + 1, piece):
                d_a[o:o + piece].copy_(h_in[o:o + piece], non_blocking=True)

    def d2h_pieces():
        with torch.cuda.stream(s2):
            for o in range(0, nb - piece + 1, piece):
                h_out[o:o + piece].copy_(d_b[o:o + piece], non_blocking=True)

    def both_pieces():
        h2d_pieces()
        d2h_pieces()
    npc = (nb // piece) * piece
    res["h2d_7MiB_pieces_GBs"] = round(npc / timed(h2d_pieces) / 1e9, 1)
    res["d2h_7MiB_pieces_GBs"] = round(npc / timed(d2h_pieces) / 1e9, 1)
    res["both_ways_7MiB_pieces_each_GBs"] = round(npc / timed(both_pieces) / 1e9, 1)
    res["h2d_GBs"] = round(nb / timed(h2d) / 1e9, 1)
    res["d2h_GBs"] = round(nb / timed(d2h) / 1e9, 1)
    res["both_ways_each_GBs"] = round(nb / timed(both) / 1e9, 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
