"""CPU experiment (round 3): probe lengths of an ORDER-PRESERVING open-addressing table for the row join.

slot = monotone map of the code into [0, R); ordered linear probing (smaller key first), no wrap-around.
Prints, per forest / image family: mean displacement of the left keys, the mean over 64-record groups of the
group's maximum (what a wave's probe loop costs), and the worst cluster.  Compared with the multiplicative hash.
"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle.pyoracle import Oracle
from opengpc_amd.synth import synth_pair

orc = Oracle(fast=True)

def table_positions(homes):
    """homes sorted by key (monotone => non-decreasing).  Position of each distinct key = max(home, prev+1)."""
    pos = np.empty_like(homes)
    p = -1
    for i, h in enumerate(homes):
        p = max(h, p + 1)
        pos[i] = p
    return pos

def stats(codes_rows, R, kind, bits):
    disp_all = []; wmax = []; worst = 0; over = 0
    for row in codes_rows:
        k = np.unique(row)
        if len(k) == 0: continue
        if kind == "mono":
            homes = ((k.astype(np.uint64) << np.uint64(32 - bits)) * np.uint64(R) >> np.uint64(32)).astype(np.int64)
            pos = table_positions(homes)
            d = pos - homes
            over = max(over, pos[-1] - (R - 1))
        else:
            S = 2048
            h = ((k.astype(np.uint64) + 1) * np.uint64(0x9E3779B1) & np.uint64(0xFFFFFFFF)) >> np.uint64(32 - 11)
            # ordered probing with a random hash: simulate by inserting in descending key order (larger first)
            tab = -np.ones(S, np.int64); d = np.zeros(len(k), np.int64)
            for i in np.argsort(-k.astype(np.int64)):
                p = int(h[i]); c = 0
                while tab[p] >= 0:
                    p = (p + 1) & (S - 1); c += 1
                tab[p] = k[i]; d[i] = c
        disp_all.append(d)
        worst = max(worst, int(d.max()))
        # records of a row in pixel order are spread over lanes; approximate a wave's cost by random groups of 64
        rng = np.random.default_rng(1)
        dd = rng.permutation(d)
        for g in range(0, len(dd), 64):
            wmax.append(dd[g:g + 64].max())
    d = np.concatenate(disp_all)
    return d.mean(), float(np.mean(wmax)), worst, over

def run(name, W, H, s, D, forest_path, thr=5, R=1546):
    L, Rr = synth_pair(W, H, s, D)
    rc, f = orc.read_forest(forest_path, W, H)
    sm, gr, mask = orc.preprocess(L, thr)
    codes = orc.hash(sm, gr, f)
    cand = np.zeros(W * H, bool); cand[mask] = True; cand = cand.reshape(H, W)
    allc = codes[cand]
    bits = int(allc.max()).bit_length() if len(allc) else 1
    bits = int(np.bitwise_or.reduce(allc)).bit_length()
    rows = [codes[y][cand[y]] for y in range(13, H - 13, 7)]
    n = np.mean([len(r) for r in rows])
    for kind in ("mono", "mult"):
        m, wm, worst, over = stats(rows, R, kind, bits)
        print(f"{name:28s} {kind}: recs/row {n:6.1f} bits {bits} mean disp {m:6.2f}  wave-max mean {wm:6.2f}  worst {worst}  overflow past R: {over}")

if __name__ == "__main__":
    here = os.path.join(os.path.dirname(__file__), "..", "..", "forests")
    run("1024x436 Zero s0", 1024, 436, 0, 24, os.path.join(here, "defaultZeroForest.txt"))
    run("1024x436 Zero s7", 1024, 436, 7, 15, os.path.join(here, "defaultZeroForest.txt"))
    run("1024x436 Tau s0", 1024, 436, 0, 24, os.path.join(here, "defaultTauForest.txt"))
    run("1024x436 stress s2", 1024, 436, 2, 64, os.path.join(here, "stress16x20Forest.txt"))
