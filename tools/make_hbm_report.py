#!/usr/bin/env python3
"""Merges one tools/config_report.sh directory into an HBM report: per kernel the rocprofv3 average duration, the HBM
bytes of the PMC passes (gfx950: FETCH_SIZE counts half of coalesced streaming reads -> doubled, MI355X_MICROARCH.md; both
counters in KiB), GB/s and the fraction of the 8 TB/s peak, beside the bytes the kernel must move given its formats.
usage: make_hbm_report.py <dir> <steps> <batch> <W> <H> <forest> [s D]"""
import collections
import csv
import json
import os
import sys

PEAK = 8000.0


def main():
    d, steps, B, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    forest = os.path.basename(sys.argv[6]) if len(sys.argv) > 6 else "defaultZeroForest.txt"
    stats = {}
    p = os.path.join(d, "kernel_stats.csv")
    if os.path.exists(p):
        for r in csv.DictReader(open(p)):
            if "gpc::" in r["Name"]:
                name = r["Name"].split("(")[0].replace("void ", "")
                stats[name] = {"launches": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2)}
    pm = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        p = os.path.join(d, c + ".csv")
        if os.path.exists(p):
            for r in csv.DictReader(open(p)):
                if "gpc::" in r["Kernel_Name"]:
                    pm[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    supports = cand = None
    try:
        for line in open(os.path.join(d, "trace.log")):
            if line.startswith("steps"):
                t = line.split()
                supports, cand = float(t[t.index("supports/pair") + 1]), float(t[t.index("cand/pair") + 1])
    except OSError:
        pass
    rows = H - 26
    M = (supports or 0.0) * B
    fused = not any("k_gather_rows" in k for k in stats)
    g = 0.125   # the gradient image between k_preprocess and k_hash is one bit per pixel in the batched pipelines
    must = {"k_preprocess": 2.0 * (2.0 + g) * W * H * B, "k_hash": 2.0 * (5.0 + g) * W * H * B,
            "k_row_join": 8.0 * W * rows * B + (12.0 if fused else 4.0) * M, "k_gather_rows": 16.0 * M}
    must["k_row_join_fused"] = must["k_row_join"]
    out = {"_comment": "rocprofv3 --kernel-trace --stats averages and --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) of "
                       "tools/prof_step.py %s; traffic = (2 * FETCH_SIZE + WRITE_SIZE) KiB per launch (gfx950 correction); "
                       "must_move = bytes the kernel has to read + write given its input / output formats (DESIGN.md 4)"
                       % " ".join(sys.argv[2:]),
           "config": {"pairs_per_launch": B, "width": W, "height": H, "forest": forest, "steps": steps,
                      "candidates_per_pair": cand, "supports_per_pair": supports},
           "kernels": {}}
    tot_us = tot_b = tot_must = 0.0
    for k, s in sorted(stats.items(), key=lambda kv: -kv[1]["avg_us"]):
        e = dict(s)
        c = pm.get(k, {})
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            fe, wr = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]), sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
            tb = (2 * fe + wr) * 1024
            e.update({"hbm_read_MB": round(2 * fe * 1024 / 1e6, 2), "hbm_write_MB": round(wr * 1024 / 1e6, 2),
                      "traffic_GBs": round(tb / (s["avg_us"] * 1e-6) / 1e9, 1), "traffic_frac_of_8TBs": round(tb / (s["avg_us"] * 1e-6) / 1e9 / PEAK, 4)})
            tot_b += tb
        base = k.replace("gpc::", "").split("<")[0]
        if base in must:
            e["must_move_MB"] = round(must[base] / 1e6, 2)
            e["must_move_frac_of_8TBs"] = round(must[base] / (s["avg_us"] * 1e-6) / 1e9 / PEAK, 4)
            tot_must += must[base]
        tot_us += s["avg_us"]
        out["kernels"][k] = e
    if tot_us:
        out["whole_launch_sequence"] = {"sum_kernel_us": round(tot_us, 2), "traffic_MB": round(tot_b / 1e6, 2),
                                        "traffic_frac_of_8TBs": round(tot_b / (tot_us * 1e-6) / 1e9 / PEAK, 4),
                                        "must_move_MB": round(tot_must / 1e6, 2),
                                        "must_move_frac_of_8TBs": round(tot_must / (tot_us * 1e-6) / 1e9 / PEAK, 4),
                                        "Mpix_per_s": round(2.0 * W * H * B / tot_us, 1)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
