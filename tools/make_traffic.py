#!/usr/bin/env python3
"""Turns a tools/pmc_collect.sh output directory into profiles/traffic.json (HBM bytes per launch
per kernel, gfx950 corrections per MI355X_MICROARCH.md: FETCH_SIZE counts half of coalesced
streaming reads -> doubled; WRITE_SIZE exact) and a PMC summary json."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src, tag, W, H, B = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for grp in ("sq_busy", "sq_insts", "sq_lds", "fetch", "write", "grbm"):
        for f in glob.glob(os.path.join(src, grp, "*", "*_counter_collection.csv")):
            for row in csv.DictReader(open(f)):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                if "gpc::" in name:
                    agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    summ = {k: {c: round(sum(v) / len(v)) for c, v in d.items()} for k, d in agg.items()}
    pmc_dir = os.path.join(ROOT, "profiles", tag.split("_")[0] + "_pmc")   # profiles/r03_pmc for tag r03_a
    os.makedirs(pmc_dir, exist_ok=True)
    json.dump({"_comment": "rocprofv3 --pmc per-launch averages, one counter group per run (tools/pmc_collect.sh); "
                           "workload %d pairs %dx%d; FETCH_SIZE / WRITE_SIZE in KiB" % (B, W, H), "kernels": summ},
              open(os.path.join(pmc_dir, "%s_pmc_summary.json" % tag), "w"), indent=1)
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    traffic["_comment"] = ("HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 from rocprofv3 PMC passes "
                           "(gfx950: FETCH_SIZE reports half of coalesced streaming reads); keys kernel@WxHxpairs")
    alias = {"k_row_join": "k_row_join", "k_row_bucket": "k_row_join", "k_row_join_fused": "k_row_join"}
    for k, d in summ.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            base = k.replace("gpc::", "").split("<")[0]
            base = alias.get(base, base)
            traffic["%s@%dx%dx%d" % (base, W, H, B)] = int((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024)
    # how busy the vector ALUs were: SQ_ACTIVE_INST_VALU counts quad-cycles (one 4-clock issue slot per vector instruction),
    # GRBM_GUI_ACTIVE the launch's clocks summed over the 8 XCDs; 256 CUs x 4 SIMDs
    for k, d in summ.items():
        if d.get("SQ_ACTIVE_INST_VALU") and d.get("GRBM_GUI_ACTIVE"):
            base = alias.get(k.replace("gpc::", "").split("<")[0], k.replace("gpc::", "").split("<")[0])
            slots = d["GRBM_GUI_ACTIVE"] / 8.0 / 4.0 * 1024.0
            traffic["valu_issue@%s@%dx%dx%d" % (base, W, H, B)] = {
                "vector_instructions": int(d.get("SQ_INSTS_VALU", 0)), "valu_quad_cycles": int(d["SQ_ACTIVE_INST_VALU"]),
                "quad_cycle_slots": int(slots), "frac": round(d["SQ_ACTIVE_INST_VALU"] / slots, 4)}
    json.dump(traffic, open(tpath, "w"), indent=1)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
