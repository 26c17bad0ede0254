#!/usr/bin/env python3
"""One 1024x436 pair host to host (the reference's literal timed region) under the library's tuning knobs: which of the
host path's choices the 0.14 ms are made of.  usage: python tools/single_pair_probe.py [W H forest]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import opengpc_amd as g  # noqa: E402
from pcie_inclusive import single_pair  # noqa: E402


def main():
    W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) >= 3 else (1024, 436)
    forest = sys.argv[3] if len(sys.argv) >= 4 else os.path.join(ROOT, "forests", "defaultZeroForest.txt")
    variants = [("default", {}), ("fuse_always", {"GPC_HIP_FUSE_ALWAYS": "1"}), ("no_direct", {"GPC_HIP_DIRECT_MAX": "0"}),
                ("default_again", {})]
    extra = os.environ.get("PROBE_VARIANTS")
    if extra:  # name:K=V,K=V;name2:...
        for spec in extra.split(";"):
            name, kv = spec.split(":", 1)
            variants.append((name, dict(p.split("=", 1) for p in kv.split(",") if p)))
    for name, env in variants:
        for k, v in env.items():
            os.environ[k] = v
        ctx = g.Context(0)  # the knobs are read when a context is made
        ctx.load_forest(forest, W, H)
        r = single_pair(ctx, W, H, g.Settings.sparsematch())
        ctx.close()
        for k in env:
            del os.environ[k]
        print(name, json.dumps({k: r[k] for k in ("pinned", "pageable")}), flush=True)


if __name__ == "__main__":
    main()
