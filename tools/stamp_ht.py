#!/usr/bin/env python3
"""Phase shares of k_ht_join (hash-table matcher) for a given image size, from a -DGPC_STAMPS build
(tools/variant_local.sh "stamps:-DGPC_STAMPS" -> tools/variants/libgpc_stamps.so).
usage: python tools/stamp_ht.py W H pairs zero|tau"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import opengpc_amd.capi as capi  # noqa: E402
capi.LIB_PATH = os.path.join(ROOT, "tools", "variants", "libgpc_stamps.so")
import opengpc_amd as g  # noqa: E402
from opengpc_amd.synth import synth_batch  # noqa: E402

W, H, B, fo = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
ctx = g.Context(0)
ctx.load_forest(os.path.join(ROOT, "forests", "default%sForest.txt" % fo.capitalize()), W, H)
L, R = synth_batch(W, H, list(range(B)))
cap = (W - 26) * (H - 26)
hs = g.Settings(5, 128, 1, True, True, 1)
buf = (C.c_ulonglong * 16)()
HJ = ["bin bounds (scalar loads)", "records arrive", "buckets + counts", "scan", "placed", "10-cap", "ranks",
      "list order + links", "walk + scan", "output"]
o, counts, ncand, st = ctx.match_batch(L, R, hs, cap)
ctx.L.gpc_hip_debug_htjoin_stamps.argtypes = [C.c_void_p, C.c_void_p]
ctx.L.gpc_hip_debug_htjoin_stamps(ctx.h, buf)
o, counts, ncand, st = ctx.match_batch(L, R, hs, cap)
ctx.L.gpc_hip_debug_htjoin_stamps(ctx.h, buf)
tot = sum(buf[i] for i in range(len(HJ)))
print("k_ht_join phase shares, %dx%d x%d %s (supports %d, candidates %d):" % (W, H, B, fo, int(counts[0]), int(ncand[0].sum())))
for i, name in enumerate(HJ):
    print("  %-28s %5.1f %%" % (name, 100.0 * buf[i] / max(tot, 1)))
ctx.close()
