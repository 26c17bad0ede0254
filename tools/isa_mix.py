#!/usr/bin/env python3
"""Static instruction mix of kernels in a hipcc -S dump.  usage: tools/isa_mix.py gpc.s 'mangled-name-regex'"""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
pat = re.compile(sys.argv[2])
cur = None; out = {}
for ln in lines:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        cur = m.group(1) if pat.search(m.group(1)) else None
        if cur: out[cur] = collections.Counter()
        continue
    if cur is None: continue
    if ln.startswith(".Lfunc_end"): cur = None; continue
    m = re.match(r"\s+([a-z_0-9]+)(\s|$)", ln)
    if not m: continue
    op = m.group(1); c = out[cur]
    c["total"] += 1
    for pre in ("s_", "v_", "ds_", "global_", "scratch_", "buffer_"):
        if op.startswith(pre): c[pre] += 1
    if op in ("v_readlane_b32", "v_writelane_b32"): c["lane_spill"] += 1
    if op == "s_barrier": c["barrier"] += 1
    if op.startswith("s_waitcnt"): c["waitcnt"] += 1
for k, c in out.items(): print(k[:60], dict(c))
