#!/usr/bin/env python3
"""Timeline of the last single-pair call of tools/single_pair_trace.py from a rocprofv3 csv directory.
usage: python tools/single_pair_timeline.py <rocprof dir> <stdout of single_pair_trace.py>"""
import csv
import glob
import json
import os
import sys

d, log = sys.argv[1], sys.argv[2]
calls = None
for line in open(log):
    if line.startswith("{\"calls_ns\""):
        calls = json.loads(line)["calls_ns"]
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")))
ev.sort()
for ci in (-3, -2, -1):
    t0, t1 = calls[ci]
    print("call %d: %.1f us wall" % (ci, (t1 - t0) / 1e3))
    for a, b, n in ev:
        if a >= t0 - 50000 and b <= t1 + 50000:
            print("  %8.1f .. %8.1f  (%6.1f us)  %s" % ((a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, n))
