#!/usr/bin/env python3
"""The drop-in's own printout on the MI355X box (BASELINE.md's one quantitative reference row is this line of
samples/sparsematch.cpp:53-57: tPreprocess / tMatch).  Runs
  * the REFERENCE's unchanged samples/sparsematch.cpp built against include/ (oracle/_ref/ref_sparsematch_on_amd_headers,
    when that build travelled with the snapshot): one-shot processes, each a cold first call;
  * this repository's samples/sparsematch with --repeat 20, the three-call API (preprocessImage x2 + rectifiedMatch: smooth,
    grad and mask go D2H and H2D again, as the reference's by-value PreprocessedImage demands) and --fused (Forest::matchPair)
on 1024x436 with defaultZeroForest and 1920x1080 with defaultTauForest; prints every line and a first-call / median summary.
usage: python tools/dropin_printout.py > profiles/r03_dropin_printout.txt"""
import os
import re
import statistics
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from PIL import Image  # noqa: E402

from opengpc_amd.synth import synth_pair  # noqa: E402

LINE = re.compile(r"tPreprocess: ([\d.e+-]+) ms, #candidatesL:(\d+), #candidatesR:(\d+), tMatch: ([\d.e+-]+) ms, num matches:(\d+)")
REF = os.path.join(ROOT, "oracle", "_ref", "ref_sparsematch_on_amd_headers")
OURS = os.path.join(ROOT, "samples", "sparsematch")


def run(args, cwd):
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "opengpc_amd") + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run(args, cwd=cwd, env=env, capture_output=True, text=True, timeout=300).stdout
    return [(float(m.group(1)), float(m.group(4)), int(m.group(2)), int(m.group(3)), int(m.group(5))) for m in LINE.finditer(out)], out


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "samples")])
    cases = [("1024x436 defaultZeroForest (BASELINE configs[0/1])", 1024, 436, 0, 24, "defaultZeroForest.txt"),
             ("1920x1080 defaultTauForest (BASELINE configs[2])", 1920, 1080, 1, 40, "defaultTauForest.txt")]
    with tempfile.TemporaryDirectory() as td:
        for name, W, H, s, D, forest in cases:
            L, R = synth_pair(W, H, s, D)
            lp, rp = os.path.join(td, "l_%d.png" % W), os.path.join(td, "r_%d.png" % W)
            Image.fromarray(L, "L").save(lp)
            Image.fromarray(R, "L").save(rp)
            fp = os.path.join(ROOT, "forests", forest)
            print("=" * 100)
            print(name)
            if os.path.exists(REF):
                print("-- the reference's unchanged sample on these headers (one process per line = first call each):")
                rows = []
                for _ in range(5):
                    r, out = run([REF, fp, lp, rp], td)
                    rows += r
                    for ln in out.splitlines():
                        if "tPreprocess" in ln:
                            print("   " + ln)
                if rows:
                    print("   => first call: tPreprocess median %.3f ms, tMatch median %.3f ms, sum %.3f ms" % (
                        statistics.median(r[0] for r in rows), statistics.median(r[1] for r in rows),
                        statistics.median(r[0] + r[1] for r in rows)))
            for label, extra in (("three-call API (preprocessImage x2 + rectifiedMatch)", []), ("--fused (Forest::matchPair)", ["--fused"])):
                r, out = run([OURS, fp, lp, rp, "--repeat", "20"] + extra, td)
                print("-- samples/sparsematch --repeat 20, %s:" % label)
                for ln in out.splitlines():
                    if "tPreprocess" in ln:
                        print("   " + ln)
                if r:
                    warm = r[5:]
                    print("   => first call %.3f + %.3f = %.3f ms; median of calls 6..20: tPreprocess %.3f ms, tMatch %.3f ms, sum %.3f ms"
                          " (%.1f Mpix/s); candidates %d / %d, supports %d" % (
                              r[0][0], r[0][1], r[0][0] + r[0][1], statistics.median(x[0] for x in warm),
                              statistics.median(x[1] for x in warm), statistics.median(x[0] + x[1] for x in warm),
                              2.0 * W * H / statistics.median(x[0] + x[1] for x in warm) / 1e3, r[0][2], r[0][3], r[0][4]))


if __name__ == "__main__":
    main()
