#!/bin/bash
# Compiles gpc_hip.hip for gfx950 with -Rpass-analysis=kernel-resource-usage and prints one line per kernel whose
# demangled name matches $1 (default: all): SGPRs, VGPRs, spills, scratch, occupancy, static LDS.
# usage: bash tools/kres.sh [name-regex] [extra hipcc flags...]
R=$(cd "$(dirname "$0")/.." && pwd)
pat=${1:-.}; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Rpass-analysis=kernel-resource-usage "$@" \
  -o ${KRES_OUT:-/tmp/libgpc_kres.so} "$R/opengpc_amd/csrc/gpc_hip.hip" 2>&1 |
python3 -c '
import re, subprocess, sys
pat = re.compile(sys.argv[1])
cur = None; rows = []
for line in sys.stdin:
    m = re.search(r"remark: [^ ]+ +(Function Name|Name): (\S+)", line) or re.search(r": +(Function Name|Name): (\S+)", line)
    if m:
        cur = {"name": m.group(2)}; rows.append(cur); continue
    m = re.search(r": +(TotalSGPRs|SGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur is not None: cur[m.group(1).split(" [")[0]] = int(m.group(2))
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n")
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    if not pat.search(n): continue
    print("%-58s sgpr %3d vgpr %3d  spill s %3d v %3d  scratch %4d  occ %d  lds %6d" % (n, r.get("TotalSGPRs", r.get("SGPRs", -1)), r.get("VGPRs", -1), r.get("SGPRs Spill", 0), r.get("VGPRs Spill", 0), r.get("ScratchSize", 0), r.get("Occupancy", 0), r.get("LDS Size", 0)))
' "$pat"
