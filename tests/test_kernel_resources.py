"""CPU (hipcc cross-compiles without a GPU, ~25 s): the register claims DESIGN.md and profiles/README.md make about the
hot-path kernels, held by the compiler's own resource report (tools/kres.sh = hipcc -Rpass-analysis=kernel-resource-usage).

A kernel that spills VECTOR registers goes to scratch memory, and a scratch reload waits for every vector-memory
operation its wave has in flight (docs/HISTORY.md 3): no instantiation a BASELINE configuration can reach may do that.
Scalar registers spilled to VGPR lanes are cheap (a v_readlane) and are only bounded, not forbidden."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = re.compile(r"^(gpc::\S.*?)\s+sgpr\s+(\d+)\s+vgpr\s+(\d+)\s+spill s\s+(\d+)\s+v\s+(\d+)\s+scratch\s+(\d+)\s+occ\s+(\d+)\s+lds\s+(\d+)")


@pytest.fixture(scope="module")
def kres(tmp_path_factory):
    out = tmp_path_factory.mktemp("kres") / "libgpc_kres.so"
    env = dict(os.environ, KRES_OUT=str(out))
    txt = subprocess.run(["bash", os.path.join(ROOT, "tools", "kres.sh"), "."], env=env, check=True, capture_output=True,
                         text=True, timeout=900).stdout
    rows = {}
    for line in txt.splitlines():
        m = LINE.match(line)
        if m:
            rows[m.group(1).strip()] = dict(zip(("sgpr", "vgpr", "sspill", "vspill", "scratch", "occ", "lds"), map(int, m.groups()[1:])))
    assert len(rows) > 60, txt[-2000:]
    return rows


def pick(rows, pattern):
    rx = re.compile(pattern)
    got = {k: v for k, v in rows.items() if rx.search(k)}
    assert got, pattern
    return got


def test_fused_join_every_instantiation(kres):
    """k_row_join_fused<SPT, NT, WIDE>: the launch of every batched BASELINE configuration (and of every row up to 4096 px)."""
    got = pick(kres, r"k_row_join_fused<")
    assert len(got) >= 18          # SPT 1 | 2 | 4 x NT 256 | 512 | 1024 x WIDE
    for name, r in got.items():
        assert r["scratch"] == 0 and r["vspill"] == 0, (name, r)
        assert r["occ"] == 8 and r["vgpr"] <= 64, (name, r)      # eight waves per SIMD: what hides the LDS latency
        wide = name.rstrip(">").endswith("true")                 # WIDE: 32-test SSE=OFF codes, on no BASELINE path
        assert r["sspill"] <= (32 if wide else 12), (name, r)    # (round 3's fused instantiations: 42 .. 55)
    hot = got["gpc::k_row_join_fused<4, 256, false>"]           # the bench's kernel
    assert hot["sspill"] <= 10 and hot["vgpr"] <= 62, hot     # (5 spilled scalars, 60 VGPRs since the search keys have registers of their own)


def test_two_launch_and_partition_joins_up_to_four_slots(kres):
    """k_row_join<SPT <= 4, ...>: small launches, the host entry point's gap-free packing, the non-epipolar partitions (VIRT);
    WIDE (32-test SSE=OFF codes) included."""
    got = pick(kres, r"k_row_join<[124], ")
    assert len(got) >= 20
    for name, r in got.items():
        assert r["scratch"] == 0 and r["vspill"] == 0, (name, r)
        assert r["occ"] == 8, (name, r)


def test_hash_preprocess_and_hash_table_kernels(kres):
    for pat, min_occ in ((r"k_hash<", 4), (r"k_preprocess<false", 8), (r"k_ht_join<4, ", 8), (r"k_gather_rows", 8)):
        for name, r in pick(kres, pat).items():
            assert r["scratch"] == 0 and r["vspill"] == 0, (name, r)
            assert r["occ"] >= min_occ, (name, r)
    # two workgroups of the hash kernel per CU: 2 x 66.8 KB of LDS and <= 128 VGPRs
    for name, r in pick(kres, r"k_hash<").items():
        assert r["lds"] <= 80 * 1024 and r["vgpr"] <= 128, (name, r)
