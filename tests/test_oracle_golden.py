"""Pins the CPU oracle (oracle/gpc_oracle.c) to the reference's known-answer vectors
(SURVEY.md Appendix C -> tests/golden/appendix_c.json).  CPU only."""
import numpy as np
import pytest

from oracle.pyoracle import sparsematch_settings, supports_fnv


def hx(v):
    return "%016x" % v


@pytest.fixture(scope="module", params=[0, 1], ids=["96x64", "1024x436"])
def case(request, golden, oracle):
    c = golden["cases"][request.param]
    L, R = oracle.synth_pair(c["W"], c["H"], c["s"], c["D"])
    pre = [oracle.preprocess(im, 5) for im in (L, R)]
    return c, (L, R), pre


def test_generator(case, oracle):
    c, (L, R), _ = case
    assert [hx(oracle.fnv(L)), hx(oracle.fnv(R))] == c["raw"]


def test_preprocess(case, oracle):
    c, _, pre = case
    for i, (smooth, grad, mask) in enumerate(pre):
        assert hx(oracle.fnv(smooth)) == c["smooth"][i]
        assert hx(oracle.fnv(grad)) == c["grad"][i]
        assert len(mask) == c["n_cand"][i]
        assert hx(oracle.fnv(mask)) == c["mask"][i]
    assert list(pre[0][2][:3]) == c["first_idx_l"]


@pytest.mark.parametrize("forest", ["zero", "tau"])
def test_codes(case, oracle, forest_paths, forest):
    c, _, pre = case
    W, H = c["W"], c["H"]
    rc, f = oracle.read_forest(forest_paths[forest], W, H)
    assert rc == 0 and f.num_tests == 30 and f.type == (0 if forest == "zero" else 1)
    for i, (smooth, grad, mask) in enumerate(pre):
        codes = oracle.hash(smooth, grad, f)
        assert hx(oracle.fnv(codes.reshape(-1)[mask])) == c[forest]["codes"][i]
        if i == 0:
            for xy, want in c[forest]["probe_l"].items():
                x, y = map(int, xy.split(","))
                assert "%08x" % codes[y, x] == want
    if "cand100_l" in c[forest]:
        k = int(pre[0][2][100])
        assert [k % W, k // W] == c[forest]["cand100_l"]


@pytest.mark.parametrize("forest", ["zero", "tau"])
@pytest.mark.parametrize("mode", ["epipolar", "global"])
def test_supports(case, oracle, forest_paths, forest, mode):
    c, (L, R), _ = case
    rc, f = oracle.read_forest(forest_paths[forest], c["W"], c["H"])
    supp, nl, nr = oracle.match_pair(L, R, f, sparsematch_settings(epipolar=(mode == "epipolar")))
    want = c[forest][mode]
    assert [nl, nr] == c["n_cand"]
    assert len(supp) == want["n"]
    assert hx(supports_fnv(oracle, supp)) == want["fnv"]
    if "first" in want:
        got = [[int(s["x"]), int(s["y"]), int(s["d"])] for s in supp[:2]]
        assert got == want["first"]
        assert [int(supp[-1]["x"]), int(supp[-1]["y"]), int(supp[-1]["d"])] == want["last"]
    if "all_d" in want:
        assert np.all(supp["d"] == want["all_d"])
    # rows H-15, H-14 carry code 0 (never computed): last support row is H-16
    assert supp["y"].max() == c["H"] - 16


def test_fixture_matches_the_survey_table():
    """The JSON fixture is a transcription of SURVEY.md Appendix C (vectors of the compiled reference)."""
    import subprocess
    import sys
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    subprocess.check_call([sys.executable, os.path.join(here, "golden", "check_against_survey.py")])


def test_fast_build_of_the_oracle_equals_the_plain_build(oracle, forest_paths, golden):
    """libgpc_oracle_fast.so is the same gpc_oracle.c compiled -O3 -march=native; the 256-pair GPU test and the
    bench's parity gate use it as the checker, so it is held to the plain build and to Appendix C here."""
    from oracle.pyoracle import Oracle
    from opengpc_amd.synth import synth_pair
    fast = Oracle(fast=True)
    for (W, H, s, D, fo) in [(1024, 436, 0, 24, "zero"), (1024, 436, 77, 21, "tau"), (272, 61, 5, 9, "tau")]:
        L, R = synth_pair(W, H, s, D)
        rc, f = oracle.read_forest(forest_paths[fo], W, H)
        for epi in (True, False):
            a, nl, nr = oracle.match_pair(L, R, f, sparsematch_settings(epipolar=epi))
            b, ml, mr = fast.match_pair(L, R, f, sparsematch_settings(epipolar=epi))
            assert (nl, nr) == (ml, mr) and np.array_equal(a, b)
            if (W, s, fo) == (1024, 0, "zero"):
                want = golden["cases"][1]["zero"]["epipolar" if epi else "global"]
                assert len(b) == want["n"] and hx(supports_fnv(fast, b)) == want["fnv"]
