"""The reference's SSE=OFF build (*Naive kernels, filter.hpp:157-282): different smooth images,
masks and codes than the default SSE build.  CPU: the oracle's restatement against the reference
compiled without -D_INTRINSICS_SSE (oracle/_ref/libgpc_ref_naive.so).  GPU: GPC_ARITH_NAIVE."""
import numpy as np
import pytest

from oracle.pyoracle import Ref, sparsematch_settings

SHAPES = [(96, 64), (160, 101), (48, 41), (1024, 436)]


def images(W, H, seed):
    rng = np.random.default_rng(seed)
    noise = rng.integers(0, 256, (H, W), dtype=np.uint8)
    blocky = (rng.integers(0, 256, (H // 4 + 1, W // 4 + 1)).repeat(4, 0).repeat(4, 1)[:H, :W] * 3 // 4
              + rng.integers(0, 64, (H, W))).astype(np.uint8)
    sat = np.where(rng.random((H, W)) < 0.5, 0, 255).astype(np.uint8)
    return [noise, blocky, sat]


def wild_forest_text(rng, ntests_per_fern=8, ferns=4):
    lines = [str(ferns)]
    for fern in range(ferns):
        lines.append("%d l %d" % (fern, ntests_per_fern))
        for t in range(ntests_per_fern):
            ix, iy, jx, jy = rng.integers(-13, 14, 4)
            lines.append("%d %d %d %d %d %d" % (t, ix, iy, jx, jy, rng.integers(-300, 301)))
    return "\n".join(lines)


@pytest.mark.skipif(not Ref.available(naive=True), reason="oracle/_ref naive build not present")
@pytest.mark.parametrize("W,H", SHAPES)
def test_oracle_naive_vs_reference_naive_build(oracle, forest_paths, W, H):
    ref = Ref(naive=True)
    assert ref.lib.gpc_ref_is_sse() == 0
    rng = np.random.default_rng(9)
    for img in images(W, H, 3):
        assert np.array_equal(ref.box(img), oracle.box_naive(img))
        for thr in (0, 5, 40, 200, 255):
            assert np.array_equal(ref.sobel(img, thr), oracle.sobel_naive(img, thr))
        sm, gr, m = oracle.preprocess_naive(img, 5)
        assert np.array_equal(ref.arr2ind(gr), np.flatnonzero(gr.reshape(-1)).astype(np.int32))
        forests = [oracle.read_forest(p, W, H)[1] for p in forest_paths.values()]
        forests.append(oracle.parse_forest_text(wild_forest_text(rng), W, H)[1])   # 32 tests, |tau| up to 300
        for f in forests:
            assert np.array_equal(ref.hash_idx(sm, gr, f, m), oracle.hash_naive(sm, m, f))


def test_naive_differs_from_sse(oracle, forest_paths):
    """Sanity: the two builds of the reference really are different algorithms."""
    W, H = 160, 100
    img = images(W, H, 4)[1]
    s1, g1, m1 = oracle.preprocess(img, 5)
    s2, g2, m2 = oracle.preprocess_naive(img, 5)
    assert not np.array_equal(s1, s2) and not np.array_equal(g1, g2)


@pytest.fixture(scope="module")
def nctx():
    import opengpc_amd as g
    c = g.Context(0)
    c.set_arithmetic(True)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("W,H", SHAPES + [(1936, 60)])
def test_gpu_naive_preprocess_and_codes(nctx, oracle, forest_paths, W, H):
    import opengpc_amd as g
    rng = np.random.default_rng(10)
    for img in images(W, H, 5):
        for thr in (5, 0, 200):
            s, gr, m = nctx.preprocess(img, thr)
            so, go, mo = oracle.preprocess_naive(img, thr)
            assert np.array_equal(s, so) and np.array_equal(gr, go) and np.array_equal(m, mo)
        sm, gr, m = oracle.preprocess_naive(img, 5)
        texts = [open(p).read() for p in forest_paths.values()] + [wild_forest_text(rng), wild_forest_text(rng, 5, 3)]
        for text in texts:
            st, fm = g.parse_forest(text, W, H)
            rc, f = oracle.parse_forest_text(text, W, H)
            nctx.set_forest(fm)
            for ty in (f.type, 0):       # also the same tests as a zero forest
                fm.type = f.type = ty
                nctx.set_forest(fm)
                assert np.array_equal(nctx.hash_codes(sm, gr), oracle.hash_naive(sm, m, f))


@pytest.mark.gpu
@pytest.mark.parametrize("epipolar", [True, False])
@pytest.mark.parametrize("hashtable", [False, True])
def test_gpu_naive_match_pair(nctx, oracle, forest_paths, epipolar, hashtable):
    import opengpc_amd as g
    from opengpc_amd.synth import synth_pair
    for (W, H, s, D, forest) in [(96, 64, 0, 5, "zero"), (1024, 436, 0, 24, "tau"), (272, 61, 3, 9, "tau")]:
        L, R = synth_pair(W, H, s, D)
        rc, f = oracle.read_forest(forest_paths[forest], W, H)
        nctx.load_forest(forest_paths[forest], W, H)
        want, nl, nr = oracle.match_pair(L, R, f, sparsematch_settings(5, 128, 0, epipolar, hashtable, True))
        got, n, ncand, st = nctx.match_pair(L, R, g.Settings(5, 128, 0, epipolar, hashtable, 1))
        assert st == 0 and (nl, nr) == ncand and n == len(want) and n > 0
        assert np.array_equal(got, want.astype(got.dtype))
        # and it is not the SSE answer
        sse, _, _ = oracle.match_pair(L, R, f, sparsematch_settings(5, 128, 0, epipolar, hashtable, False))
        assert len(sse) != len(want) or not np.array_equal(sse, want)


def mostly_true_forest_text(tau):
    """32 tests that hold for most pixels under the Naive predicate a > b - tau (filter.hpp:276): many
    candidates then carry the all-ones code 0xFFFFFFFF, the one value that collides with both sentinels of
    the HIP matchers (GPC_NOCAND in the code image, key 0 = code + 1 in the join table)."""
    rng = np.random.default_rng(1234)
    lines = ["4"]
    for fern in range(4):
        lines.append("%d l 8" % fern)
        for t in range(8):
            ix, iy, jx, jy = rng.integers(-13, 14, 4)
            lines.append("%d %d %d %d %d %d" % (t, ix, iy, jx, jy, tau))
    return "\n".join(lines)


@pytest.mark.gpu
@pytest.mark.parametrize("epipolar", [True, False])
@pytest.mark.parametrize("hashtable", [False, True])
def test_gpu_naive_32_tests_use_all_32_code_bits(nctx, oracle, epipolar, hashtable):
    """SSE=OFF arithmetic with a 32-test forest: test 0 lands on bit 31 and 0xFFFFFFFF is a legal code.
    Every matcher mode against the oracle on the stress forest (first 32 of 320 tests), wild 32-test
    forests (|tau| up to 300) and forests whose tests mostly hold; the sweep must actually produce
    supports whose code has bit 31 set and supports whose code is 0xFFFFFFFF."""
    import os
    import opengpc_amd as g
    from opengpc_amd.synth import synth_pair
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(321)
    texts = [open(os.path.join(root, "forests", "stress16x20Forest.txt")).read(), wild_forest_text(rng),
             wild_forest_text(rng), mostly_true_forest_text(25), mostly_true_forest_text(60), mostly_true_forest_text(300)]
    n_bit31 = n_ones = n_total = n_ones_cand = 0
    for (W, H, s, D) in [(96, 64, 0, 5), (272, 61, 3, 9), (1040, 44, 5, 17), (2064, 40, 2, 30)]:
        L, R = synth_pair(W, H, s, D)
        for text in texts:
            st, fm = g.parse_forest(text, W, H)
            rc, f = oracle.parse_forest_text(text, W, H)
            assert fm.num_tests == 32 and f.num_tests == 32
            nctx.set_forest(fm)
            for disp_high, vtol in ((128, 0), (12, 1)):
                want, nl, nr = oracle.match_pair(L, R, f, sparsematch_settings(5, disp_high, vtol, epipolar, hashtable, True))
                got, n, ncand, st = nctx.match_pair(L, R, g.Settings(5, disp_high, vtol, epipolar, hashtable, 1))
                assert st == 0 and (nl, nr) == ncand and n == len(want), (W, H, disp_high)
                assert np.array_equal(got, want.astype(got.dtype)), (W, H, disp_high)
            # which codes did the supports of the last run carry?
            sm, gr, m = oracle.preprocess_naive(L, 5)
            codes = oracle.hash_naive(sm, m, f)
            c = codes[want["y"], want["x"]]
            n_total += len(c)
            n_bit31 += int((c >> 31).sum())
            n_ones += int((c == 0xFFFFFFFF).sum())
            n_ones_cand += int((codes.reshape(-1)[m] == 0xFFFFFFFF).sum())
    # candidates with the all-ones code were there to be told from non-candidates (their counts are checked above)
    assert n_total > 0 and n_bit31 > 0 and n_ones_cand > 0
    if epipolar and not hashtable:
        assert n_ones > 0   # the key-less code went through the join's side channel and matched
