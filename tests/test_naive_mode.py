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
