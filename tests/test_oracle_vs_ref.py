"""Checks the C restatement against the reference's own SSE kernels (oracle/_ref, built from
/root/reference/lib/gpc/filter.hpp).  Skipped where the reference build is absent."""
import numpy as np
import pytest

from oracle.pyoracle import Ref

pytestmark = pytest.mark.skipif(not Ref.available(), reason="oracle/_ref not built (no reference tree)")

SHAPES = [(96, 64), (160, 101), (176, 67), (48, 41), (1024, 436)]


@pytest.fixture(scope="module")
def ref():
    r = Ref()
    assert r.lib.gpc_ref_is_sse() == 1
    return r


def images(W, H, seed):
    rng = np.random.default_rng(seed)
    noise = rng.integers(0, 256, (H, W), dtype=np.uint8)
    smoothish = (rng.integers(0, 256, (H // 4 + 1, W // 4 + 1)).repeat(4, 0).repeat(4, 1)[:H, :W] * 3 // 4
                 + rng.integers(0, 64, (H, W))).astype(np.uint8)
    sat = np.where(rng.random((H, W)) < 0.5, 0, 255).astype(np.uint8)
    return [noise, smoothish, sat]


@pytest.mark.parametrize("W,H", SHAPES)
def test_box(ref, oracle, W, H):
    for img in images(W, H, 1):
        assert np.array_equal(ref.box(img), oracle.box(img))


@pytest.mark.parametrize("W,H", SHAPES)
@pytest.mark.parametrize("thr", [0, 5, 10, 40, 181, 182, 255])
def test_sobel(ref, oracle, W, H, thr):
    for img in images(W, H, 2):
        assert np.array_equal(ref.sobel(img, thr), oracle.sobel(img, thr))


@pytest.mark.parametrize("W,H", SHAPES)
def test_arr2ind(ref, oracle, W, H):
    rng = np.random.default_rng(3)
    grad = np.where(rng.random((H, W)) < 0.4, 255, 0).astype(np.uint8)
    got = ref.arr2ind(grad)
    got = got[got < W * H]  # the AVX loop may run 16 bytes past n when n % 32 == 16
    assert np.array_equal(got, np.flatnonzero(grad.reshape(-1)).astype(np.int32))


@pytest.mark.parametrize("W,H", SHAPES)
@pytest.mark.parametrize("forest", ["zero", "tau"])
def test_hash(ref, oracle, forest_paths, W, H, forest):
    rc, f = oracle.read_forest(forest_paths[forest], W, H)
    assert rc == 0
    for img in images(W, H, 4):
        smooth, grad, mask = oracle.preprocess(img, 5)
        # sparse gradient so that the 16-pixel group skip is exercised too
        grad2 = grad.copy()
        grad2[:, (np.arange(W) // 16) % 3 == 0] = 0
        for g in (grad, grad2):
            assert np.array_equal(ref.hash(smooth, g, f), oracle.hash(smooth, g, f))


def test_hash_random_forest_32_tests(ref, oracle):
    """T = 32 (all four byte planes full), offsets on the +-13 border, tau over int8 range."""
    W, H = 160, 100
    rng = np.random.default_rng(5)
    lines = ["4"]
    for fern in range(4):
        lines.append("%d l 8" % fern)
        for t in range(8):
            ix, iy, jx, jy = rng.integers(-13, 14, 4)
            lines.append("%d %d %d %d %d %d" % (t, ix, iy, jx, jy, rng.integers(-128, 128)))
    rc, f = oracle.parse_forest_text("\n".join(lines), W, H)
    assert rc == 0 and f.num_tests == 32 and f.type == 1
    for img in images(W, H, 6):
        smooth, grad, _ = oracle.preprocess(img, 5)
        assert np.array_equal(ref.hash(smooth, grad, f), oracle.hash(smooth, grad, f))
        f.type = 0
        assert np.array_equal(ref.hash(smooth, grad, f), oracle.hash(smooth, grad, f))
        f.type = 1


def test_hash_threads_identical(ref, oracle, forest_paths):
    W, H = 176, 67
    rc, f = oracle.read_forest(forest_paths["zero"], W, H)
    img = images(W, H, 7)[1]
    smooth, grad, _ = oracle.preprocess(img, 5)
    assert np.array_equal(ref.hash(smooth, grad, f, 1), ref.hash(smooth, grad, f, 4))


def test_hashmatch_restatement_vs_reference_template(ref, oracle):
    """oracle hash_correspondences == the reference's ndb::Hashmatch<T> (hashmatch.hpp) driven as
    depthPriorFast does: colliding buckets (multiples of 214673), overflow beyond 10, triplets."""
    rng = np.random.default_rng(1)
    pool = np.array([5, 5 + 214673, 5 + 2 * 214673, 7, 9, 9 + 214673 * 3, 11, (3 << 32) | 7], np.uint64)
    for trial in range(400):
        ns, nt = rng.integers(0, 40, 2)
        ss, ts = rng.choice(pool, ns), rng.choice(pool, nt)
        sk = rng.permutation(1000)[:ns].astype(np.int32)
        tk = (rng.permutation(1000)[:nt] + 2000).astype(np.int32)
        a = oracle.hash_correspondences(ss, sk, ts, tk, 100000)
        b = ref.hashmatch(ss, sk, ts, tk)
        assert [(int(c["sx"]) + 100000 * int(c["sy"]), int(c["tx"]) + 100000 * int(c["ty"])) for c in a] == \
               [(int(x), int(y)) for x, y in b]


def test_hashmatch_on_real_descriptors(ref, oracle, forest_paths):
    W, H = 1024, 436
    L, R = oracle.synth_pair(W, H, 0, 24)
    rc, f = oracle.read_forest(forest_paths["zero"], W, H)
    pl, pr = oracle.preprocess(L, 5), oracle.preprocess(R, 5)
    cl, cr = oracle.hash(pl[0], pl[1], f), oracle.hash(pr[0], pr[1], f)
    for epi in (True, False):
        sl, sr = oracle.descriptors(cl, pl[2], W, epi), oracle.descriptors(cr, pr[2], W, epi)
        a = oracle.hash_correspondences(sl, pl[2], sr, pr[2], W)
        b = ref.hashmatch(sl, pl[2], sr, pr[2])
        assert len(a) == len(b) and len(a) > 200000
        assert np.array_equal(a["sx"] + W * a["sy"], b[:, 0]) and np.array_equal(a["tx"] + W * a["ty"], b[:, 1])
