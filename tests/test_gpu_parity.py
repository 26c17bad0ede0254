"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the committed
golden vectors.  Integer/byte work: bit-exact.  Disparities are integer-valued floats: exact."""
import os

import numpy as np
import pytest

from oracle.pyoracle import sparsematch_settings, supports_fnv

pytestmark = pytest.mark.gpu


def hx(v):
    return "%016x" % v


@pytest.fixture(scope="module")
def ctx():
    import opengpc_amd as g
    c = g.Context(0)
    yield c
    c.close()


def gpu_settings(epipolar=True, thr=5, disp_high=128, vtol=0):
    import opengpc_amd as g
    return g.Settings(thr, disp_high, vtol, epipolar, False, 1)


def test_library_is_the_hip_one(ctx):
    import opengpc_amd.capi as capi
    assert capi.LIB_PATH.endswith("opengpc_amd/libgpc_hip.so")
    assert ctx.L.gpc_hip_abi_version() == 1


def images(W, H, seed):
    rng = np.random.default_rng(seed)
    noise = rng.integers(0, 256, (H, W), dtype=np.uint8)
    blocky = (rng.integers(0, 256, (H // 4 + 1, W // 4 + 1)).repeat(4, 0).repeat(4, 1)[:H, :W] * 3 // 4
              + rng.integers(0, 64, (H, W))).astype(np.uint8)
    sat = np.where(rng.random((H, W)) < 0.5, 0, 255).astype(np.uint8)
    flat = np.full((H, W), 77, np.uint8)
    return [noise, blocky, sat, flat]


SHAPES = [(96, 64), (160, 101), (176, 67), (48, 41), (1024, 436), (1936, 120)]


@pytest.mark.parametrize("W,H", SHAPES)
@pytest.mark.parametrize("thr", [5, 0, 40, 182])
def test_preprocess(ctx, oracle, W, H, thr):
    for img in images(W, H, 10):
        s, g, m = ctx.preprocess(img, thr)
        so, go, mo = oracle.preprocess(img, thr)
        assert np.array_equal(s, so)
        assert np.array_equal(g, go)
        assert np.array_equal(m, mo)


@pytest.mark.parametrize("rows", [14, 10, 7, 6, 4, 2])
def test_preprocess_every_strip_height_on_ragged_images(oracle, forest_paths, rows):
    """k_preprocess has six strip heights (14 / 10 / 7 / 6 / 4 / 2 rows per thread) picked by the launch's size: each forced
    (GPC_HIP_PRE_ROWS) on heights that leave the last strip -- and the last block of four strips -- partly or wholly empty,
    in both arithmetics, and through a whole match_pair."""
    import opengpc_amd as g
    os.environ["GPC_HIP_PRE_ROWS"] = str(rows)
    try:
        c = g.Context(0)
    finally:
        del os.environ["GPC_HIP_PRE_ROWS"]
    try:
        for (W, H) in [(96, 41), (160, 57), (176, 113), (48, 31), (1936, 59), (272, 171)]:
            for img in images(W, H, 10 + rows):
                for thr in (5, 40):
                    s, gr, m = c.preprocess(img, thr)
                    so, go, mo = oracle.preprocess(img, thr)
                    assert np.array_equal(s, so) and np.array_equal(gr, go) and np.array_equal(m, mo), (W, H, thr)
            c.set_arithmetic(True)     # the reference built with SSE=OFF: k_preprocess<true, rows>
            for img in images(W, H, 20 + rows)[:2]:
                s, gr, m = c.preprocess(img, 5)
                so, go, mo = oracle.preprocess_naive(img, 5)
                assert np.array_equal(s, so) and np.array_equal(gr, go) and np.array_equal(m, mo), (W, H, "naive")
            c.set_arithmetic(False)
        rc, f = oracle.read_forest(forest_paths["zero"], 176, 113)
        c.load_forest(forest_paths["zero"], 176, 113)
        base = images(176 + 64, 113, 77)[1]
        L, R = np.ascontiguousarray(base[:, 32:32 + 176]), np.ascontiguousarray(base[:, 40:40 + 176])
        so, nl, nr = oracle.match_pair(L, R, f, sparsematch_settings(5, 128, 0, True))
        sg, n, ncand, st = c.match_pair(L, R, gpu_settings(True, 5, 128, 0))
        assert st == 0 and (nl, nr) == tuple(ncand) and n == len(so) and np.array_equal(sg, so.astype(sg.dtype))
    finally:
        c.close()


@pytest.mark.parametrize("W,H", SHAPES)
@pytest.mark.parametrize("forest", ["zero", "tau"])
def test_hash_codes_dense(ctx, oracle, forest_paths, W, H, forest):
    fm = ctx.load_forest(forest_paths[forest], W, H)
    rc, f = oracle.read_forest(forest_paths[forest], W, H)
    assert list(fm.mask[:60]) == list(f.offs[:60]) and fm.type == f.type
    for img in images(W, H, 11)[:3]:
        smooth, grad, mask = oracle.preprocess(img, 5)
        grad2 = grad.copy()
        grad2[:, (np.arange(W) // 16) % 3 == 0] = 0
        for g in (grad, grad2):
            assert np.array_equal(ctx.hash_codes(smooth, g), oracle.hash(smooth, g, f))


@pytest.mark.parametrize("taus", ["random", "edges", "random_without_m128", "edges_without_m128", "one_m128"])
def test_hash_codes_32_tests_full_tau_range(ctx, oracle, taus):
    """32 tests whose tau covers int8: random values, and every value at which the saturating subtract changes regime
    (+-127 / +-1 / 0 and their neighbours).  A forest that holds a tau of -128 takes k_hash's every-tau form of the
    complemented subtract (GpcForestDev::tau_m128), every other forest the four-operation form: both, and a forest whose only
    -128 is its last test."""
    W, H = 160, 100
    rng = np.random.default_rng(5)
    edge = [-128, -127, -126, -1, 0, 1, 2, 126, 127, -128, 64, -64, 127, -127, 1, -1]
    if taus == "edges_without_m128":
        edge = [-127, -126, -125, -1, 0, 1, 2, 126, 127, 3, 64, -64, 127, -127, 1, -1]
    lines = ["4"]
    for fern in range(4):
        lines.append("%d l 8" % fern)
        for t in range(8):
            ix, iy, jx, jy = rng.integers(-13, 14, 4)
            if taus == "random":
                tau = rng.integers(-128, 128) if (fern, t) != (1, 3) else -128
            elif taus in ("random_without_m128", "one_m128"):
                tau = rng.integers(-127, 128)
                if taus == "one_m128" and (fern, t) == (3, 7):
                    tau = 128      # (int8_t) 128 = -128: _mm_set1_epi8 truncates (filter.hpp:651)
            else:
                tau = edge[(fern * 8 + t) % len(edge)]
            lines.append("%d %d %d %d %d %d" % (t, ix, iy, jx, jy, tau))
    text = "\n".join(lines)
    import opengpc_amd as g
    st, fm = g.parse_forest(text, W, H)
    assert st == 0 and fm.num_tests == 32
    rc, f = oracle.parse_forest_text(text, W, H)
    ctx.set_forest(fm)
    for img in images(W, H, 6)[:3]:
        smooth, grad, _ = oracle.preprocess(img, 5)
        assert np.array_equal(ctx.hash_codes(smooth, grad), oracle.hash(smooth, grad, f))
        # the image itself as the smoothed plane: every byte value (0 / 127 / 128 / 255 beside each other) meets every tau
        assert np.array_equal(ctx.hash_codes(img, grad), oracle.hash(img, grad, f))
    ramp = ((np.arange(W)[None, :] * 5 + np.arange(H)[:, None] * 3) % 256).astype(np.uint8)
    grad = np.full((H, W), 9, np.uint8)
    assert np.array_equal(ctx.hash_codes(ramp, grad), oracle.hash(ramp, grad, f))


@pytest.mark.parametrize("case_idx", [0, 1], ids=["96x64", "1024x436"])
@pytest.mark.parametrize("forest", ["zero", "tau"])
@pytest.mark.parametrize("mode", ["epipolar", "global"])
def test_golden_supports(ctx, oracle, golden, forest_paths, case_idx, forest, mode):
    c = golden["cases"][case_idx]
    W, H = c["W"], c["H"]
    L, R = oracle.synth_pair(W, H, c["s"], c["D"])
    ctx.load_forest(forest_paths[forest], W, H)
    supp, n, ncand, st = ctx.match_pair(L, R, gpu_settings(epipolar=(mode == "epipolar")))
    want = c[forest][mode]
    assert st == 0
    assert list(ncand) == c["n_cand"]
    assert n == want["n"]
    assert hx(supports_fnv(oracle, supp)) == want["fnv"]


@pytest.mark.parametrize("W,H", [(96, 64), (176, 67), (1024, 436)])
@pytest.mark.parametrize("epipolar", [True, False])
def test_match_pair_vs_oracle_random(ctx, oracle, forest_paths, W, H, epipolar):
    """Translated noisy pairs: plenty of duplicates, non-matches and out-of-range disparities."""
    rng = np.random.default_rng(21)
    rc, f = oracle.read_forest(forest_paths["tau"], W, H)
    ctx.load_forest(forest_paths["tau"], W, H)
    for trial in range(3):
        base = images(W + 64, H, 30 + trial)[1]
        d = int(rng.integers(0, 40))
        L = np.ascontiguousarray(base[:, 32:32 + W])
        R = np.ascontiguousarray(base[:, 32 + d:32 + d + W])
        R = np.where(rng.random(R.shape) < 0.02, rng.integers(0, 256, R.shape), R).astype(np.uint8)
        for disp_high, vtol in ((128, 0), (16, 1)):
            so, nl, nr = oracle.match_pair(L, R, f, sparsematch_settings(5, disp_high, vtol, epipolar))
            sg, n, ncand, st = ctx.match_pair(L, R, gpu_settings(epipolar, 5, disp_high, vtol))
            assert (nl, nr) == tuple(ncand)
            assert n == len(so)
            assert np.array_equal(sg, so.astype(sg.dtype))


def test_tail_quirks_epipolar(ctx, oracle, forest_paths):
    """Rows H-15/H-14 carry code 0; the last target row decides quirks Q1/Q2."""
    W, H = 96, 64
    rc, f = oracle.read_forest(forest_paths["zero"], W, H)
    ctx.load_forest(forest_paths["zero"], W, H)
    base = images(W, H, 40)[1]
    smooth, grad, mask = oracle.preprocess(base, 5)
    y = H - 14
    for nl, nr in [(1, 1), (1, 2), (1, 3), (2, 2), (0, 2), (1, 0)]:
        gl, gr = grad.copy(), grad.copy()
        gl[y, :] = 0
        gr[y, :] = 0
        gl[y, 20:20 + nl] = 255
        gr[y, 30:30 + nr] = 255
        ml = np.flatnonzero(gl.reshape(-1)).astype(np.int32)
        mr = np.flatnonzero(gr.reshape(-1)).astype(np.int32)
        keep = lambda m: m[(m % W >= 13) & (m % W < W - 13) & (m // W >= 13) & (m // W < H - 13)]
        ml, mr = keep(ml), keep(mr)
        s = sparsematch_settings(5, 128, 0, True)
        cl = oracle.hash(smooth, gl, f)
        cr = oracle.hash(smooth, gr, f)
        corr = oracle.find_correspondences(oracle.descriptors(cl, ml, W, True), ml,
                                           oracle.descriptors(cr, mr, W, True), mr, W)
        want = oracle.rectified_filter(corr, s)
        got, n, st = ctx.rectified_match((smooth, gl, ml), (smooth, gr, mr), gpu_settings(True))
        assert n == len(want), (nl, nr)
        assert np.array_equal(got, want.astype(got.dtype)), (nl, nr)


def test_stereo_match_correspondences(ctx, oracle, forest_paths):
    W, H = 176, 67
    rc, f = oracle.read_forest(forest_paths["zero"], W, H)
    ctx.load_forest(forest_paths["zero"], W, H)
    base = images(W + 32, H, 50)[1]
    L = np.ascontiguousarray(base[:, 16:16 + W])
    R = np.ascontiguousarray(base[:, 25:25 + W])
    pl = oracle.preprocess(L, 5)
    pr = oracle.preprocess(R, 5)
    for epi in (True, False):
        cl = oracle.hash(pl[0], pl[1], f)
        cr = oracle.hash(pr[0], pr[1], f)
        want = oracle.find_correspondences(oracle.descriptors(cl, pl[2], W, epi), pl[2],
                                           oracle.descriptors(cr, pr[2], W, epi), pr[2], W)
        got, n, st = ctx.stereo_match(pl, pr, gpu_settings(epi))
        assert n == len(want)
        assert np.array_equal(got["src_x"], want["sx"]) and np.array_equal(got["tar_x"], want["tx"])
        assert np.array_equal(got["src_y"], want["sy"]) and np.array_equal(got["tar_y"], want["ty"])


def test_capacity_and_errors(ctx, oracle, forest_paths):
    import opengpc_amd as g
    W, H = 96, 64
    L, R = oracle.synth_pair(W, H, 0, 5)
    ctx.load_forest(forest_paths["zero"], W, H)
    full, n, _, st = ctx.match_pair(L, R, gpu_settings(True))
    assert st == 0 and n == 1044
    part, n2, _, st2 = ctx.match_pair(L, R, gpu_settings(True), cap=100)
    assert st2 == g.capi.E_CAPACITY and n2 == 1044 and np.array_equal(part, full[:100])
    with pytest.raises(g.GpcError):  # forest was read for another image size
        ctx.match_pair(np.zeros((64, 112), np.uint8), np.zeros((64, 112), np.uint8), gpu_settings(True))
    with pytest.raises(g.GpcError):  # width not a multiple of 16
        ctx.preprocess(np.zeros((64, 100), np.uint8), 5)
    s = gpu_settings(True)
    s.gradient_threshold = 300
    with pytest.raises(g.GpcError):
        ctx.match_pair(L, R, s)


def test_batch_equals_single(ctx, oracle, forest_paths):
    W, H, P = 1024, 436, 6
    ctx.load_forest(forest_paths["zero"], W, H)
    Ls, Rs = [], []
    for i in range(P):
        L, R = oracle.synth_pair(W, H, i, 8 + (i % 64))
        Ls.append(L)
        Rs.append(R)
    cap = 300000
    out, counts, ncand, st = ctx.match_batch(np.stack(Ls), np.stack(Rs), gpu_settings(True), cap)
    assert st == 0
    rc, f = oracle.read_forest(forest_paths["zero"], W, H)
    for i in range(P):
        single, n, nc, _ = ctx.match_pair(Ls[i], Rs[i], gpu_settings(True))
        assert counts[i] == n and np.array_equal(out[i, :n], single)
        assert tuple(ncand[i]) == nc
        if i < 2:
            so, nl, nr = oracle.match_pair(Ls[i], Rs[i], f, sparsematch_settings())
            assert np.array_equal(single, so.astype(single.dtype))
            assert np.median(single["d"]) == 8 + i  # a few chance collisions carry other disparities


@pytest.mark.parametrize("W,H", [(96, 64), (176, 67), (1024, 436)])
@pytest.mark.parametrize("epipolar", [True, False])
def test_hashtable_mode_vs_oracle(ctx, oracle, forest_paths, W, H, epipolar):
    """settings.useHashtable_ = true: ndb::Hashmatch semantics (214673 buckets, 10 per bucket,
    pair/triplet rules, bucket order).  The oracle's restatement is checked against the reference's
    own template in tests/test_oracle_vs_ref.py."""
    import opengpc_amd as g
    rng = np.random.default_rng(77)
    rc, f = oracle.read_forest(forest_paths["zero"], W, H)
    ctx.load_forest(forest_paths["zero"], W, H)
    for trial in range(2):
        base = images(W + 64, H, 60 + trial)[1]
        d = int(rng.integers(0, 40))
        L = np.ascontiguousarray(base[:, 32:32 + W])
        R = np.ascontiguousarray(base[:, 32 + d:32 + d + W])
        for disp_high, vtol in ((128, 0), (16, 1)):
            so, nl, nr = oracle.match_pair(L, R, f, sparsematch_settings(5, disp_high, vtol, epipolar, True))
            sg, n, ncand, st = ctx.match_pair(L, R, g.Settings(5, disp_high, vtol, epipolar, True, 1))
            assert st == 0 and (nl, nr) == tuple(ncand)
            assert n == len(so)
            assert np.array_equal(sg, so.astype(sg.dtype))


def test_hashtable_mode_bucket_overflow_and_triplets(ctx, oracle, forest_paths):
    """Striped images: few distinct codes, so buckets overflow their 10-element cap and the
    pair/triplet rules of getDuplicates decide."""
    import opengpc_amd as g
    W, H = 256, 64
    rc, f = oracle.read_forest(forest_paths["zero"], W, H)
    ctx.load_forest(forest_paths["zero"], W, H)
    img = np.tile((np.arange(W) // 3 * 37 % 256).astype(np.uint8), (H, 1))
    img2 = np.roll(img, 5, axis=1)
    for ep in (True, False):
        so, nl, nr = oracle.match_pair(img, img2, f, sparsematch_settings(5, 128, 1, ep, True))
        sg, n, ncand, st = ctx.match_pair(img, img2, g.Settings(5, 128, 1, ep, True, 1))
        assert (nl, nr) == tuple(ncand) and nl > 0
        assert n == len(so) and np.array_equal(sg, so.astype(sg.dtype))


@pytest.mark.parametrize("epipolar,hashtable", [(False, False), (True, True), (False, True)])
def test_batched_device_wide_modes_equal_single(ctx, oracle, forest_paths, epipolar, hashtable):
    """The non-epipolar and hash-table matchers run one launch per kernel over the whole batch."""
    import opengpc_amd as g
    from opengpc_amd.synth import synth_pair
    W, H, P = 272, 61, 5
    ctx.load_forest(forest_paths["tau"], W, H)
    Ls, Rs = zip(*[synth_pair(W, H, 10 + i, 5 + 3 * i) for i in range(P)])
    s = g.Settings(5, 128, 1, epipolar, hashtable, 1)
    out, counts, ncand, st = ctx.match_batch(np.stack(Ls), np.stack(Rs), s, 20000)
    assert st == 0
    rc, f = oracle.read_forest(forest_paths["tau"], W, H)
    for i in range(P):
        want, nl, nr = oracle.match_pair(Ls[i], Rs[i], f, sparsematch_settings(5, 128, 1, epipolar, hashtable))
        assert (nl, nr) == tuple(ncand[i]) and counts[i] == len(want)
        assert np.array_equal(out[i, :counts[i]], want.astype(out.dtype))
