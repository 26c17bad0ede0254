"""Fern training scoring loop: Fern::evalSplit / markSplitSamples / train.

PARITY UNPINNED: the reference holds no tests or vectors for training and Fern.hpp / Feature.hpp
cannot be compiled in this image (Eigen), so the C oracle (oracle/gpc_oracle_train.c) is checked
here against an independent numpy restatement of the same reference lines, and the HIP path is
checked against the oracle (bit-exact counts and parameters; the double statistics are the same
expressions evaluated in the same order, compared exactly)."""
import numpy as np
import pytest

from oracle.pyoracle import SPLIT_DTYPE, STATS_DTYPE


def make_triplets(n, seed, noise=6):
    """ref random texture; pos = ref + small noise (a true match); neg = an unrelated patch that
    shares its low-frequency part with ref (so that shallow ferns confuse them)."""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (n, 729)).astype(np.int32)
    ref = base
    pos = np.clip(base + rng.integers(-noise, noise + 1, (n, 729)), 0, 255)
    neg = np.clip((base // 64) * 64 + rng.integers(0, 64, (n, 729)), 0, 255)
    return np.stack([ref, pos, neg], 1).astype(np.uint8)


def make_cands(count, seed):
    rng = np.random.default_rng(seed)
    c = np.zeros(count, SPLIT_DTYPE)
    c["i"] = rng.integers(0, 729, count)
    c["j"] = (c["i"] + rng.integers(1, 729, count)) % 729  # i != j like sampleHyperplane (Feature.hpp:128-167)
    c["tau"] = rng.integers(-15, 16, count)                # randTAU; overwritten by the tau loop
    return c


# ---- independent numpy model of Fern.hpp:209-262 / 271-291 / 312-372
def np_codes(t, params, count):
    ref = np.zeros(len(t), np.uint64)
    pos = np.zeros(len(t), np.uint64)
    neg = np.zeros(len(t), np.uint64)
    for l in range(count):
        i, j, tau = int(params["i"][l]), int(params["j"][l]), int(params["tau"][l])
        d = t[:, :, i].astype(np.int32) - t[:, :, j].astype(np.int32) < tau
        ref = (ref << np.uint64(1)) + d[:, 0].astype(np.uint64)
        pos = (pos << np.uint64(1)) + d[:, 1].astype(np.uint64)
        neg = (neg << np.uint64(1)) + d[:, 2].astype(np.uint64)
    return ref, pos, neg


def np_eval_split(t, marks, params, until, w1):
    ref, pos, neg = np_codes(t, params, until + 1)
    counted = ~(((marks & 1) != 0) & ((marks & 2) != 0))
    eq, ne = ref == pos, ref != neg
    tp = int(np.sum(counted & eq & ne))
    fp = int(np.sum(counted & ~eq & ~ne))
    tot = int(np.sum(counted))
    fn = tot - tp - fp
    w2 = 1.0 - w1
    prec = 0.0 if tp + fp == 0 else tp / (tp + fp)
    rec = 0.0 if tp + fn == 0 else tp / (tp + fn)
    hmean = 0.0 if prec + rec == 0.0 else prec * rec / ((1.0 - w2) * prec + w2 * rec)
    conv = (1.0 - w2) * prec + w2 * rec
    return dict(tp=tp, fp=fp, fn=fn, tot=tot, prec=prec, rec=rec, hmean=hmean, convcomb=conv)


def np_mark(t, marks, params, count):
    ref, pos, neg = np_codes(t, params, count)
    marks |= (ref == pos).astype(np.uint8)
    marks |= ((ref != neg).astype(np.uint8) << 1)


def np_train(t, marks, depth, cand, nres, taulo, tauhi, only, w1):
    fern = np.zeros(depth, SPLIT_DTYPE)
    stats_out = []
    stats = dict(tp=0, fp=0, fn=0, tot=0, prec=0.0, rec=0.0, hmean=0.0, convcomb=0.0)
    best = np.zeros(1, SPLIT_DTYPE)[0].copy()
    if only:
        marks[:] = 0
    for level in range(depth):
        max_score = np.float32(0.0)
        for k in range(nres):
            fern[level] = cand[level * nres + k]
            for tau in range(taulo, tauhi):
                fern["tau"][level] = tau
                stats = np_eval_split(t, marks, fern, level, w1)
                if stats["hmean"] > float(max_score):
                    best = fern[level].copy()
                    max_score = np.float32(stats["hmean"])
        fern[level] = best
        if only:
            np_mark(t, marks, fern, level)
        stats_out.append(dict(stats))
    return fern, stats_out


def stats_equal(a, b):
    for k in ("tp", "fp", "fn", "tot"):
        assert int(a[k]) == int(b[k]), (k, a, b)
    for k in ("prec", "rec", "hmean", "convcomb"):
        assert float(a[k]) == float(b[k]), (k, a, b)


# --------------------------------------------------------------------------- CPU: oracle vs numpy model
@pytest.mark.parametrize("n,seed", [(1, 1), (257, 2), (3000, 3)])
def test_oracle_eval_split_and_marks_vs_numpy(oracle, n, seed):
    t = make_triplets(n, seed)
    rng = np.random.default_rng(seed)
    marks = rng.integers(0, 4, n).astype(np.uint8)
    params = make_cands(6, seed + 10)
    params["tau"] = rng.integers(-4, 5, 6)
    for until in (0, 2, 5):
        for w1 in (0.5, 0.3):
            stats_equal(oracle.eval_split(t, marks, params, until, w1), np_eval_split(t, marks, params, until, w1))
    m1, m2 = marks.copy(), marks.copy()
    for count in (0, 1, 4):  # count 0: every code word is 0 -> pos.split set for all (Fern.hpp:286-287)
        oracle.mark_split_samples(t, m1, params, count)
        np_mark(t, m2, params, count)
        assert np.array_equal(m1, m2)
    assert np.all(m1 & 1)


@pytest.mark.parametrize("only,taulo,tauhi", [(False, 0, 1), (True, 0, 1), (True, -3, 4), (False, -2, 2)])
def test_oracle_train_fern_vs_numpy(oracle, only, taulo, tauhi):
    n, depth, nres = 1500, 4, 5
    t = make_triplets(n, 5)
    cand = make_cands(depth * nres, 6)
    m1 = np.random.default_rng(1).integers(0, 4, n).astype(np.uint8)
    m2 = m1.copy()
    fp, st = oracle.train_fern(t, m1, depth, cand, nres, taulo, tauhi, only, 0.5)
    fern, stats = np_train(t, m2, depth, cand, nres, taulo, tauhi, only, 0.5)
    assert np.array_equal(fp, fern)
    assert np.array_equal(m1, m2)
    for level in range(depth):
        stats_equal(st[level], stats[level])


def test_oracle_train_quirks(oracle):
    """A level on which no candidate scores above 0 keeps the previous best (Fern.hpp:352); with an
    empty tau range nothing is evaluated at all and the printed stats stay zero."""
    t = make_triplets(64, 9)
    t[:, 2] = t[:, 0]          # neg == ref: ref != neg never holds -> tp = 0 -> hmean = 0 everywhere
    cand = make_cands(6, 10)
    marks = np.zeros(64, np.uint8)
    fp, st = oracle.train_fern(t, marks, 3, cand, 2, 0, 1, False, 0.5)
    assert all(tuple(p) == (0, 0, 0) for p in fp) and all(s["tp"] == 0 for s in st)
    fp, st = oracle.train_fern(t, marks, 2, cand, 3, 2, 2, False, 0.5)
    assert all(tuple(p) == (0, 0, 0) for p in fp) and all(s["tot"] == 0 for s in st)


# --------------------------------------------------------------------------- GPU: HIP path vs oracle
@pytest.fixture(scope="module")
def ctx():
    import opengpc_amd as g
    c = g.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,seed", [(1, 1), (255, 2), (256, 3), (4097, 4), (20000, 5)])
def test_gpu_eval_split_and_marks(ctx, oracle, n, seed):
    t = make_triplets(n, seed)
    rng = np.random.default_rng(seed)
    marks = rng.integers(0, 4, n).astype(np.uint8)
    params = make_cands(8, seed + 10)
    params["tau"] = rng.integers(-6, 7, 8)
    ts = ctx.train_set(t)
    assert np.array_equal(ts.marks(), np.zeros(n, np.uint8))
    assert np.array_equal(ts.marks(marks), marks)
    for until in (0, 3, 7):
        for w1 in (0.5, 0.25):
            stats_equal(ts.eval_split(params, until, w1), oracle.eval_split(t, marks, params, until, w1))
    want = marks.copy()
    for count in (0, 2, 8):
        ts.mark_split_samples(params, count)
        oracle.mark_split_samples(t, want, params, count)
        assert np.array_equal(ts.marks(), want)
    ts.close()


@pytest.mark.gpu
@pytest.mark.parametrize("only,taulo,tauhi,w1", [(False, 0, 1, 0.5), (True, 0, 1, 0.5), (True, -10, 10, 0.5),
                                                 (False, -3, 4, 0.3), (True, 2, 2, 0.5)])
def test_gpu_train_fern_vs_oracle(ctx, oracle, only, taulo, tauhi, w1):
    n, depth, nres = 12345, 5, 10  # samples/train.cpp: depth 5, 10 resamples
    t = make_triplets(n, 21)
    cand = make_cands(depth * nres, 22)
    marks = np.random.default_rng(3).integers(0, 4, n).astype(np.uint8)
    ts = ctx.train_set(t)
    ts.marks(marks)
    fp, st = ts.train_fern(depth, cand, nres, taulo, tauhi, only, w1)
    want_marks = marks.copy()
    wfp, wst = oracle.train_fern(t, want_marks, depth, cand, nres, taulo, tauhi, only, w1)
    assert np.array_equal(fp, wfp)
    for level in range(depth):
        stats_equal(st[level], wst[level])
    assert np.array_equal(ts.marks(), want_marks)
    ts.close()


@pytest.mark.gpu
def test_gpu_level_pieces_and_errors(ctx, oracle):
    import opengpc_amd as g
    n = 5000
    t = make_triplets(n, 31)
    ts = ctx.train_set(t)
    cand = make_cands(7, 32)
    ts.begin_fern(True)
    tp, fp, tot = ts.eval_level(cand, -2, 3)
    marks = np.zeros(n, np.uint8)
    for k in range(7):
        for q, tau in enumerate(range(-2, 3)):
            p = cand[k:k + 1].copy()
            p["tau"] = tau
            s = oracle.eval_split(t, marks, p, 0, 0.5)
            assert (tp[k, q], fp[k, q], tot) == (s["tp"], s["fp"], s["tot"])
    best = cand[3:4].copy()
    best["tau"] = 1
    ts.commit_level(best[0], True)
    # level 1 against the oracle with level 0 fixed; marks after markSplitSamples(.., 0): pos.split everywhere
    oracle.mark_split_samples(t, marks, best, 0)
    assert np.array_equal(ts.marks(), marks)
    tp, fp, tot = ts.eval_level(cand[:2], 0, 1)
    for k in range(2):
        p = np.concatenate([best, cand[k:k + 1]])
        p["tau"][1] = 0
        s = oracle.eval_split(t, marks, p, 1, 0.5)
        assert (tp[k, 0], fp[k, 0], tot) == (s["tp"], s["fp"], s["tot"])
    bad = cand[:1].copy()
    bad["i"] = 729
    with pytest.raises(g.GpcError):
        ts.eval_level(bad, 0, 1)
    with pytest.raises(g.GpcError):
        ts.eval_level(cand, 0, 65)  # more than 64 intercepts per candidate
    with pytest.raises(ValueError):
        ctx.train_set(np.zeros((4, 3, 700), np.uint8))
    ts.close()


@pytest.mark.gpu
def test_gpu_training_limits(ctx, oracle):
    """The edges of the C ABI: 64 levels (the reference's code words have 64 bits), 64 intercepts per
    candidate, first / last patch byte, intercepts beyond the byte range, a set that is not a multiple
    of the 4 triplets a lane handles."""
    import opengpc_amd as g
    n = 1023
    t = make_triplets(n, 41, noise=40)
    ts = ctx.train_set(t)
    # 64 levels, one candidate each, a single intercept
    cand = make_cands(64, 42)
    cand["i"][:2], cand["j"][:2] = [0, 728], [728, 0]
    marks = np.zeros(n, np.uint8)
    fp, st = ts.train_fern(64, cand, 1, 0, 1, True, 0.5)
    wfp, wst = oracle.train_fern(t, marks, 64, cand, 1, 0, 1, True, 0.5)
    assert np.array_equal(fp, wfp) and np.array_equal(ts.marks(), marks)
    stats_equal(st[63], wst[63])
    s1 = ts.eval_split(wfp, 63, 0.5)
    stats_equal(s1, oracle.eval_split(t, marks, wfp, 63, 0.5))
    with pytest.raises(g.GpcError):
        ts.train_fern(65, np.concatenate([cand, cand[:1]]), 1, 0, 1, False, 0.5)
    # 64 intercepts straddling the whole difference range
    for taulo in (-300, -32, 200):
        marks2 = np.random.default_rng(taulo & 0xFF).integers(0, 4, n).astype(np.uint8)
        ts.marks(marks2)
        c2 = make_cands(3 * 2, 43)
        fp, st = ts.train_fern(3, c2, 2, taulo, taulo + 64, False, 0.3)
        want = marks2.copy()
        wfp, wst = oracle.train_fern(t, want, 3, c2, 2, taulo, taulo + 64, False, 0.3)
        assert np.array_equal(fp, wfp)
        for level in range(3):
            stats_equal(st[level], wst[level])
    ts.close()
