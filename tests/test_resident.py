"""Forest::preprocessImage -> Forest::rectifiedMatch without the round trip over the link (include/gpc_hip.h, "Resident
images"): a PreprocessedImage whose host arrays are the ones the library delivered is matched from the image it kept on
the device; anything else -- copies, edited arrays, arrays of another context -- takes the upload path.  Either way the
supports are the oracle's, bit for bit (reference: lib/gpc/inference.hpp:302-333, 375-393)."""
import os

import numpy as np
import pytest

from oracle.pyoracle import sparsematch_settings

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx(forest_paths):
    import opengpc_amd as g
    c = g.Context(0)
    yield c
    c.close()


def oracle_from_preprocessed(oracle, pl, pr, f, W, s, epipolar, hashtable):
    """the reference's rectifiedMatch on (smooth, grad, mask) triples as given (edited masks included)"""
    cl, cr = oracle.hash(pl[0], pl[1], f), oracle.hash(pr[0], pr[1], f)
    sl, sr = oracle.descriptors(cl, pl[2], W, epipolar), oracle.descriptors(cr, pr[2], W, epipolar)
    fn = oracle.hash_correspondences if hashtable else oracle.find_correspondences
    corr = fn(sl, pl[2], sr, pr[2], W)
    return corr, oracle.rectified_filter(corr, s)


@pytest.mark.parametrize("epipolar,hashtable", [(1, 0), (0, 0), (1, 1), (0, 1)])
@pytest.mark.parametrize("forest", ["zero", "tau"])
def test_resident_match_equals_upload_path_and_oracle(ctx, oracle, forest_paths, epipolar, hashtable, forest):
    import opengpc_amd as g
    W, H = 272, 61
    L, R = oracle.synth_pair(W, H, 3, 9)
    ctx.load_forest(forest_paths[forest], W, H)
    rc, f = oracle.read_forest(forest_paths[forest], W, H)
    gs = g.Settings(5, 64, 1, bool(epipolar), bool(hashtable), 1)
    os_ = sparsematch_settings(5, 64, 1, bool(epipolar), bool(hashtable), False)
    pl, pr = ctx.preprocess_resident(L, 5), ctx.preprocess_resident(R, 5)
    for got, want in zip(pl + pr, oracle.preprocess(L, 5) + oracle.preprocess(R, 5)):
        assert np.array_equal(got, want)
    h0 = ctx.resident_hits()
    supp, n, st = ctx.rectified_match(pl, pr, gs)
    corr, nc, st2 = ctx.stereo_match(pl, pr, gs)
    assert ctx.resident_hits() == h0 + 2 and st == 0 and st2 == 0
    wcorr, wsupp = oracle_from_preprocessed(oracle, pl, pr, f, W, os_, bool(epipolar), bool(hashtable))
    assert n == len(wsupp) and np.array_equal(supp, wsupp)
    assert nc == len(wcorr) and np.array_equal(corr.view(np.int32), wcorr.view(np.int32))
    # copies of the same arrays are not the arrays the library delivered: the upload path, the same supports
    cl, cr = tuple(a.copy() for a in pl), tuple(a.copy() for a in pr)
    supp2, n2, _ = ctx.rectified_match(cl, cr, gs)
    assert ctx.resident_hits() == h0 + 2 and np.array_equal(supp2, supp)
    # a short capacity on the resident path: true count, GPC_E_CAPACITY, the first records
    if n > 3:
        supp3, n3, st3 = ctx.rectified_match(pl, pr, gs, cap=n - 3)
        assert st3 == g.capi.E_CAPACITY and n3 == n and np.array_equal(supp3, supp[:n - 3])
        assert ctx.resident_hits() == h0 + 3


def test_order_same_image_and_eviction(ctx, oracle, forest_paths):
    import opengpc_amd as g
    W, H = 304, 75
    ctx.load_forest(forest_paths["tau"], W, H)
    rc, f = oracle.read_forest(forest_paths["tau"], W, H)
    gs, os_ = g.Settings.sparsematch(), sparsematch_settings()
    L, R = oracle.synth_pair(W, H, 5, 12)
    want = oracle.match_pair(L, R, f, os_)[0]
    # right image preprocessed first: the slots are in the other order
    pr, pl = ctx.preprocess_resident(R, 5), ctx.preprocess_resident(L, 5)
    h0 = ctx.resident_hits()
    supp, n, _ = ctx.rectified_match(pl, pr, gs)
    assert ctx.resident_hits() == h0 + 1 and np.array_equal(supp, want)
    # one image on both sides
    supp, n, _ = ctx.rectified_match(pl, pl, gs)
    assert ctx.resident_hits() == h0 + 2 and np.array_equal(supp, oracle.match_pair(L, L, f, os_)[0])
    # a third image takes the older slot (R's): (L, R) is no longer resident as a pair, (L, third) is
    T = oracle.synth_pair(W, H, 6, 3)[1]
    pt = ctx.preprocess_resident(T, 5)
    supp, n, _ = ctx.rectified_match(pl, pr, gs)
    assert ctx.resident_hits() == h0 + 2 and np.array_equal(supp, want)
    supp, n, _ = ctx.rectified_match(pl, pt, gs)
    assert ctx.resident_hits() == h0 + 3 and np.array_equal(supp, oracle.match_pair(L, T, f, os_)[0])


def test_edited_arrays_take_the_upload_path(oracle, forest_paths, monkeypatch):
    """A caller may edit a PreprocessedImage (public fields).  An edit that changes a size, touches a sampled word, or --
    with GPC_HIP_RESIDENT=2 -- changes any byte is seen, and the match then uses the arrays as given, like the reference."""
    import opengpc_amd as g
    W, H = 272, 61
    L, R = oracle.synth_pair(W, H, 3, 9)
    rc, f = oracle.read_forest(forest_paths["zero"], W, H)
    gs, os_ = g.Settings.sparsematch(), sparsematch_settings()
    for mode in ("1", "2"):
        monkeypatch.setenv("GPC_HIP_RESIDENT", mode)
        c = g.Context(0)
        try:
            c.load_forest(forest_paths["zero"], W, H)
            pl, pr = c.preprocess_resident(L, 5), c.preprocess_resident(R, 5)
            h0 = c.resident_hits()
            # shorter mask (a caller restricting the candidates): sizes differ
            short = (pl[0], pl[1], pl[2][: len(pl[2]) // 2].copy())
            supp, n, _ = c.rectified_match(short, pr, gs)
            assert c.resident_hits() == h0
            assert np.array_equal(supp, oracle_from_preprocessed(oracle, short, pr, f, W, os_, True, False)[1])
            # first candidate replaced in place (the first word of an array is always sampled)
            keep = pl[2][0]
            pl[2][0] = pl[2][1]
            supp, n, _ = c.rectified_match(pl, pr, gs)
            assert c.resident_hits() == h0
            assert np.array_equal(supp, oracle_from_preprocessed(oracle, pl, pr, f, W, os_, True, False)[1])
            pl[2][0] = keep
            supp, n, _ = c.rectified_match(pl, pr, gs)   # restored: resident again
            assert c.resident_hits() == h0 + 1
            if mode == "2":   # every byte is hashed: an edit anywhere is seen (here: one pixel of the smooth image)
                keep = pl[0][30, 101]
                pl[0][30, 101] = keep ^ 0x40
                supp, n, _ = c.rectified_match(pl, pr, gs)
                assert c.resident_hits() == h0 + 1
                assert np.array_equal(supp, oracle_from_preprocessed(oracle, pl, pr, f, W, os_, True, False)[1])
                pl[0][30, 101] = keep
        finally:
            c.close()
    monkeypatch.setenv("GPC_HIP_RESIDENT", "0")
    c = g.Context(0)
    try:
        c.load_forest(forest_paths["zero"], W, H)
        pl, pr = c.preprocess_resident(L, 5), c.preprocess_resident(R, 5)
        supp, n, _ = c.rectified_match(pl, pr, gs)
        assert c.resident_hits() == 0 and np.array_equal(supp, oracle.match_pair(L, R, f, os_)[0])
    finally:
        c.close()


def test_overwritten_and_foreign_arrays(oracle, forest_paths):
    import ctypes as C
    import opengpc_amd as g
    from opengpc_amd.capi import _ptr
    W, H = 272, 61
    rc, f = oracle.read_forest(forest_paths["zero"], W, H)
    gs, os_ = g.Settings.sparsematch(), sparsematch_settings()
    L, R = oracle.synth_pair(W, H, 3, 9)
    T = oracle.synth_pair(W, H, 8, 4)[0]
    a, b = g.Context(0), g.Context(0)
    try:
        for c in (a, b):
            c.load_forest(forest_paths["zero"], W, H)
        pl, pr = a.preprocess_resident(L, 5), a.preprocess_resident(R, 5)
        # another context delivers a different image INTO the arrays context a remembers as L's host copies
        n = C.c_int()
        assert b.L.gpc_hip_preprocess_begin(b.h, _ptr(T), W, H, 5) == 0
        st = b.L.gpc_hip_preprocess_fetch(b.h, _ptr(pl[0]), _ptr(pl[1]), _ptr(pl[2]), len(pl[2]), C.byref(n))
        assert st in (0, g.capi.E_CAPACITY)
        m = min(n.value, len(pl[2]))
        tl = (pl[0], pl[1], pl[2][:m])
        ha, hb = a.resident_hits(), b.resident_hits()
        supp, _, _ = a.rectified_match(tl, pr, gs)
        assert a.resident_hits() == ha      # context a forgot those arrays when b wrote over them
        assert np.array_equal(supp, oracle_from_preprocessed(oracle, tl, pr, f, W, os_, True, False)[1])
        # arrays of context a handed to context b: not b's host copies
        supp, _, _ = b.rectified_match(pr, pr, gs)
        assert b.resident_hits() == hb and np.array_equal(supp, oracle.match_pair(R, R, f, os_)[0])
    finally:
        a.close()
        b.close()


def test_naive_arithmetic_and_mode_switch(ctx, oracle, forest_paths):
    import opengpc_amd as g
    W, H = 272, 61
    L, R = oracle.synth_pair(W, H, 3, 9)
    ctx.load_forest(forest_paths["tau"], W, H)
    rc, f = oracle.read_forest(forest_paths["tau"], W, H)
    gs = g.Settings.sparsematch()
    ctx.set_arithmetic(True)
    pl, pr = ctx.preprocess_resident(L, 5), ctx.preprocess_resident(R, 5)
    h0 = ctx.resident_hits()
    supp, n, _ = ctx.rectified_match(pl, pr, gs)
    want = oracle.match_pair(L, R, f, sparsematch_settings(5, 128, 0, True, False, True))[0]
    assert ctx.resident_hits() == h0 + 1 and np.array_equal(supp, want)
    ctx.set_arithmetic(False)   # the resident images were made by the other arithmetic: upload path, SSE codes of the arrays as given
    supp, n, _ = ctx.rectified_match(pl, pr, gs)
    assert ctx.resident_hits() == h0 + 1


def test_warmup_every_mode_and_sizes(ctx, oracle, forest_paths):
    import opengpc_amd as g
    for W, H, forest in ((1024, 436, "zero"), (272, 61, "tau"), (1920, 1080, "tau")):
        ctx.load_forest(forest_paths[forest], W, H)
        ctx.warmup(W, H)                       # all four matcher modes
        ctx.warmup(W, H, g.Settings.sparsematch())
        assert ctx.resident_hits() == 0        # (its own calls are not counted)
        L, R = oracle.synth_pair(W, H, 1, 7)
        rc, f = oracle.read_forest(forest_paths[forest], W, H)
        supp, n, ncand, st = ctx.match_pair(L, R, g.Settings.sparsematch())
        want, nl, nr = oracle.match_pair(L, R, f, sparsematch_settings())
        assert (nl, nr) == tuple(ncand) and np.array_equal(supp, want)
    c2 = g.Context(0)
    try:
        with pytest.raises(g.capi.GpcError) as e:
            c2.warmup(1024, 436)
        assert e.value.status == g.capi.E_NO_FOREST
    finally:
        c2.close()


@pytest.mark.parametrize("epipolar,hashtable", [(1, 0), (0, 0), (1, 1), (0, 1)])
def test_two_step_forms(ctx, oracle, forest_paths, epipolar, hashtable):
    """gpc_hip_*_match_begin + gpc_hip_match_fetch (what the C++ API calls): the synchronous calls' results, short
    capacities answered with the true count and fetched again, candidate counts of the pair form."""
    import opengpc_amd as g
    W, H = 272, 61
    L, R = oracle.synth_pair(W, H, 3, 9)
    ctx.load_forest(forest_paths["tau"], W, H)
    rc, f = oracle.read_forest(forest_paths["tau"], W, H)
    gs = g.Settings(5, 64, 1, bool(epipolar), bool(hashtable), 1)
    os_ = sparsematch_settings(5, 64, 1, bool(epipolar), bool(hashtable), False)
    want, nl, nr = oracle.match_pair(L, R, f, os_)
    supp, n, st, nc = ctx.match_async("pair", L, R, gs)
    assert st == 0 and nc == (nl, nr) and n == len(want) and np.array_equal(supp, want)
    supp, n, st, nc = ctx.match_async("pair", L, R, gs, cap=max(len(want) // 2, 1))
    assert st == g.capi.E_CAPACITY and n == len(want) and np.array_equal(supp, want[:max(len(want) // 2, 1)])
    pl, pr = ctx.preprocess_resident(L, 5), ctx.preprocess_resident(R, 5)
    h0 = ctx.resident_hits()
    supp, n, st, _ = ctx.match_async("rectified", pl, pr, gs)
    assert st == 0 and np.array_equal(supp, want) and ctx.resident_hits() == h0 + 1
    corr, nc_, st, _ = ctx.match_async("stereo", pl, pr, gs)
    wcorr = oracle_from_preprocessed(oracle, pl, pr, f, W, os_, bool(epipolar), bool(hashtable))[0]
    assert st == 0 and np.array_equal(corr.view(np.int32), wcorr.view(np.int32))
    cl, cr = tuple(a.copy() for a in pl), tuple(a.copy() for a in pr)      # the upload path
    supp, n, st, _ = ctx.match_async("rectified", cl, cr, gs, cap=3)
    assert st == g.capi.E_CAPACITY and n == len(want) and np.array_equal(supp, want[:3]) and ctx.resident_hits() == h0 + 2
    # a fetch without a begin is refused
    import ctypes as C
    k = C.c_int()
    ctx.match_pair(L, R, gs)
    assert ctx.L.gpc_hip_preprocess_fetch(ctx.h, None, None, None, 0, C.byref(k)) == g.capi.E_INVALID


def test_pair_results_packed_over_the_link_equal_the_12_byte_form(oracle, forest_paths, monkeypatch):
    """The two-step forms bring an epipolar pair's supports over the link packed (4 bytes each + row counts) and expand them
    into the caller's array on the library's threads; GPC_HIP_NO_PAIR_PACKED=1 keeps the 12-byte records.  Same results, also
    with a capacity that cuts a row in two and with a fetch repeated after GPC_E_CAPACITY."""
    import opengpc_amd as g
    W, H = 1024, 436
    L, R = oracle.synth_pair(W, H, 0, 24)
    rc, f = oracle.read_forest(forest_paths["zero"], W, H)
    want, nl, nr = oracle.match_pair(L, R, f, sparsematch_settings())
    gs = g.Settings.sparsematch()
    res = {}
    for packed, workers in ((True, None), (False, None), (True, "1"), (False, "1"), (True, "3")):
        if packed:
            monkeypatch.delenv("GPC_HIP_NO_PAIR_PACKED", raising=False)
        else:
            monkeypatch.setenv("GPC_HIP_NO_PAIR_PACKED", "1")
        if workers:      # one worker thread: the jobs are still waited for; three: row ranges that do not divide evenly
            monkeypatch.setenv("GPC_HIP_EXPAND_THREADS", workers)
        else:
            monkeypatch.delenv("GPC_HIP_EXPAND_THREADS", raising=False)
        c = g.Context(0)
        try:
            c.load_forest(forest_paths["zero"], W, H)
            supp, n, st, nc = c.match_async("pair", L, R, gs)
            assert st == 0 and nc == (nl, nr) and np.array_equal(supp, want)
            cut = len(want) // 2 + 7            # ends inside a row
            supp, n, st, nc = c.match_async("pair", L, R, gs, cap=cut)      # (match_async fetches again after E_CAPACITY and compares)
            assert st == g.capi.E_CAPACITY and n == len(want) and np.array_equal(supp, want[:cut])
            pl, pr = c.preprocess_resident(L, 5), c.preprocess_resident(R, 5)
            supp, n, st, _ = c.match_async("rectified", pl, pr, gs)
            assert st == 0 and np.array_equal(supp, want)
            supp, n, st = c.rectified_match(pl, pr, gs)                     # the synchronous form, pageable `out`
            assert st == 0 and np.array_equal(supp, want)
            assert c.L.gpc_hip_host_threads(c.h) == (int(workers) if workers else c.L.gpc_hip_host_threads(c.h))
            res[(packed, workers)] = supp
        finally:
            c.close()
    for v in res.values():
        assert np.array_equal(v, res[(True, None)])


def test_two_host_threads_with_a_context_each(oracle, forest_paths):
    """The reference's Forest is stateless and re-entrant; here every host thread has a context of its own.  Two threads run
    the three-call sequence concurrently (ctypes releases the GIL: the library calls really overlap) on different pairs: the
    records of resident images are per context, the table of them is shared and locked."""
    import threading
    import opengpc_amd as g
    W, H = 304, 75
    rc, f = oracle.read_forest(forest_paths["tau"], W, H)
    gs, os_ = g.Settings.sparsematch(), sparsematch_settings()
    pairs = [oracle.synth_pair(W, H, 11, 6), oracle.synth_pair(W, H, 12, 17)]
    wants = [oracle.match_pair(L, R, f, os_)[0] for L, R in pairs]
    errors = []

    def work(k):
        try:
            c = g.Context(0)
            try:
                c.load_forest(forest_paths["tau"], W, H)
                L, R = pairs[k]
                for it in range(40):
                    pl, pr = c.preprocess_resident(L, 5), c.preprocess_resident(R, 5)
                    h0 = c.resident_hits()
                    supp, n, st = c.rectified_match(pl, pr, gs)
                    assert st == 0 and c.resident_hits() == h0 + 1 and np.array_equal(supp, wants[k]), (k, it)
                    s2, n2, st2, nc = c.match_async("pair", L, R, gs)
                    assert st2 == 0 and np.array_equal(s2, wants[k]), (k, it)
            finally:
                c.close()
        except Exception as e:     # (an assertion in a thread must fail the test)
            errors.append(e)
    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
