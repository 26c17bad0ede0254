"""GPU parity on randomly drawn shapes, images, thresholds, forests, matcher modes and arithmetic
variants (fixed seeds): whole path raw pair -> supports through the C ABI against the oracle, plus a
ragged batch of the same shape.  Complements the hand-picked shapes of test_gpu_parity.py."""
import os

import numpy as np
import pytest

from oracle.pyoracle import sparsematch_settings

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import opengpc_amd as g
    c = g.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def fused_ctx():
    """A context that takes the join + output in one launch wherever it can (by itself the library picks it only for
    launches with several rows per resident workgroup, which these small images never are)."""
    import opengpc_amd as g
    os.environ["GPC_HIP_FUSE_ALWAYS"] = "1"
    try:
        c = g.Context(0)
    finally:
        del os.environ["GPC_HIP_FUSE_ALWAYS"]
    yield c
    c.close()


def draw_pair(rng, W, H):
    kind = rng.integers(0, 4)
    wide = W + 64
    if kind == 0:    # fine noise
        base = rng.integers(0, 256, (H, wide), dtype=np.uint8)
    elif kind == 1:  # blocky texture + noise (many repeated codes)
        k = int(rng.integers(2, 7))
        base = (rng.integers(0, 256, (H // k + 1, wide // k + 1)).repeat(k, 0).repeat(k, 1)[:H, :wide] * 3 // 4
                + rng.integers(0, 64, (H, wide))).astype(np.uint8)
    elif kind == 2:  # smooth gradient + little noise (few candidates, long duplicate runs)
        yy, xx = np.mgrid[0:H, 0:wide]
        base = ((xx * 3 + yy * 2) % 256 + rng.integers(0, 8, (H, wide))).clip(0, 255).astype(np.uint8)
    else:            # sparse dots on a flat background
        base = np.full((H, wide), 60, np.uint8)
        m = rng.random((H, wide)) < 0.03
        base[m] = rng.integers(0, 256, int(m.sum()), dtype=np.uint8)
    d = int(rng.integers(0, 33))
    L = np.ascontiguousarray(base[:, 32:32 + W])
    R = np.ascontiguousarray(base[:, 32 + d - 16:32 + d - 16 + W]) if d >= 16 else np.ascontiguousarray(base[:, 32 + d:32 + d + W])
    if rng.random() < 0.3:  # a different right image altogether
        R = rng.integers(0, 256, (H, W), dtype=np.uint8)
    return L, R


# GPC_FUZZ_SEEDS=N widens the sweep for a one-off soak (the default 36 take ~2 s)
@pytest.mark.parametrize("seed", range(int(os.environ.get("GPC_FUZZ_SEEDS", "36"))))
def test_random_configuration(ctx, fused_ctx, oracle, forest_paths, seed):
    import opengpc_amd as g
    rng = np.random.default_rng(1000 + seed)
    W = 16 * int(rng.integers(3, 140))            # 48 .. 2224
    H = int(rng.integers(30, 150))
    forest = "tau" if seed % 2 else "zero"
    epipolar, hashtable = bool((seed >> 1) & 1), bool((seed >> 2) & 1)
    naive = (seed % 9) == 4
    thr = int(rng.choice([0, 3, 5, 10, 40, 181, 182, 255]))
    disp_high = int(rng.choice([0, 7, 64, 128, 4000]))
    vtol = int(rng.choice([-1, 0, 1, 3]))
    L, R = draw_pair(rng, W, H)
    rc, f = oracle.read_forest(forest_paths[forest], W, H)
    ctx.set_arithmetic(naive)
    try:
        ctx.load_forest(forest_paths[forest], W, H)
        want, nl, nr = oracle.match_pair(L, R, f, sparsematch_settings(thr, disp_high, vtol, epipolar, hashtable, naive))
        got, n, ncand, st = ctx.match_pair(L, R, g.Settings(thr, disp_high, vtol, epipolar, hashtable, 1))
        assert st == 0 and (nl, nr) == tuple(ncand), (W, H, forest, epipolar, hashtable, naive, thr)
        assert n == len(want) and np.array_equal(got, want.astype(got.dtype)), (W, H, forest, epipolar, hashtable, naive, thr)
        if epipolar and not hashtable:  # the device entry point with the fused join + output: the pair, its mirror, the pair again
            import torch
            dev = torch.device("cuda", 0)
            fused_ctx.set_arithmetic(naive)
            fused_ctx.load_forest(forest_paths[forest], W, H)
            dL = torch.from_numpy(np.stack([L, R, L])).to(dev)
            dR = torch.from_numpy(np.stack([R, L, R])).to(dev)
            capd = max(n, 1) + W
            d_out = torch.zeros((3, capd, 3), dtype=torch.int32, device=dev)
            d_cnt = torch.zeros(3, dtype=torch.int32, device=dev)
            torch.cuda.synchronize(dev)
            fused_ctx.match_batch_device(dL.data_ptr(), dR.data_ptr(), W, H, 3, g.Settings(thr, disp_high, vtol, True, False, 1),
                                         d_out.data_ptr(), capd, d_cnt.data_ptr(), 0)
            fused_ctx.synchronize()
            fused_ctx.set_arithmetic(False)
            cnt = d_cnt.cpu().numpy()
            o = d_out.cpu().numpy()
            assert cnt[0] == n and cnt[2] == n
            for q in (0, 2):
                assert np.array_equal(o[q, :n, 0], got["x"]) and np.array_equal(o[q, :n, 1], got["y"])
                assert np.array_equal(o[q, :n, 2].view(np.float32), got["d"])
        # the reference's three calls through the two-step forms the C++ API uses: preprocess x2 (the images stay on the
        # device), rectifiedMatch from the arrays as delivered (resident) or from copies of them (upload path), and the
        # whole region as one call (matchPair's form)
        gs = g.Settings(thr, disp_high, vtol, epipolar, hashtable, 1)
        pl, pr = ctx.preprocess_resident(L, thr), ctx.preprocess_resident(R, thr)
        pre = oracle.preprocess_naive if naive else oracle.preprocess
        for got_p, want_p in zip(pl + pr, pre(L, thr) + pre(R, thr)):
            assert np.array_equal(got_p, want_p), (W, H, naive, thr)
        h0 = ctx.resident_hits()
        if seed % 4 == 1:
            pl, pr = tuple(a.copy() for a in pl), tuple(a.copy() for a in pr)
        s3, n3, st3, _ = ctx.match_async("rectified", pl, pr, gs)
        assert ctx.resident_hits() == h0 + (0 if seed % 4 == 1 else 1)
        assert st3 == 0 and n3 == n and np.array_equal(s3, got), (W, H, forest, epipolar, hashtable, naive, thr)
        s4, n4, st4, nc4 = ctx.match_async("pair", L, R, gs, cap=(max(n // 2, 1) if seed % 5 == 2 else None))
        assert n4 == n and nc4 == (nl, nr) and np.array_equal(s4, got[:len(s4)]) and len(s4) == (min(n, max(n // 2, 1)) if seed % 5 == 2 else n)
        if epipolar and not hashtable and seed % 2 == 0:   # two lanes: the pair and its mirror in flight together, twice
            import torch
            dev = torch.device("cuda", 0)
            fused_ctx.set_arithmetic(naive)
            fused_ctx.load_forest(forest_paths[forest], W, H)
            fused_ctx.set_pipeline(2)
            try:
                w2, l2, r2 = oracle.match_pair(R, L, f, sparsematch_settings(thr, disp_high, vtol, True, False, naive))
                dA, dB = torch.from_numpy(np.stack([L])).to(dev), torch.from_numpy(np.stack([R])).to(dev)
                capd = max(n, len(w2), 1) + W
                outs = [torch.zeros((1, capd, 3), dtype=torch.int32, device=dev) for _ in range(4)]
                cnts = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(4)]
                torch.cuda.synchronize(dev)
                for k in range(4):
                    a, b = (dA, dB) if k % 2 == 0 else (dB, dA)
                    fused_ctx.match_batch_device(a.data_ptr(), b.data_ptr(), W, H, 1, gs, outs[k].data_ptr(), capd, cnts[k].data_ptr(), 0)
                fused_ctx.synchronize()
                for k in range(4):
                    ref = got if k % 2 == 0 else w2.astype(got.dtype)
                    o = outs[k].cpu().numpy()[0]
                    assert int(cnts[k].item()) == len(ref)
                    assert np.array_equal(o[:len(ref), 0], ref["x"]) and np.array_equal(o[:len(ref), 1], ref["y"])
                    assert np.array_equal(o[:len(ref), 2].view(np.float32), ref["d"])
            finally:
                fused_ctx.set_pipeline(1)
                fused_ctx.set_arithmetic(False)
        if seed % 3 == 0:  # the batch entry point on the same shape: pair 1 swaps the images
            out, counts, nc, st = ctx.match_batch(np.stack([L, R]), np.stack([R, L]),
                                                  g.Settings(thr, disp_high, vtol, epipolar, hashtable, 1), max(n, 1) * 2 + W * H)
            w2, l2, r2 = oracle.match_pair(R, L, f, sparsematch_settings(thr, disp_high, vtol, epipolar, hashtable, naive))
            assert counts[0] == n and np.array_equal(out[0, :n], got)
            assert counts[1] == len(w2) and np.array_equal(out[1, :counts[1]], w2.astype(got.dtype))
            assert tuple(nc[1]) == (l2, r2)
    finally:
        ctx.set_arithmetic(False)
