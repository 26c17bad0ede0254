"""GPU parity at BASELINE.json's configurations and across kernel variants."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle.pyoracle import sparsematch_settings, supports_fnv

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    import opengpc_amd as g
    c = g.Context(0)
    yield c
    c.close()


def gset(epipolar=True, thr=5, disp_high=128, vtol=0):
    import opengpc_amd as g
    return g.Settings(thr, disp_high, vtol, epipolar, False, 1)


def check_pair(ctx, oracle, forest, W, H, s, D, epipolar=True, disp_high=128):
    from opengpc_amd.synth import synth_pair
    L, R = synth_pair(W, H, s, D)
    rc, f = oracle.read_forest(forest, W, H)
    ctx.load_forest(forest, W, H)
    want, nl, nr = oracle.match_pair(L, R, f, sparsematch_settings(5, disp_high, 0, epipolar))
    got, n, ncand, st = ctx.match_pair(L, R, gset(epipolar, 5, disp_high, 0))
    assert st == 0 and (nl, nr) == ncand
    assert n == len(want)
    assert supports_fnv(oracle, got) == supports_fnv(oracle, want)
    assert np.array_equal(got, want.astype(got.dtype))
    return n


# every SPT instantiation of the row-join kernel (256*SPT >= W) and the ragged widths between
@pytest.mark.parametrize("W,H", [(48, 40), (256, 48), (272, 40), (512, 64), (528, 40), (1024, 60), (1040, 44),
                                  (2048, 40), (2064, 36), (3840, 36)])
def test_row_join_width_sweep(ctx, oracle, forest_paths, W, H):
    assert check_pair(ctx, oracle, forest_paths["tau"], W, H, 5, 9) > 0


def test_config3_1920x1080_tau(ctx, oracle, forest_paths):
    """BASELINE configs[2]: defaultTauForest, 1920x1080, s=1, D=40."""
    n = check_pair(ctx, oracle, forest_paths["tau"], 1920, 1080, 1, 40)
    assert n > 500000


def test_config5_4k_stress_forest(ctx, oracle):
    """BASELINE configs[4]: 16x20-test forest (first 32 tests kept, as the reference does), 3840x2160, s=2, D=64."""
    forest = os.path.join(ROOT, "forests", "stress16x20Forest.txt")
    n = check_pair(ctx, oracle, forest, 3840, 2160, 2, 64)
    assert n > 1000000


def test_config5_global_mode_small_strip(ctx, oracle):
    forest = os.path.join(ROOT, "forests", "stress16x20Forest.txt")
    check_pair(ctx, oracle, forest, 3840, 64, 2, 64, epipolar=False)


def test_disparity_filter_and_flat_images(ctx, oracle, forest_paths):
    W, H = 256, 64
    check_pair(ctx, oracle, forest_paths["zero"], W, H, 3, 30, disp_high=16)   # everything filtered or kept by |dx|
    rc, f = oracle.read_forest(forest_paths["zero"], W, H)
    ctx.load_forest(forest_paths["zero"], W, H)
    # vertical stripes: every row identical, codes heavily duplicated -> (almost) nothing is unique
    img = np.tile((np.arange(W) // 3 * 37 % 256).astype(np.uint8), (H, 1))
    for ep in (True, False):
        want, nl, nr = oracle.match_pair(img, img, f, sparsematch_settings(5, 128, 0, ep))
        got, n, ncand, st = ctx.match_pair(img, img, gset(ep))
        assert (nl, nr) == ncand and nl > 0
        assert n == len(want) and np.array_equal(got, want.astype(got.dtype))
    flat = np.full((H, W), 9, np.uint8)
    got, n, ncand, st = ctx.match_pair(flat, flat, gset(True))
    assert n == 0 and ncand == (0, 0)


def test_edge_inputs(ctx, oracle, forest_paths):
    """Empty and ragged inputs: no candidates at all, candidates on one side only, the smallest
    legal image, an odd height (box writes one more row), thresholds at both ends."""
    import opengpc_amd as g
    from opengpc_amd.synth import synth_pair
    fz = forest_paths["zero"]

    def both(L, R, thr=5, epi=True, ht=False):
        H, W = L.shape
        rc, f = oracle.read_forest(fz, W, H)
        ctx.load_forest(fz, W, H)
        want, nl, nr = oracle.match_pair(L, R, f, sparsematch_settings(thr, 128, 1, epi, ht))
        got, n, ncand, st = ctx.match_pair(L, R, g.Settings(thr, 128, 1, epi, ht, 1))
        assert st == 0 and (nl, nr) == ncand and n == len(want)
        assert np.array_equal(got, want.astype(got.dtype))
        return n, ncand

    L, R = synth_pair(48, 30, 1, 3)                      # smallest legal size: 22 x 4 candidate window
    both(L, R)
    L, R = synth_pair(160, 101, 2, 7)                    # odd height
    for epi in (True, False):
        for ht in (False, True):
            both(L, R, epi=epi, ht=ht)
    zero = np.zeros((64, 96), np.uint8)
    Ln, Rn = synth_pair(96, 64, 3, 4)
    assert both(zero, zero) == (0, (0, 0))
    for epi in (True, False):
        for ht in (False, True):
            assert both(Ln, zero, epi=epi, ht=ht)[0] == 0    # nothing on the right (n_t = 0)
            assert both(zero, Rn, epi=epi, ht=ht)[0] == 0    # nothing on the left
    both(Ln, Rn, thr=0)                                  # every pixel with any gradient is a candidate
    both(Ln, Rn, thr=181)                                # largest threshold before the int16 wrap
    both(Ln, Rn, thr=255)                                # thr^2 wraps negative: everything is a candidate


@pytest.mark.parametrize("W,H", [(4112, 34), (8192, 32), (8240, 31), (16384, 30)])
def test_very_wide_rows(ctx, oracle, forest_paths, W, H):
    """The reference takes any width that is a multiple of 16 (filter.hpp:549).  Rows beyond 4096 px run the
    join with 1024 threads x 8 / 16 pixel slots; beyond 8218 px the 16384-slot table (all the LDS a workgroup
    can have) is filled past one half.  16384 is the widest image the C ABI admits."""
    assert check_pair(ctx, oracle, forest_paths["zero"], W, H, 6, 21) > 0
    if W == 8240:
        check_pair(ctx, oracle, forest_paths["tau"], W, H, 7, 100, epipolar=False)


def test_wider_than_the_abi_admits_is_refused(ctx, forest_paths):
    import opengpc_amd as g
    st, fm = g.read_forest(forest_paths["zero"], 16400, 32)
    assert st == 0
    with pytest.raises(g.GpcError) as e:
        ctx.set_forest(fm)
    assert e.value.status == g.capi.E_UNSUPPORTED


def test_config4_batch_of_256_every_pair_and_shards(oracle, forest_paths):
    """BASELINE configs[3]: 256 pairs 1024x436 (pair i: s = i, D = 8 + i mod 64), defaultZeroForest, sparsematch
    settings, through gpc_hip_match_batch_device.  EVERY pair's candidate counts, support count and supports are
    compared with the oracle; then the eight shards `i mod 8` (what rank r of an 8-GPU job owns, opengpc_amd.dist)
    go through the one GPU one after the other and their union must equal the unsharded result."""
    import torch
    import opengpc_amd as g
    from opengpc_amd import dist as gdist
    from opengpc_amd.synth import synth_batch
    from oracle.pyoracle import Oracle
    fast = Oracle(fast=True)  # same C restatement compiled -O3 (tests/test_oracle_units.py holds it equal to the -O0 build)
    W, H, B, N = 1024, 436, 256, 8
    dev = torch.device("cuda", 0)
    rc, f = fast.read_forest(forest_paths["zero"], W, H)
    ctx = g.Context(0)
    try:
        ctx.load_forest(forest_paths["zero"], W, H)
        s = g.Settings.sparsematch()
        cap = (W - 26) * (H - 26)
        Lh, Rh = synth_batch(W, H, list(range(B)))
        d_L, d_R = torch.from_numpy(Lh).to(dev), torch.from_numpy(Rh).to(dev)
        d_out = torch.empty((B, cap, 3), dtype=torch.int32, device=dev)
        d_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
        d_nc = torch.zeros((B, 2), dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)
        ctx.match_batch_device(d_L.data_ptr(), d_R.data_ptr(), W, H, B, s, d_out.data_ptr(), cap, d_cnt.data_ptr(),
                               d_nc.data_ptr())
        ctx.synchronize()
        counts = d_cnt.cpu().numpy()
        ncand = d_nc.cpu().numpy()
        whole = []
        for i in range(B):
            want, nl, nr = fast.match_pair(Lh[i], Rh[i], f, sparsematch_settings())
            got = d_out[i, : int(counts[i])].cpu().numpy()
            assert (nl, nr) == tuple(int(v) for v in ncand[i]), i
            assert int(counts[i]) == len(want), i
            assert np.array_equal(got[:, 0], want["x"]) and np.array_equal(got[:, 1], want["y"]), i
            assert np.array_equal(got[:, 2].view(np.float32), want["d"]), i
            assert np.median(want["d"]) == 8 + i % 64, i
            whole.append(got)
        # shard composition: rank r owns pairs r, r+8, ...; all eight shards on this one GPU, one after the other
        seen = set()
        per = B // N
        for r in range(N):
            idx = gdist.shard_indices(r, N, per)
            assert all(gdist.owner_of(i, N) == r for i in idx)
            seen.update(idx)
            sel = torch.as_tensor(idx, device=dev)
            s_L, s_R = d_L[sel].contiguous(), d_R[sel].contiguous()
            s_out = torch.empty((per, cap, 3), dtype=torch.int32, device=dev)
            s_cnt = torch.zeros(per, dtype=torch.int32, device=dev)
            s_nc = torch.zeros((per, 2), dtype=torch.int32, device=dev)
            torch.cuda.synchronize(dev)
            ctx.match_batch_device(s_L.data_ptr(), s_R.data_ptr(), W, H, per, s, s_out.data_ptr(), cap,
                                   s_cnt.data_ptr(), s_nc.data_ptr())
            ctx.synchronize()
            sc, sn = s_cnt.cpu().numpy(), s_nc.cpu().numpy()
            for j, i in enumerate(idx):
                assert sc[j] == counts[i] and tuple(sn[j]) == tuple(ncand[i]), (r, i)
                assert np.array_equal(s_out[j, : int(sc[j])].cpu().numpy(), whole[i]), (r, i)
        assert seen == set(range(B))
    finally:
        ctx.close()


def test_packed_results_equal_the_12_byte_path(oracle, forest_paths):
    """gpc_hip_match_batch_device_packed + gpc_hip_expand_packed == gpc_hip_match_batch_device, and the host entry
    point gpc_hip_match_batch (which moves packed results over PCIe and expands them on worker threads) delivers the
    same records: several shapes incl. ragged widths and an odd height, both forests, a tight disparity filter,
    capacities that cut pairs short, chunk sizes that leave a ragged last chunk."""
    import torch
    import opengpc_amd as g
    from opengpc_amd.synth import synth_batch
    dev = torch.device("cuda", 0)
    ctx = g.Context(0)
    try:
        # (on the link a support takes three bytes when x and the filtered disparity fit 24 bits, else the 32-bit word:
        #  2064 px with dispHigh 4000 needs 12 + 13 bits; dispHigh 0 and 1 are the narrowest disparity fields)
        for (W, H, P, fo, disp) in [(1024, 436, 11, "zero", 128), (272, 61, 37, "tau", 128), (528, 41, 5, "tau", 9),
                                    (2064, 36, 3, "zero", 128), (48, 30, 2, "zero", 128), (2064, 36, 3, "tau", 4000),
                                    (1024, 60, 4, "zero", 4000), (272, 61, 3, "zero", 0), (272, 61, 3, "zero", 1),
                                    # 128+ pairs: chunks of 16 with the first one split 4 + 12 and the last 8 + 4 + 4 / ragged
                                    (96, 40, 131, "zero", 128), (96, 40, 128, "tau", 128)]:
            ctx.load_forest(forest_paths[fo], W, H)
            s = g.Settings(5, disp, 0, True, False, 1)
            Lh, Rh = synth_batch(W, H, [3 * i + 1 for i in range(P)])
            d_L, d_R = torch.from_numpy(Lh).to(dev), torch.from_numpy(Rh).to(dev)
            cap = (W - 26) * (H - 26)
            d_out = torch.zeros((P, cap, 3), dtype=torch.int32, device=dev)
            d_cnt = torch.zeros(P, dtype=torch.int32, device=dev)
            d_nc = torch.zeros((P, 2), dtype=torch.int32, device=dev)
            d_pk = torch.zeros((P, cap), dtype=torch.int32, device=dev)
            d_rows = torch.zeros((P, H), dtype=torch.int32, device=dev)
            d_cnt2 = torch.zeros(P, dtype=torch.int32, device=dev)
            d_nc2 = torch.zeros((P, 2), dtype=torch.int32, device=dev)
            torch.cuda.synchronize(dev)
            ctx.match_batch_device(d_L.data_ptr(), d_R.data_ptr(), W, H, P, s, d_out.data_ptr(), cap, d_cnt.data_ptr(), d_nc.data_ptr())
            ctx.match_batch_device_packed(d_L.data_ptr(), d_R.data_ptr(), W, H, P, s, d_pk.data_ptr(), cap, d_rows.data_ptr(),
                                          d_cnt2.data_ptr(), d_nc2.data_ptr())
            ctx.synchronize()
            cnt = d_cnt.cpu().numpy()
            assert np.array_equal(cnt, d_cnt2.cpu().numpy()) and torch.equal(d_nc, d_nc2) and (cnt.sum() > 0 or disp < 2)
            pk, rows = d_pk.cpu().numpy().view(np.uint32), d_rows.cpu().numpy()
            ref = []
            for i in range(P):
                want = d_out[i, : int(cnt[i])].cpu().numpy()
                assert int(rows[i, 13:H - 13].sum()) == cnt[i]
                got = g.capi.expand_packed(pk[i], rows[i], int(cnt[i]))
                assert np.array_equal(got["x"], want[:, 0]) and np.array_equal(got["y"], want[:, 1])
                assert np.array_equal(got["d"], want[:, 2].view(np.float32))
                ref.append(got)
            # the host entry point: pageable and page-locked buffers, full and short capacities
            for hcap in (cap, max(int(cnt.max()) - 7, 1), 100):
                for pinned in (False, True):
                    out = ctx.pinned_empty((P, hcap), g.SUPPORT_DTYPE) if pinned else None
                    o, c2, n2, st = ctx.match_batch(Lh, Rh, s, hcap, out=out)
                    assert np.array_equal(c2, cnt) and np.array_equal(n2, d_nc.cpu().numpy())
                    assert st == (g.capi.E_CAPACITY if (cnt > hcap).any() else 0)
                    for i in range(P):
                        k = min(int(cnt[i]), hcap)
                        assert np.array_equal(o[i, :k], ref[i][:k]), (W, H, hcap, pinned, i)
                # gpc_hip_match_batch_packed: the same pipeline with the records LEFT packed in host memory
                hp, hr, c3, n3, st = ctx.match_batch_packed(Lh, Rh, s, hcap)
                assert np.array_equal(c3, cnt) and np.array_equal(n3, d_nc.cpu().numpy())
                assert st == (g.capi.E_CAPACITY if (cnt > hcap).any() else 0)
                for i in range(P):
                    k = min(int(cnt[i]), hcap)
                    assert np.array_equal(hr[i, 13:H - 13], rows[i, 13:H - 13]) and hr[i, :13].sum() == 0 and hr[i, H - 13:].sum() == 0
                    assert np.array_equal(hp[i, :k], pk[i, :k]), (W, H, hcap, i)
                    assert np.array_equal(g.capi.expand_packed(hp[i], hr[i], k), ref[i][:k])
            with pytest.raises(g.GpcError):   # rows are the unit of the packed format: the other matcher modes refuse it
                ctx.match_batch_packed(Lh, Rh, g.Settings(5, disp, 0, False, False, 1), cap)
    finally:
        ctx.close()


def test_single_pair_host_path_with_page_locked_images(forest_paths):
    """One or two pairs with a page-locked result array take the direct path of gpc_hip_match_batch: page-locked,
    16-byte aligned IMAGES are fetched by one kernel launch (k_upload2), anything else by hipMemcpyAsync; GPC_HIP_UPLOAD=2
    lets the preprocess kernel read the host's pages itself.  All of them against the device-resident entry point:
    aligned and deliberately misaligned page-locked images, pageable ones, both knob settings."""
    import torch
    import opengpc_amd as g
    from opengpc_amd.synth import synth_batch
    dev = torch.device("cuda", 0)
    ctxs = []
    try:
        for mode in ("1", "2", "0"):
            os.environ["GPC_HIP_UPLOAD"] = mode
            try:
                ctxs.append(g.Context(0))
            finally:
                del os.environ["GPC_HIP_UPLOAD"]
        for (W, H, P, fo) in [(1024, 436, 1, "zero"), (272, 61, 2, "tau"), (48, 30, 1, "zero"), (528, 41, 2, "zero")]:
            s = g.Settings.sparsematch()
            Lh, Rh = synth_batch(W, H, [5 * i + 2 for i in range(P)])
            cap = (W - 26) * (H - 26)
            want = None
            for ctx in ctxs:
                ctx.load_forest(forest_paths[fo], W, H)
                if want is None:
                    d_L, d_R = torch.from_numpy(Lh).to(dev), torch.from_numpy(Rh).to(dev)
                    d_out = torch.zeros((P, cap, 3), dtype=torch.int32, device=dev)
                    d_cnt = torch.zeros(P, dtype=torch.int32, device=dev)
                    d_nc = torch.zeros((P, 2), dtype=torch.int32, device=dev)
                    ctx.match_batch_device(d_L.data_ptr(), d_R.data_ptr(), W, H, P, s, d_out.data_ptr(), cap, d_cnt.data_ptr(), d_nc.data_ptr())
                    ctx.synchronize()
                    want = (d_cnt.cpu().numpy(), d_nc.cpu().numpy(), d_out.cpu().numpy())
                    assert want[0].sum() > 0
                n = Lh.size
                flatL, flatR = ctx.pinned_empty((n + 16,), np.uint8), ctx.pinned_empty((n + 16,), np.uint8)
                for off in (0, 1):  # offset 1: a page-locked image the 16-byte loads of the upload kernel cannot take
                    Lp, Rp = flatL[off:off + n].reshape(Lh.shape), flatR[off:off + n].reshape(Rh.shape)
                    Lp[:] = Lh
                    Rp[:] = Rh
                    for (a, b) in ((Lp, Rp), (Lp, Rh), (Lh, Rh)):
                        out = ctx.pinned_empty((P, cap), g.SUPPORT_DTYPE)
                        o, c2, n2, st = ctx.match_batch(a, b, s, cap, out=out)
                        assert st == 0 and np.array_equal(c2, want[0]) and np.array_equal(n2, want[1])
                        for i in range(P):
                            k = int(c2[i])
                            assert np.array_equal(o[i, :k]["x"], want[2][i, :k, 0]) and np.array_equal(o[i, :k]["y"], want[2][i, :k, 1])
                            assert np.array_equal(o[i, :k]["d"], want[2][i, :k, 2].view(np.float32)), (W, H, off, i)
    finally:
        for ctx in ctxs:
            ctx.close()


def test_packed_results_are_refused_outside_the_epipolar_sort_matcher(ctx, forest_paths):
    import torch
    import opengpc_amd as g
    W, H = 96, 64
    ctx.load_forest(forest_paths["zero"], W, H)
    z = torch.zeros((1, H, W), dtype=torch.uint8, device="cuda:0")
    buf = torch.zeros((1, 4096), dtype=torch.int32, device="cuda:0")
    for s in (g.Settings(5, 128, 0, False, False, 1), g.Settings(5, 128, 0, True, True, 1)):
        with pytest.raises(g.GpcError) as e:
            ctx.match_batch_device_packed(z.data_ptr(), z.data_ptr(), W, H, 1, s, buf.data_ptr(), 4096, buf.data_ptr(),
                                          buf.data_ptr(), 0)
        assert e.value.status == g.capi.E_UNSUPPORTED


def test_partitioned_and_radix_sort_matchers_agree(oracle, forest_paths):
    """The non-epipolar sort-matcher has two implementations: partition into code ranges + LDS join (k_partition.h)
    and, as the fallback when a partition is over-full, the device-wide radix sort (k_global.h).
    GPC_HIP_NO_PARTITION forces the second.  Both against the oracle: a textured pair (partitions fit), a batch, and
    striped images whose few codes put thousands of records into one partition (the fallback is taken by itself)."""
    import opengpc_amd as g
    from opengpc_amd.synth import synth_pair
    os.environ["GPC_HIP_NO_PARTITION"] = "1"
    try:
        radix = g.Context(0)
    finally:
        del os.environ["GPC_HIP_NO_PARTITION"]
    part = g.Context(0)
    try:
        cases = []
        for (W, H, s_, D, fo) in [(1024, 436, 3, 21, "zero"), (272, 61, 4, 9, "tau"), (2064, 40, 1, 30, "zero")]:
            cases.append((fo,) + synth_pair(W, H, s_, D))
        stripes = np.tile((np.arange(1024) // 3 * 37 % 256).astype(np.uint8), (200, 1))
        cases.append(("zero", stripes, np.roll(stripes, 7, axis=1)))
        for fo, L, R in cases:
            H, W = L.shape
            rc, f = oracle.read_forest(forest_paths[fo], W, H)
            for ctx in (part, radix):
                ctx.load_forest(forest_paths[fo], W, H)
            for disp, vtol in ((128, 1), (6, 0)):
                want, nl, nr = oracle.match_pair(L, R, f, sparsematch_settings(5, disp, vtol, False))
                for ctx in (part, radix):
                    got, n, ncand, st = ctx.match_pair(L, R, g.Settings(5, disp, vtol, False, False, 1))
                    assert st == 0 and (nl, nr) == ncand and n == len(want)
                    assert np.array_equal(got, want.astype(got.dtype))
        # a batch through both
        Ls, Rs = zip(*[synth_pair(528, 90, 20 + i, 4 + 2 * i) for i in range(5)])
        for ctx in (part, radix):
            ctx.load_forest(forest_paths["tau"], 528, 90)
        sset = g.Settings(5, 128, 1, False, False, 1)
        a = part.match_batch(np.stack(Ls), np.stack(Rs), sset, 40000)
        b = radix.match_batch(np.stack(Ls), np.stack(Rs), sset, 40000)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[1].sum() > 0
        for i in range(5):
            assert np.array_equal(a[0][i, : a[1][i]], b[0][i, : b[1][i]])
        # the hash-table matcher has the same pair of implementations (k_htjoin.h: bins of 1024 buckets; k_hashtable.h:
        # radix sort by bucket id).  The stripes put more records into one bin than a workgroup holds: fallback.
        for fo, L, R in cases:
            H, W = L.shape
            rc, f = oracle.read_forest(forest_paths[fo], W, H)
            for ctx in (part, radix):
                ctx.load_forest(forest_paths[fo], W, H)
            for epi, disp, vtol in ((True, 128, 1), (False, 128, 1), (True, 6, 0)):
                want, nl, nr = oracle.match_pair(L, R, f, sparsematch_settings(5, disp, vtol, epi, True))
                for ctx in (part, radix):
                    got, n, ncand, st = ctx.match_pair(L, R, g.Settings(5, disp, vtol, epi, True, 1))
                    assert st == 0 and (nl, nr) == ncand and n == len(want)
                    assert np.array_equal(got, want.astype(got.dtype))
        for ctx in (part, radix):
            ctx.load_forest(forest_paths["tau"], 528, 90)
        hset = g.Settings(5, 128, 1, True, True, 1)
        a = part.match_batch(np.stack(Ls), np.stack(Rs), hset, 40000)
        b = radix.match_batch(np.stack(Ls), np.stack(Rs), hset, 40000)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[1].sum() > 0
        for i in range(5):
            assert np.array_equal(a[0][i, : a[1][i]], b[0][i, : b[1][i]])
    finally:
        part.close()
        radix.close()


def test_forests_whose_top_tests_never_hold(ctx, oracle):
    """Tests that compare a pixel with itself never set their bit: the codes' top bits are then constant and every match of
    a row falls into the same top-bit range.  The join divides the range the row's matched codes really span (not the
    forest's nominal code width) into its rank buckets, so such forests neither lose order nor walk long buckets."""
    import opengpc_amd as g
    from opengpc_amd.synth import synth_pair
    rng = np.random.default_rng(99)
    for live, dead_first in ((18, False), (11, False), (20, True)):
        lines = ["6"]
        t = 0
        for fern in range(6):
            lines.append("%d l 5" % fern)
            for k in range(5):
                dead = (t < 30 - live) if dead_first else (t >= live)
                ix, iy, jx, jy = rng.integers(-13, 14, 4)
                if dead:
                    jx, jy = ix, iy
                lines.append("%d %d %d %d %d 0" % (k, ix, iy, jx, jy))
                t += 1
        text = "\n".join(lines)
        for (W, H, s_, D) in [(1024, 80, 2, 14), (272, 61, 5, 6)]:
            L, R = synth_pair(W, H, s_, D)
            st, fm = g.parse_forest(text, W, H)
            rc, f = oracle.parse_forest_text(text, W, H)
            ctx.set_forest(fm)
            for epi in (True, False):
                want, nl, nr = oracle.match_pair(L, R, f, sparsematch_settings(5, 128, 1, epi))
                got, n, ncand, st = ctx.match_pair(L, R, gset(epi, 5, 128, 1))
                assert (nl, nr) == ncand and n == len(want)
                assert np.array_equal(got, want.astype(got.dtype))


def test_very_tall_image_in_the_device_wide_modes(ctx, oracle, forest_paths):
    """The partitioned hash-table matcher packs a position as y << 14 | x and hands images of 2^17 rows or more to the
    radix-sort matcher (k_hashtable.h); the row of a 64-bit state `y << 32 | code` stays exact either way."""
    import opengpc_amd as g
    W, H = 48, (1 << 17) + 40
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, (H, W + 8), dtype=np.uint8)
    L, R = np.ascontiguousarray(base[:, 4:4 + W]), np.ascontiguousarray(base[:, 1:1 + W])
    rc, f = oracle.read_forest(forest_paths["zero"], W, H)
    ctx.load_forest(forest_paths["zero"], W, H)
    for epi, ht in ((True, True), (False, True), (False, False), (True, False)):
        want, nl, nr = oracle.match_pair(L, R, f, sparsematch_settings(5, 128, 1, epi, ht))
        got, n, ncand, st = ctx.match_pair(L, R, g.Settings(5, 128, 1, epi, ht, 1))
        assert st == 0 and (nl, nr) == ncand and n == len(want) and n > 0
        assert np.array_equal(got, want.astype(got.dtype))


def test_large_images_in_the_device_wide_modes(ctx, forest_paths):
    """1280x720 and 1920x1080 (BASELINE configs[2]'s size) through the non-epipolar sort-matcher and the hash-table
    matcher.  The partition kernels take 512 / 1024 code-range bins and bins of 256 buckets there, so that a bin still
    fits one workgroup (720p: both partitioned matchers run; 1080p with the Tau forest: one code range holds more than a
    workgroup takes and the radix path runs instead) -- the results are the oracle's either way."""
    import opengpc_amd as g
    from opengpc_amd.synth import synth_pair
    from oracle.pyoracle import Oracle
    fast = Oracle(fast=True)
    for (W, H, s_, D, fo) in ((1280, 720, 3, 30, "zero"), (1920, 1080, 1, 40, "tau")):
        L, R = synth_pair(W, H, s_, D)
        rc, f = fast.read_forest(forest_paths[fo], W, H)
        ctx.load_forest(forest_paths[fo], W, H)
        for epi, ht in ((False, False), (True, True), (False, True)):
            want, nl, nr = fast.match_pair(L, R, f, sparsematch_settings(5, 128, 1, epi, ht))
            got, n, ncand, st = ctx.match_pair(L, R, g.Settings(5, 128, 1, epi, ht, 1))
            assert st == 0 and (nl, nr) == ncand and n == len(want) and n > 100000
            assert np.array_equal(got, want.astype(got.dtype))


def test_hash_table_lists_that_fill_up_per_record_and_per_wave(forest_paths):
    """ndb::Hashmatch keeps the first ten elements of a bucket.  k_ht_join handles a bucket with more records either with
    one wave per bucket or -- where such buckets are the rule, by the batch's records per bucket -- with one thread per
    record (HtjArgs::mid).  GPC_HIP_HT_MID forces the threshold: 10 (always a wave), 12 and 32 (per record up to there) on
    a textured 1280x720 pair (8.5 records per bucket), a small pair and striped images whose few states put hundreds of
    records into one bucket; all against the oracle, both row conventions."""
    import opengpc_amd as g
    from opengpc_amd.synth import synth_pair
    from oracle.pyoracle import Oracle
    fast = Oracle(fast=True)
    ctxs = []
    try:
        for mid in ("10", "12", "32"):
            os.environ["GPC_HIP_HT_MID"] = mid
            try:
                ctxs.append(g.Context(0))
            finally:
                del os.environ["GPC_HIP_HT_MID"]
        cases = [("tau",) + synth_pair(1280, 720, 2, 25), ("zero",) + synth_pair(528, 200, 4, 12)]
        stripes = np.tile((np.arange(1024) // 5 * 53 % 256).astype(np.uint8), (300, 1))
        stripes[::7] = np.roll(stripes[::7], 3, axis=1)
        cases.append(("zero", stripes, np.roll(stripes, 9, axis=1)))
        for fo, L, R in cases:
            H, W = L.shape
            rc, f = fast.read_forest(forest_paths[fo], W, H)
            for epi in (True, False):
                want, nl, nr = fast.match_pair(L, R, f, sparsematch_settings(5, 128, 1, epi, True))
                for ctx in ctxs:
                    ctx.load_forest(forest_paths[fo], W, H)
                    got, n, ncand, st = ctx.match_pair(L, R, g.Settings(5, 128, 1, epi, True, 1))
                    assert st == 0 and (nl, nr) == ncand and n == len(want)
                    assert np.array_equal(got, want.astype(got.dtype)), (fo, W, H, epi)
    finally:
        for ctx in ctxs:
            ctx.close()


def test_4k_pair_in_the_device_wide_modes(ctx, forest_paths):
    """3840x2160 (BASELINE configs[4]'s size) with epipolarMode off and with the hash table: beyond ~6.5 M pixels the
    partition plan used to ask for more LDS than a workgroup has (8 bytes per POSSIBLE partition) and the call failed
    with a HIP error although the radix path serves such images.  The plan is bounded by its bin count now; whichever
    path runs, the supports are the oracle's."""
    import opengpc_amd as g
    from opengpc_amd.synth import synth_pair
    from oracle.pyoracle import Oracle
    fast = Oracle(fast=True)
    W, H = 3840, 2160
    L, R = synth_pair(W, H, 2, 64)
    rc, f = fast.read_forest(forest_paths["zero"], W, H)
    ctx.load_forest(forest_paths["zero"], W, H)
    for epi, ht in ((False, False), (False, True)):
        want, nl, nr = fast.match_pair(L, R, f, sparsematch_settings(5, 128, 1, epi, ht))
        got, n, ncand, st = ctx.match_pair(L, R, g.Settings(5, 128, 1, epi, ht, 1))
        # (the reference's 214673-bucket table is far over-full at this size: 10 records per bucket kept of ~53)
        assert st == 0 and (nl, nr) == ncand and n == len(want) and n > (100 if ht else 100000)
        assert np.array_equal(got, want.astype(got.dtype))


def test_fused_join_equals_the_two_launch_path(oracle, forest_paths):
    """The join that writes the supports itself (k_row_join<..., FUSE>: rows drawn from ticket counters, the rows'
    places settled by a look-back between workgroups, 12-byte records / packed words / correspondences written in place)
    against join + k_gather_rows as two launches (GPC_HIP_NO_FUSE) and the oracle: batches whose rows are in flight in
    different numbers per pair (1, 3, 40 pairs), widths of every threads-per-row instantiation, a capacity that cuts
    the output in the middle of a row, repeated launches on one context (ticket counters and launch epochs carry
    over), the packed device entry point and stereoMatch's correspondences."""
    import torch
    import opengpc_amd as g
    from opengpc_amd.synth import synth_batch, synth_pair
    from oracle.pyoracle import Oracle
    fast = Oracle(fast=True)
    dev = torch.device("cuda", 0)
    os.environ["GPC_HIP_NO_FUSE"] = "1"
    try:
        two = g.Context(0)
    finally:
        del os.environ["GPC_HIP_NO_FUSE"]
    os.environ["GPC_HIP_FUSE_ALWAYS"] = "1"  # (by itself the library fuses only launches with several rows per resident workgroup)
    try:
        one = g.Context(0)
    finally:
        del os.environ["GPC_HIP_FUSE_ALWAYS"]
    try:
        s = g.Settings.sparsematch()
        for (W, H, B) in ((1024, 436, 3), (1024, 120, 40), (528, 64, 5), (2064, 60, 2), (3840, 48, 1), (96, 40, 7)):
            Lh, Rh = synth_batch(W, H, list(range(B)))
            rc, f = fast.read_forest(forest_paths["tau"], W, H)
            cap_full = (W - 26) * (H - 26)
            want = [fast.match_pair(Lh[i], Rh[i], f, sparsematch_settings()) for i in range(B)]
            d_L, d_R = torch.from_numpy(Lh).to(dev), torch.from_numpy(Rh).to(dev)
            for cap in (cap_full, max(1, len(want[0][0]) // 2 + 3)):
                outs = []
                for ctx in (one, two):
                    ctx.load_forest(forest_paths["tau"], W, H)
                    d_out = torch.zeros((B, cap, 3), dtype=torch.int32, device=dev)
                    d_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
                    d_nc = torch.zeros((B, 2), dtype=torch.int32, device=dev)
                    torch.cuda.synchronize(dev)
                    for _ in range(3):  # the same launch again and again: counters and epochs carry over
                        ctx.match_batch_device(d_L.data_ptr(), d_R.data_ptr(), W, H, B, s, d_out.data_ptr(), cap, d_cnt.data_ptr(),
                                               d_nc.data_ptr())
                    ctx.synchronize()
                    outs.append((d_out.cpu().numpy(), d_cnt.cpu().numpy(), d_nc.cpu().numpy()))
                (o1, c1, n1), (o2, c2, n2) = outs
                assert np.array_equal(c1, c2) and np.array_equal(n1, n2) and np.array_equal(o1, o2)
                for i in range(B):
                    w, nl, nr = want[i]
                    assert c1[i] == len(w) and tuple(n1[i]) == (nl, nr)
                    k = min(cap, len(w))
                    assert np.array_equal(o1[i, :k, 0], w["x"][:k]) and np.array_equal(o1[i, :k, 1], w["y"][:k])
                    assert np.array_equal(o1[i, :k, 2].view(np.float32), w["d"][:k])
            # packed words + row counts left in HBM (the fused join writes them with the pair's own stride)
            d_pk = torch.zeros((B, cap_full), dtype=torch.int32, device=dev)
            d_rows = torch.zeros((B, H), dtype=torch.int32, device=dev)
            d_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
            torch.cuda.synchronize(dev)
            one.load_forest(forest_paths["tau"], W, H)
            one.match_batch_device_packed(d_L.data_ptr(), d_R.data_ptr(), W, H, B, s, d_pk.data_ptr(), cap_full, d_rows.data_ptr(),
                                          d_cnt.data_ptr())
            one.synchronize()
            pk, rows, cnt = d_pk.cpu().numpy().view(np.uint32), d_rows.cpu().numpy(), d_cnt.cpu().numpy()
            for i in range(B):
                w = want[i][0]
                assert cnt[i] == len(w) and rows[i, 13:H - 13].sum() == len(w)
                got = g.capi.expand_packed(pk[i], rows[i], len(w))
                assert np.array_equal(got, w.astype(got.dtype))
        # stereoMatch (correspondences, no disparity filter) goes through the same join in its third output mode
        W, H = 272, 61
        L, R = synth_pair(W, H, 4, 9)
        rc, f = oracle.read_forest(forest_paths["zero"], W, H)
        res = []
        for ctx in (one, two):
            ctx.load_forest(forest_paths["zero"], W, H)
            res.append(ctx.stereo_match(ctx.preprocess(L, 5), ctx.preprocess(R, 5), g.Settings(5, 128, 0, True, False, 1)))
        assert res[0][1] == res[1][1] and res[0][1] > 0 and np.array_equal(res[0][0], res[1][0])
    finally:
        one.close()
        two.close()


def test_two_lane_pipeline_equals_the_strict_form(forest_paths):
    """gpc_hip_set_pipeline(2): consecutive device-resident batches alternate between two lanes, batch k+1's preprocess and
    hash kernels run beside batch k's join.  Six batches of different pairs: every support of every pair equals the strict
    (one stream, call by call) result; a caller's own stream orders the results through gpc_hip_pipeline_join."""
    import torch
    import opengpc_amd as g
    from opengpc_amd.synth import synth_batch
    W, H, B, NB = 1024, 436, 24, 6
    dev = torch.device("cuda", 0)
    cap = 300000
    s = g.Settings.sparsematch()
    batches = []
    for k in range(NB):
        L, R = synth_batch(W, H, [100 * k + j for j in range(B)])
        batches.append((torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)))

    def run(lanes, own_stream):
        c = g.Context(0)
        try:
            st = torch.cuda.Stream(device=dev) if own_stream else None
            if st is not None:
                c.set_stream(st.cuda_stream)
            c.load_forest(forest_paths["zero"], W, H)
            c.set_pipeline(lanes)
            outs = [(torch.zeros((B, cap, 3), dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev),
                     torch.zeros((B, 2), dtype=torch.int32, device=dev)) for _ in range(NB)]
            for rep in range(3):       # (the lanes' first calls allocate: later rounds run with everything in place)
                for k, (dl, dr) in enumerate(batches):
                    o, n, nc = outs[k]
                    c.match_batch_device(dl.data_ptr(), dr.data_ptr(), W, H, B, s, o.data_ptr(), cap, n.data_ptr(), nc.data_ptr())
            if st is not None:
                c.pipeline_join()      # the caller's stream now waits for every call queued so far
                st.synchronize()
                assert c.L.gpc_hip_synchronize(c.h) == 0
            else:
                c.synchronize()
            return [(o.cpu().numpy(), n.cpu().numpy(), nc.cpu().numpy()) for o, n, nc in outs]
        finally:
            c.close()

    want = run(1, False)
    for own in (False, True):
        got = run(2, own)
        for (wo, wn, wc), (go, gn, gc) in zip(want, got):
            assert np.array_equal(wn, gn) and np.array_equal(wc, gc) and wn.min() > 1000
            assert np.array_equal(wo, go)        # (zero-filled arrays: only valid records are ever written)
