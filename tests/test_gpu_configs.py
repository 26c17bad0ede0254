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


def test_row_kernel_generations_agree(oracle, forest_paths):
    """GPC_HIP_ROWMATCH selects the row kernel: join (default), bucket, lds -- identical supports."""
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import opengpc_amd as g
from opengpc_amd.synth import synth_pair
ctx = g.Context(0)
out = []
for (W, H, fo) in [(1024, 436, "defaultZeroForest.txt"), (272, 40, "defaultTauForest.txt")]:
    ctx.load_forest(%r + "/forests/" + fo, W, H)
    L, R = synth_pair(W, H, 4, 17)
    supp, n, nc, st = ctx.match_pair(L, R, g.Settings.sparsematch())
    out.append((n, nc, int(np.frombuffer(supp.tobytes(), np.uint8).astype(np.uint64).sum()), supp.tobytes()[:64].hex()))
print(out)
''' % (ROOT, ROOT)
    res = []
    for mode in ("", "bucket", "lds"):
        env = dict(os.environ, GPC_HIP_ROWMATCH=mode)
        res.append(subprocess.run([sys.executable, "-c", code], env=env, check=True, capture_output=True,
                                  text=True).stdout.strip())
    assert res[0] == res[1] == res[2] and res[0].startswith("[(")


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


def test_very_wide_image_falls_back_to_lds_sort_kernel(ctx, oracle, forest_paths):
    """W > 4096 exceeds the join kernel's instantiations; the LDS-sort row kernel takes over."""
    check_pair(ctx, oracle, forest_paths["zero"], 4112, 34, 6, 21)
