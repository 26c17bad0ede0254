"""CPU: the C-ABI library loads and exports every symbol include/gpc_hip.h declares; host-only
entry points (forest parsing, status strings) work without a GPU; compute entry points fail
loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from opengpc_amd import build
    build.build()
    import opengpc_amd as g
    return g.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gpc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gpc_hip_\w+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    import opengpc_amd.capi as capi
    names = declared_symbols()
    assert len(names) >= 24
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(capi.SYMBOLS) == names
    assert lib.gpc_hip_abi_version() == 1


def test_struct_layouts_match_the_header():
    import opengpc_amd as g
    assert g.SUPPORT_DTYPE.itemsize == 12 and g.CORR_DTYPE.itemsize == 16
    assert C.sizeof(g.Settings) == 24
    assert C.sizeof(g.FilterMask) == 4 * (64 + 32 + 5)


def test_host_only_entry_points(lib, forest_paths, oracle):
    import opengpc_amd as g
    for name, path in forest_paths.items():
        st, fm = g.read_forest(path, 1024, 436)
        rc, f = oracle.read_forest(path, 1024, 436)
        assert st == 0 and fm.num_tests == f.num_tests == 30 and fm.type == f.type
        assert list(fm.mask[:60]) == list(f.offs[:60]) and list(fm.tau[:30]) == list(f.tau[:30])
    st, fm = g.read_forest("/nonexistent.txt", 96, 64)
    assert st == g.capi.E_IO and fm.num_tests == 0 and fm.type == 0
    st, fm = g.read_forest(os.path.join(ROOT, "forests", "stress16x20Forest.txt"), 3840, 2160)
    assert (st, fm.num_tests, fm.discarded, fm.type) == (0, 32, 288, 1)
    assert lib.gpc_hip_status_string(4).decode() == "output capacity too small"
    names = [lib.gpc_hip_kernel_name(i).decode() for i in range(lib.gpc_hip_kernel_count())]
    assert "k_hash" in names and "k_row_join" in names


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import opengpc_amd as g
    with pytest.raises(g.GpcError) as e:
        g.Context(0)
    assert e.value.status == g.capi.E_NO_DEVICE


def test_product_never_imports_the_oracle():
    for root in ("opengpc_amd", "include", "samples"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, root)):
            for fn in files:
                if fn.endswith((".py", ".h", ".hpp", ".hip", ".cpp")):
                    text = open(os.path.join(dirpath, fn), errors="ignore").read()
                    assert "pyoracle" not in text and "gpc_oracle" not in text and "libgpc_ref" not in text, fn


def test_expand_packed_is_host_only_and_exact(lib):
    """gpc_hip_expand_packed: x | xR << 16 words + per-row counts -> ndb::Support records {x, y, float(x - xR)}
    (inference.hpp:384-391).  Host code, so it runs without a GPU; every alignment of the output and every
    tail length of the 4-record streaming-store groups is exercised."""
    import opengpc_amd as g
    rng = np.random.default_rng(5)
    H = 61
    for trial in range(30):
        rows = np.zeros(H, np.int32)
        rows[13:H - 13] = rng.integers(0, 9, H - 26) if trial % 3 else rng.integers(0, 300, H - 26)
        rows[:13] = -7            # never read
        n_all = int(rows[13:H - 13].sum())
        xl = rng.integers(13, 4000, n_all).astype(np.uint32)
        xr = rng.integers(0, 4096, n_all).astype(np.uint32)
        packed = xl | (xr << 16)
        y = np.repeat(np.arange(13, H - 13), rows[13:H - 13])
        for n in sorted({n_all, max(n_all - 5, 0), n_all // 2, min(3, n_all), 0}):
            got = g.capi.expand_packed(packed, rows, n)
            assert len(got) == n
            assert np.array_equal(got["x"], xl[:n].astype(np.int32)) and np.array_equal(got["y"], y[:n])
            assert np.array_equal(got["d"], (xl[:n].astype(np.int64) - xr[:n].astype(np.int64)).astype(np.float32))
