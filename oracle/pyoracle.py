"""ctypes front-end for the CPU oracle (oracle/libgpc_oracle.so) and, when built,
the real reference kernels (oracle/_ref/libgpc_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (opengpc_amd/) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libgpc_oracle.so")
ORACLE_FAST_SO = os.path.join(HERE, "libgpc_oracle_fast.so")
REF_SO = os.path.join(HERE, "_ref", "libgpc_ref.so")
REF_NAIVE_SO = os.path.join(HERE, "_ref", "libgpc_ref_naive.so")
MAX_TESTS = 32


class Forest(C.Structure):
    _fields_ = [
        ("offs", C.c_int32 * (2 * MAX_TESTS)),
        ("dxy", C.c_int32 * (4 * MAX_TESTS)),
        ("tau", C.c_int32 * MAX_TESTS),
        ("num_tests", C.c_int32),
        ("type", C.c_int32),
        ("discarded", C.c_int32),
        ("width", C.c_int32),
        ("height", C.c_int32),
    ]


class Settings(C.Structure):
    _fields_ = [
        ("gradient_threshold", C.c_int32),
        ("disp_high", C.c_int32),
        ("vertical_tolerance", C.c_int32),
        ("epipolar_mode", C.c_int32),
        ("use_hashtable", C.c_int32),
        ("naive", C.c_int32),
    ]


SUPPORT_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("d", "<f4")])
CORR_DTYPE = np.dtype([("sx", "<i4"), ("sy", "<i4"), ("tx", "<i4"), ("ty", "<i4")])
# training (oracle/gpc_oracle_train.h): Feature::params fields used by scoring; splitStats
SPLIT_DTYPE = np.dtype([("i", "<i4"), ("j", "<i4"), ("tau", "<i4")])
STATS_DTYPE = np.dtype([("prec", "<f8"), ("rec", "<f8"), ("hmean", "<f8"), ("convcomb", "<f8"),
                        ("tp", "<i4"), ("fp", "<i4"), ("fn", "<i4"), ("tot", "<i4")])
PATCH = 729


def build(fast=False):
    target = "libgpc_oracle_fast.so" if fast else "libgpc_oracle.so"
    subprocess.check_call(["make", "-s", "-C", HERE, target])


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _i32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _u64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


class Oracle:
    def __init__(self, fast=False):
        path = ORACLE_FAST_SO if fast else ORACLE_SO
        if not os.path.exists(path):
            build(fast)
        self.lib = L = C.CDLL(path)
        L.gpc_oracle_mix.restype = C.c_uint32
        L.gpc_oracle_mix.argtypes = [C.c_uint32, C.c_uint32]
        L.gpc_oracle_fnv1a64.restype = C.c_uint64
        L.gpc_oracle_fnv1a64.argtypes = [C.c_void_p, C.c_uint64]
        L.gpc_oracle_arr2ind.restype = C.c_int
        L.gpc_oracle_margin.restype = C.c_int
        L.gpc_oracle_parse_forest_text.restype = C.c_int
        L.gpc_oracle_read_forest.restype = C.c_int
        L.gpc_oracle_preprocess.restype = C.c_int
        L.gpc_oracle_preprocess_naive.restype = C.c_int
        L.gpc_oracle_find_correspondences.restype = C.c_int
        L.gpc_oracle_hash_correspondences.restype = C.c_int
        L.gpc_oracle_rectified_filter.restype = C.c_int
        L.gpc_oracle_match_pair.restype = C.c_int

    # ---- training scoring loop (parity unpinned, see gpc_oracle_train.h)
    def eval_split(self, triplets, marks, params, score_until_level, w1):
        """Fern::evalSplit.  triplets: (n, 3, 729) u8; marks: (n,) u8; params: SPLIT_DTYPE array."""
        t = np.ascontiguousarray(triplets, np.uint8)
        m = np.ascontiguousarray(marks, np.uint8)
        p = np.ascontiguousarray(params, SPLIT_DTYPE)
        out = np.zeros(1, STATS_DTYPE)
        self.lib.gpc_oracle_eval_split(C.c_void_p(t.ctypes.data), C.c_void_p(m.ctypes.data), C.c_int(len(t)),
                                       C.c_void_p(p.ctypes.data), C.c_int(score_until_level), C.c_double(w1),
                                       C.c_void_p(out.ctypes.data))
        return out[0]

    def mark_split_samples(self, triplets, marks, params, num_params):
        """Fern::markSplitSamples; marks is updated in place."""
        t = np.ascontiguousarray(triplets, np.uint8)
        p = np.ascontiguousarray(params, SPLIT_DTYPE)
        assert marks.dtype == np.uint8 and marks.flags.c_contiguous
        self.lib.gpc_oracle_mark_split_samples(C.c_void_p(t.ctypes.data), C.c_void_p(marks.ctypes.data),
                                               C.c_int(len(t)), C.c_void_p(p.ctypes.data), C.c_int(num_params))

    def train_fern(self, triplets, marks, max_depth, cand, num_resamples, taulo, tauhi, only_non_split, w1):
        """Fern::train with injected hyperplane samples cand[level * num_resamples + k]."""
        t = np.ascontiguousarray(triplets, np.uint8)
        c = np.ascontiguousarray(cand, SPLIT_DTYPE)
        assert len(c) >= max_depth * num_resamples
        assert marks.dtype == np.uint8 and marks.flags.c_contiguous
        fp = np.zeros(max_depth, SPLIT_DTYPE)
        st = np.zeros(max_depth, STATS_DTYPE)
        self.lib.gpc_oracle_train_fern(C.c_void_p(t.ctypes.data), C.c_void_p(marks.ctypes.data), C.c_int(len(t)),
                                       C.c_int(max_depth), C.c_void_p(c.ctypes.data), C.c_int(num_resamples),
                                       C.c_int(taulo), C.c_int(tauhi), C.c_int(int(only_non_split)), C.c_double(w1),
                                       C.c_void_p(fp.ctypes.data), C.c_void_p(st.ctypes.data))
        return fp, st

    # ---- inputs / checksums
    def synth_pair(self, W, H, s=0, D=24):
        left = np.empty((H, W), np.uint8)
        right = np.empty((H, W), np.uint8)
        self.lib.gpc_oracle_synth_pair(_u8p(left), _u8p(right), W, H, s, D)
        return left, right

    def fnv(self, arr):
        a = np.ascontiguousarray(arr)
        return int(self.lib.gpc_oracle_fnv1a64(a.ctypes.data, a.nbytes))

    # ---- kernels
    def box(self, raw):
        H, W = raw.shape
        out = np.zeros((H, W), np.uint8)
        self.lib.gpc_oracle_box(_u8p(raw), _u8p(out), W, H)
        return out

    def clear_boundary(self, buf):
        H, W = buf.shape
        self.lib.gpc_oracle_clear_boundary(_u8p(buf), W, H)
        return buf

    def sobel(self, raw, thr):
        H, W = raw.shape
        out = np.zeros((H, W), np.uint8)
        self.lib.gpc_oracle_sobel(_u8p(raw), _u8p(out), W, H, int(thr))
        return out

    def hash(self, smooth, grad, forest):
        H, W = smooth.shape
        codes = np.zeros((H, W), np.uint32)
        self.lib.gpc_oracle_hash(_u8p(smooth), _u8p(grad), _u32p(codes), C.byref(forest), W, H)
        return codes

    def preprocess(self, raw, thr):
        raw = np.ascontiguousarray(raw)
        H, W = raw.shape
        smooth = np.empty((H, W), np.uint8)
        grad = np.empty((H, W), np.uint8)
        mask = np.empty(H * W, np.int32)
        n = self.lib.gpc_oracle_preprocess(_u8p(raw), W, H, int(thr), _u8p(smooth), _u8p(grad), _i32p(mask))
        return smooth, grad, mask[:n].copy()

    # ---- the -DSSE=OFF build (*Naive kernels)
    def box_naive(self, raw):
        H, W = raw.shape
        out = np.zeros((H, W), np.uint8)
        self.lib.gpc_oracle_box_naive(_u8p(raw), _u8p(out), W, H)
        return out

    def sobel_naive(self, raw, thr):
        H, W = raw.shape
        out = np.zeros((H, W), np.uint8)
        self.lib.gpc_oracle_sobel_naive(_u8p(raw), _u8p(out), W, H, int(thr))
        return out

    def hash_naive(self, smooth, mask, forest):
        H, W = smooth.shape
        codes = np.zeros((H, W), np.uint32)
        mask = np.ascontiguousarray(mask, np.int32)
        self.lib.gpc_oracle_hash_naive(_u8p(smooth), _i32p(mask), len(mask), _u32p(codes), C.byref(forest), W, H)
        return codes

    def preprocess_naive(self, raw, thr):
        raw = np.ascontiguousarray(raw)
        H, W = raw.shape
        smooth = np.empty((H, W), np.uint8)
        grad = np.empty((H, W), np.uint8)
        mask = np.empty(H * W, np.int32)
        n = self.lib.gpc_oracle_preprocess_naive(_u8p(raw), W, H, int(thr), _u8p(smooth), _u8p(grad), _i32p(mask))
        return smooth, grad, mask[:n].copy()

    # ---- forest
    def read_forest(self, path, W, H):
        f = Forest()
        rc = self.lib.gpc_oracle_read_forest(path.encode(), W, H, C.byref(f))
        return rc, f

    def parse_forest_text(self, text, W, H):
        f = Forest()
        rc = self.lib.gpc_oracle_parse_forest_text(text.encode(), W, H, C.byref(f))
        return rc, f

    # ---- matching
    def descriptors(self, codes, mask, W, epipolar):
        st = np.empty(len(mask), np.uint64)
        codes = np.ascontiguousarray(codes)
        mask = np.ascontiguousarray(mask, np.int32)
        self.lib.gpc_oracle_descriptors(_u32p(codes), _i32p(mask), len(mask), W, int(epipolar), _u64p(st))
        return st

    def find_correspondences(self, ss, sk, ts, tk, W):
        ss = np.ascontiguousarray(ss, np.uint64)
        ts = np.ascontiguousarray(ts, np.uint64)
        sk = np.ascontiguousarray(sk, np.int32)
        tk = np.ascontiguousarray(tk, np.int32)
        out = np.empty(max(len(ss), 1), CORR_DTYPE)
        n = self.lib.gpc_oracle_find_correspondences(
            _u64p(ss), _i32p(sk), len(ss), _u64p(ts), _i32p(tk), len(ts), W, out.ctypes.data_as(C.c_void_p))
        return out[:n].copy()

    def hash_correspondences(self, ss, sk, ts, tk, W):
        ss = np.ascontiguousarray(ss, np.uint64)
        ts = np.ascontiguousarray(ts, np.uint64)
        sk = np.ascontiguousarray(sk, np.int32)
        tk = np.ascontiguousarray(tk, np.int32)
        out = np.empty(max(len(ss), len(ts), 1), CORR_DTYPE)
        n = self.lib.gpc_oracle_hash_correspondences(
            _u64p(ss), _i32p(sk), len(ss), _u64p(ts), _i32p(tk), len(ts), W, out.ctypes.data_as(C.c_void_p))
        return out[:n].copy()

    def rectified_filter(self, corr, settings):
        corr = np.ascontiguousarray(corr)
        out = np.empty(max(len(corr), 1), SUPPORT_DTYPE)
        n = self.lib.gpc_oracle_rectified_filter(
            corr.ctypes.data_as(C.c_void_p), len(corr), C.byref(settings), out.ctypes.data_as(C.c_void_p))
        return out[:n].copy()

    def match_pair(self, rawL, rawR, forest, settings):
        rawL = np.ascontiguousarray(rawL)
        rawR = np.ascontiguousarray(rawR)
        H, W = rawL.shape
        out = np.empty(H * W, SUPPORT_DTYPE)
        nl = C.c_int32()
        nr = C.c_int32()
        n = self.lib.gpc_oracle_match_pair(
            _u8p(rawL), _u8p(rawR), W, H, C.byref(forest), C.byref(settings),
            out.ctypes.data_as(C.c_void_p), C.byref(nl), C.byref(nr))
        return out[:n].copy(), nl.value, nr.value


def sparsematch_settings(thr=5, disp_high=128, vtol=0, epipolar=True, hashtable=False, naive=False):
    """Settings of samples/sparsematch.cpp:29-34."""
    return Settings(thr, disp_high, vtol, int(epipolar), int(hashtable), int(naive))


def supports_fnv(oracle, supp):
    """Appendix C convention: int32 (x, y, d) triples in output order."""
    tri = np.empty((len(supp), 3), np.int32)
    tri[:, 0] = supp["x"]
    tri[:, 1] = supp["y"]
    tri[:, 2] = supp["d"].astype(np.int32)
    return oracle.fnv(tri)


class Ref:
    """The reference's own SSE kernels (filter.hpp), when oracle/_ref is built."""

    def __init__(self, naive=False):
        path = REF_NAIVE_SO if naive else REF_SO
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = L = C.CDLL(path)
        L.gpc_ref_arr2ind.restype = C.c_int
        L.gpc_ref_is_sse.restype = C.c_int
        L.gpc_ref_cpu_baseline_pair.restype = C.c_int
        L.gpc_ref_hashmatch.restype = C.c_int

    @staticmethod
    def available(naive=False):
        return os.path.exists(REF_NAIVE_SO if naive else REF_SO)

    @staticmethod
    def _padded(img, front=64, back=64):
        """Copy `img` into a buffer with zeroed slack on both sides (the reference reads
        in[-1]) whose payload is 32-byte aligned."""
        n = img.size
        raw = np.zeros(n + front + back + 64, np.uint8)
        off = (-raw.ctypes.data - front) % 32 + front
        view = raw[off:off + n]
        view[:] = img.reshape(-1)
        return raw, view

    def box(self, img):
        H, W = img.shape
        keep, src = self._padded(img)
        out = np.zeros((H, W), np.uint8)
        self.lib.gpc_ref_box(_u8p(src), _u8p(out), W, H)
        return out

    def sobel(self, img, thr):
        H, W = img.shape
        keep, src = self._padded(img)
        out = np.zeros((H, W), np.uint8)
        self.lib.gpc_ref_sobel(_u8p(src), _u8p(out), W, H, int(thr))
        return out

    def arr2ind(self, grad):
        keep, src = self._padded(grad)
        ind = np.empty(grad.size + 64, np.int32)
        m = self.lib.gpc_ref_arr2ind(_u8p(src), grad.size, _i32p(ind))
        return ind[:m].copy()

    def hash_idx(self, smooth, grad, forest, idx):
        """gpcFilter/gpcFilterTau with the index list the non-SSE build walks (filter.hpp:237-281)."""
        H, W = smooth.shape
        keep1, s = self._padded(smooth, front=64 + 16 * W, back=64 + 16 * W)
        keep2, g = self._padded(grad)
        codes = np.zeros((H, W), np.uint32)
        offs = np.array(forest.offs[: 2 * forest.num_tests], np.int32)
        tau = np.array(forest.tau[: max(forest.num_tests, 1)], np.int32)
        idx = np.ascontiguousarray(idx, np.int32)
        self.lib.gpc_ref_hash_idx(_u8p(s), _u8p(g), _u32p(codes), _i32p(offs), _i32p(tau), forest.num_tests,
                                  forest.type, W, H, _i32p(idx), len(idx))
        return codes

    def hash(self, smooth, grad, forest, nthreads=1):
        H, W = smooth.shape
        keep1, s = self._padded(smooth, front=64 + 16 * W, back=64 + 16 * W)
        keep2, g = self._padded(grad)
        codes = np.zeros((H, W), np.uint32)
        offs = np.array(forest.offs[: 2 * forest.num_tests], np.int32)
        tau = np.array(forest.tau[: max(forest.num_tests, 1)], np.int32)
        self.lib.gpc_ref_hash(_u8p(s), _u8p(g), _u32p(codes), _i32p(offs), _i32p(tau),
                              forest.num_tests, forest.type, W, H, nthreads)
        return codes

    def cpu_baseline_pair(self, rawL, rawR, forest, settings):
        """Reference SSE kernels + C++ port of the inference.hpp glue (std::sort); see
        ref_harness.cpp.  Returns (supports as int32 [n][3], ms_preprocess, ms_match)."""
        rawL = np.ascontiguousarray(rawL)
        rawR = np.ascontiguousarray(rawR)
        H, W = rawL.shape
        offs = np.array(forest.offs[: 2 * forest.num_tests], np.int32)
        tau = np.array(forest.tau[: max(forest.num_tests, 1)], np.int32)
        out = np.empty((W * H, 3), np.int32)
        t_pre, t_match = C.c_double(), C.c_double()
        n = self.lib.gpc_ref_cpu_baseline_pair(
            _u8p(rawL), _u8p(rawR), W, H, _i32p(offs), _i32p(tau), forest.num_tests, forest.type,
            settings.gradient_threshold, settings.disp_high, settings.vertical_tolerance,
            settings.epipolar_mode, _i32p(out), W * H, C.byref(t_pre), C.byref(t_match))
        return out[:n].copy(), t_pre.value, t_match.value

    def hashmatch(self, ss, sk, ts, tk):
        """The reference's ndb::Hashmatch (hashmatch.hpp) driven as inference.hpp:204-225 does.
        Returns an int32 [n][2] array of (source k, target k)."""
        ss = np.ascontiguousarray(ss, np.uint64)
        ts = np.ascontiguousarray(ts, np.uint64)
        sk = np.ascontiguousarray(sk, np.int32)
        tk = np.ascontiguousarray(tk, np.int32)
        out = np.empty((max(len(ss), len(ts), 1), 2), np.int32)
        n = self.lib.gpc_ref_hashmatch(_u64p(ss), _i32p(sk), len(ss), _u64p(ts), _i32p(tk), len(ts), _i32p(out))
        return out[:n].copy()
