"""ctypes binding of the C ABI in include/gpc_hip.h (libgpc_hip.so).

Plumbing only: numpy for host buffers, raw integer device pointers for HBM-resident
batches (typically torch tensors' data_ptr()).  There is no CPU fallback -- if the
shared library is missing or no gfx950 device is usable, this raises.
"""
import atexit
import ctypes as C
import os
import sys
import weakref

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# GPC_HIP_LIB overrides the library path (used only to A/B differently compiled builds of this library)
LIB_PATH = os.environ.get("GPC_HIP_LIB") or os.path.join(HERE, "libgpc_hip.so")

MAX_TESTS = 32
OK, E_INVALID, E_NO_DEVICE, E_HIP, E_CAPACITY, E_NO_FOREST, E_FOREST_RANGE, E_IO, E_UNSUPPORTED = range(9)

SUPPORT_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("d", "<f4")])
CORR_DTYPE = np.dtype([("src_x", "<i4"), ("src_y", "<i4"), ("tar_x", "<i4"), ("tar_y", "<i4")])


class Settings(C.Structure):
    """gpc::inference::InferenceSettings (reference inference.hpp:71-131), same defaults."""
    _fields_ = [
        ("gradient_threshold", C.c_int32),
        ("disp_high", C.c_int32),
        ("vertical_tolerance", C.c_int32),
        ("epipolar_mode", C.c_int32),
        ("use_hashtable", C.c_int32),
        ("num_threads", C.c_int32),
    ]

    def __init__(self, gradient_threshold=10, disp_high=128, vertical_tolerance=1,
                 epipolar_mode=False, use_hashtable=False, num_threads=1):
        super().__init__(int(gradient_threshold), int(disp_high), int(vertical_tolerance),
                         int(bool(epipolar_mode)), int(bool(use_hashtable)), int(num_threads))

    @classmethod
    def sparsematch(cls):
        """The settings of samples/sparsematch.cpp:29-34."""
        return cls(5, 128, 0, True, False, 1)


class FilterMask(C.Structure):
    """gpc::inference::Forest::FilterMask (reference inference.hpp:137-156)."""
    _fields_ = [
        ("mask", C.c_int32 * (2 * MAX_TESTS)),
        ("tau", C.c_int32 * MAX_TESTS),
        ("num_tests", C.c_int32),
        ("type", C.c_int32),
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("discarded", C.c_int32),
    ]


class GpcError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("gpc_hip status %d: %s" % (status, msg))
        self.status = status


_lib = None

# every symbol include/gpc_hip.h declares (tests check the library exports all of them)
SYMBOLS = [
    "gpc_hip_abi_version", "gpc_hip_status_string", "gpc_hip_device_count", "gpc_hip_create",
    "gpc_hip_destroy", "gpc_hip_last_error", "gpc_hip_set_stream", "gpc_hip_synchronize",
    "gpc_hip_reserve", "gpc_hip_set_arithmetic", "gpc_hip_host_alloc", "gpc_hip_host_free", "gpc_hip_read_forest", "gpc_hip_parse_forest", "gpc_hip_set_forest",
    "gpc_hip_warmup", "gpc_hip_preprocess", "gpc_hip_preprocess_begin", "gpc_hip_preprocess_fetch", "gpc_hip_resident_hits",
    "gpc_hip_rectified_match_begin", "gpc_hip_stereo_match_begin", "gpc_hip_match_pair_begin", "gpc_hip_match_fetch",
    "gpc_hip_hash_codes", "gpc_hip_rectified_match", "gpc_hip_stereo_match",
    "gpc_hip_match_pair", "gpc_hip_match_batch_device", "gpc_hip_set_pipeline", "gpc_hip_pipeline_join", "gpc_hip_match_batch",
    "gpc_hip_match_batch_device_packed", "gpc_hip_match_batch_packed", "gpc_hip_expand_packed", "gpc_hip_host_threads", "gpc_hip_host_numa_node", "gpc_hip_batch_stages", "gpc_hip_host_worker_cpus", "gpc_hip_fed_calls",
    "gpc_hip_enable_kernel_timing", "gpc_hip_set_kernel_timing_mask", "gpc_hip_reset_kernel_timing", "gpc_hip_kernel_count",
    "gpc_hip_kernel_name", "gpc_hip_kernel_launch_name", "gpc_hip_kernel_time",
    "gpc_hip_train_set_create", "gpc_hip_train_set_destroy", "gpc_hip_train_set_size", "gpc_hip_train_set_marks",
    "gpc_hip_train_eval_split", "gpc_hip_train_mark_split_samples", "gpc_hip_train_fern",
    "gpc_hip_train_begin_fern", "gpc_hip_train_eval_level", "gpc_hip_train_commit_level",
]


def load():
    """Loads libgpc_hip.so (built by opengpc_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            "%s is missing: build it with `python -m opengpc_amd.build` (needs hipcc). "
            "There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.gpc_hip_status_string.restype = C.c_char_p
    L.gpc_hip_last_error.restype = C.c_char_p
    L.gpc_hip_last_error.argtypes = [C.c_void_p]
    L.gpc_hip_kernel_name.restype = C.c_char_p
    L.gpc_hip_kernel_launch_name.restype = C.c_char_p
    L.gpc_hip_kernel_launch_name.argtypes = [C.c_void_p, C.c_int]
    L.gpc_hip_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.gpc_hip_destroy.argtypes = [C.c_void_p]
    L.gpc_hip_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.gpc_hip_synchronize.argtypes = [C.c_void_p]
    L.gpc_hip_reserve.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.gpc_hip_set_arithmetic.argtypes = [C.c_void_p, C.c_int]
    L.gpc_hip_host_alloc.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
    L.gpc_hip_host_free.argtypes = [C.c_void_p, C.c_void_p]
    L.gpc_hip_read_forest.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(FilterMask)]
    L.gpc_hip_parse_forest.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(FilterMask)]
    L.gpc_hip_set_forest.argtypes = [C.c_void_p, C.POINTER(FilterMask)]
    L.gpc_hip_preprocess.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.gpc_hip_warmup.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(Settings)]
    L.gpc_hip_preprocess_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.gpc_hip_preprocess_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.gpc_hip_resident_hits.argtypes = [C.c_void_p]
    L.gpc_hip_hash_codes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    pre = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
           C.c_int, C.c_int, C.c_int, C.POINTER(Settings), C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.gpc_hip_rectified_match.argtypes = pre
    L.gpc_hip_stereo_match.argtypes = pre
    L.gpc_hip_rectified_match_begin.argtypes = pre[:12]
    L.gpc_hip_stereo_match_begin.argtypes = pre[:12]
    L.gpc_hip_match_pair_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(Settings)]
    L.gpc_hip_match_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                      C.POINTER(C.c_int)]
    L.gpc_hip_match_pair.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                     C.POINTER(Settings), C.c_void_p, C.c_int, C.POINTER(C.c_int),
                                     C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.gpc_hip_match_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                             C.POINTER(Settings), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.gpc_hip_match_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(Settings), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.gpc_hip_match_batch_device_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                    C.POINTER(Settings), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                    C.c_void_p]
    L.gpc_hip_match_batch_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                             C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.gpc_hip_expand_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.gpc_hip_host_threads.argtypes = [C.c_void_p]
    L.gpc_hip_host_numa_node.argtypes = [C.c_void_p]
    L.gpc_hip_fed_calls.argtypes = [C.c_void_p]
    L.gpc_hip_set_pipeline.argtypes = [C.c_void_p, C.c_int]
    L.gpc_hip_pipeline_join.argtypes = [C.c_void_p]
    L.gpc_hip_batch_stages.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.gpc_hip_host_worker_cpus.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
    L.gpc_hip_enable_kernel_timing.argtypes = [C.c_void_p, C.c_int]
    L.gpc_hip_set_kernel_timing_mask.argtypes = [C.c_void_p, C.c_uint]
    L.gpc_hip_reset_kernel_timing.argtypes = [C.c_void_p]
    L.gpc_hip_kernel_time.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    vp, ci = C.c_void_p, C.c_int
    L.gpc_hip_train_set_create.argtypes = [vp, vp, ci, C.POINTER(C.c_void_p)]
    L.gpc_hip_train_set_destroy.argtypes = [vp, vp]
    L.gpc_hip_train_set_size.argtypes = [vp]
    L.gpc_hip_train_set_marks.argtypes = [vp, vp, vp, vp]
    L.gpc_hip_train_eval_split.argtypes = [vp, vp, vp, ci, C.c_double, vp]
    L.gpc_hip_train_mark_split_samples.argtypes = [vp, vp, vp, ci]
    L.gpc_hip_train_fern.argtypes = [vp, vp, ci, vp, ci, ci, ci, ci, C.c_double, vp, vp]
    L.gpc_hip_train_begin_fern.argtypes = [vp, vp, ci]
    L.gpc_hip_train_eval_level.argtypes = [vp, vp, vp, ci, ci, ci, vp, vp, vp]
    L.gpc_hip_train_commit_level.argtypes = [vp, vp, vp, ci]
    _lib = L
    return L


def _check(L, ctx, status, allow=()):
    if status == OK or status in allow:
        return status
    msg = L.gpc_hip_status_string(status).decode()
    if status == E_HIP and ctx:
        msg += ": " + L.gpc_hip_last_error(ctx).decode()
    raise GpcError(status, msg)


def expand_packed(packed, rows, n):
    """Host-side expansion of one pair's packed supports (gpc_hip_expand_packed) -> SUPPORT_DTYPE array of n records."""
    L = load()
    packed = np.ascontiguousarray(packed, np.uint32)
    rows = np.ascontiguousarray(rows, np.int32)
    out = np.empty(max(int(n), 1), SUPPORT_DTYPE)
    st = L.gpc_hip_expand_packed(_ptr(packed), _ptr(rows), len(rows), int(n), _ptr(out))
    _check(L, None, st)
    return out[:int(n)]


def read_forest(path, width, height):
    """Forest::readForest.  Returns (status, FilterMask); a missing file gives (E_IO, empty mask)."""
    L = load()
    fm = FilterMask()
    st = L.gpc_hip_read_forest(os.fsencode(path), width, height, C.byref(fm))
    return st, fm


def parse_forest(text, width, height):
    L = load()
    fm = FilterMask()
    st = L.gpc_hip_parse_forest(text.encode(), width, height, C.byref(fm))
    return st, fm


# training (include/gpc_hip.h): gpc_split = the scoring fields of Feature::params; gpc_split_stats = splitStats
SPLIT_DTYPE = np.dtype([("i", "<i4"), ("j", "<i4"), ("tau", "<i4")])
STATS_DTYPE = np.dtype([("prec", "<f8"), ("rec", "<f8"), ("hmean", "<f8"), ("convcomb", "<f8"),
                        ("tp", "<i4"), ("fp", "<i4"), ("fn", "<i4"), ("tot", "<i4")])
PATCH_BYTES = 729


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


_live_contexts = weakref.WeakSet()


def _close_live_contexts():
    """Contexts still open at interpreter exit are closed here, while the HIP runtime is still up;
    a destructor that reaches the library during interpreter teardown could otherwise call into a
    runtime that is already being destroyed."""
    for c in list(_live_contexts):
        try:
            c.close()
        except Exception:
            pass


atexit.register(_close_live_contexts)


class Context:
    """One gpc_hip_ctx: one device, one stream, one host thread at a time."""

    def __init__(self, device=0):
        self.L = load()
        self.h = None
        h = C.c_void_p()
        _check(self.L, None, self.L.gpc_hip_create(device, C.byref(h)))
        self.h = h
        self.device = device
        self._pinned = []
        _live_contexts.add(self)

    def close(self):
        if self.h:
            for p in self._pinned:
                self.L.gpc_hip_host_free(self.h, p)
            self._pinned = []
            self.L.gpc_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        if sys.is_finalizing():  # atexit already closed what was open; never call HIP from teardown
            return
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, st, allow=()):
        return _check(self.L, self.h, st, allow)

    # ---- setup
    def set_stream(self, stream_ptr):
        self._ck(self.L.gpc_hip_set_stream(self.h, C.c_void_p(stream_ptr or 0)))

    def synchronize(self):
        self._ck(self.L.gpc_hip_synchronize(self.h))

    def reserve(self, width, height, max_pairs):
        self._ck(self.L.gpc_hip_reserve(self.h, width, height, max_pairs))

    def pinned_empty(self, shape, dtype):
        """numpy array in page-locked host memory (gpc_hip_host_alloc); freed with the context."""
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        p = C.c_void_p()
        self._ck(self.L.gpc_hip_host_alloc(self.h, max(nbytes, 1), C.byref(p)))
        self._pinned.append(p)
        buf = (C.c_uint8 * max(nbytes, 1)).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def set_arithmetic(self, naive):
        """False: the reference's default SSE build; True: its SSE=OFF (*Naive) arithmetic."""
        self._ck(self.L.gpc_hip_set_arithmetic(self.h, 1 if naive else 0))

    def set_forest(self, fm):
        self._ck(self.L.gpc_hip_set_forest(self.h, C.byref(fm)))

    def load_forest(self, path, width, height):
        st, fm = read_forest(path, width, height)
        _check(self.L, None, st)
        self.set_forest(fm)
        return fm

    # ---- host-buffer entry points
    def preprocess(self, raw, threshold):
        raw = np.ascontiguousarray(raw, np.uint8)
        H, W = raw.shape
        smooth = np.empty((H, W), np.uint8)
        grad = np.empty((H, W), np.uint8)
        mask = np.empty(W * H, np.int32)
        n = C.c_int()
        self._ck(self.L.gpc_hip_preprocess(self.h, _ptr(raw), W, H, int(threshold), _ptr(smooth), _ptr(grad),
                                           _ptr(mask), mask.size, C.byref(n)))
        return smooth, grad, mask[:n.value].copy()

    def preprocess_resident(self, raw, threshold):
        """gpc_hip_preprocess_begin + _fetch: the arrays returned are the ones the library filled and remembers as the
        host copies of the image it keeps on the device -- hand THEM (not copies) to rectified_match / stereo_match and
        the match runs from the resident image (resident_hits() counts those calls)."""
        raw = np.ascontiguousarray(raw, np.uint8)
        H, W = raw.shape
        self._ck(self.L.gpc_hip_preprocess_begin(self.h, _ptr(raw), W, H, int(threshold)))
        smooth = np.empty((H, W), np.uint8)
        grad = np.empty((H, W), np.uint8)
        full = np.empty((W - 26) * (H - 26), np.int32)    # (allocated while the device works)
        n = C.c_int()
        self._ck(self.L.gpc_hip_preprocess_fetch(self.h, _ptr(smooth), _ptr(grad), _ptr(full), full.size, C.byref(n)))
        return smooth, grad, full[:n.value]      # a view: the same address the library remembers

    def match_async(self, kind, a, b, settings, cap=None):
        """The two-step forms: kind 'rectified' / 'stereo' on (smooth, grad, mask) triples, 'pair' on raw images.
        Returns (records, true count, status, (candidates L, candidates R) or None)."""
        s_ = settings
        if kind == "pair":
            a = np.ascontiguousarray(a, np.uint8)
            b = np.ascontiguousarray(b, np.uint8)
            H, W = a.shape
            self._ck(self.L.gpc_hip_match_pair_begin(self.h, _ptr(a), _ptr(b), W, H, C.byref(s_)))
            dtype = SUPPORT_DTYPE
        else:
            (sl, gl, ml), (sr, gr, mr) = a, b
            H, W = sl.shape
            fn = self.L.gpc_hip_rectified_match_begin if kind == "rectified" else self.L.gpc_hip_stereo_match_begin
            self._ck(fn(self.h, _ptr(sl), _ptr(gl), _ptr(ml), len(ml), _ptr(sr), _ptr(gr), _ptr(mr), len(mr), W, H, C.byref(s_)))
            dtype = SUPPORT_DTYPE if kind == "rectified" else CORR_DTYPE
        cap = cap if cap is not None else W * H
        out = np.empty(max(cap, 1), dtype)
        n, nl, nr = C.c_int(), C.c_int(-1), C.c_int(-1)
        st = self._ck(self.L.gpc_hip_match_fetch(self.h, _ptr(out), cap, C.byref(n), C.byref(nl), C.byref(nr)), allow=(E_CAPACITY,))
        if st == E_CAPACITY:     # the results stay until the next call: fetch again with room for all of them
            big = np.empty(n.value, dtype)
            self._ck(self.L.gpc_hip_match_fetch(self.h, _ptr(big), n.value, C.byref(n), C.byref(nl), C.byref(nr)))
            assert np.array_equal(big[:cap].view(np.uint8), out[:cap].view(np.uint8))
        return out[:min(n.value, cap)].copy(), n.value, st, ((nl.value, nr.value) if kind == "pair" else None)

    def batch_stages(self):
        """ms since entry of the last match_batch / match_batch_packed call: upload seen done, kernels done, last packed
        chunk landed, delivery done (gpc_hip_batch_stages)."""
        a = (C.c_float * 4)()
        self.L.gpc_hip_batch_stages(self.h, a)
        return [float(v) for v in a]

    def worker_cpus(self):
        a = (C.c_int * 64)()
        n = self.L.gpc_hip_host_worker_cpus(self.h, a, 64)
        return [int(a[i]) for i in range(min(n, 64))]

    def resident_hits(self):
        return self.L.gpc_hip_resident_hits(self.h)

    def warmup(self, width, height, settings=None):
        """gpc_hip_warmup: first-use costs of a pair of this size, paid now (the forest must be set)."""
        self._ck(self.L.gpc_hip_warmup(self.h, width, height, C.byref(settings) if settings is not None else None))

    def hash_codes(self, smooth, grad):
        smooth = np.ascontiguousarray(smooth, np.uint8)
        grad = np.ascontiguousarray(grad, np.uint8)
        H, W = smooth.shape
        codes = np.empty((H, W), np.uint32)
        self._ck(self.L.gpc_hip_hash_codes(self.h, _ptr(smooth), _ptr(grad), W, H, _ptr(codes)))
        return codes

    def _match_pre(self, fn, dtype, L_img, R_img, settings, cap):
        (sl, gl, ml), (sr, gr, mr) = L_img, R_img
        sl, gl, sr, gr = [np.ascontiguousarray(a, np.uint8) for a in (sl, gl, sr, gr)]
        ml = np.ascontiguousarray(ml, np.int32)
        mr = np.ascontiguousarray(mr, np.int32)
        H, W = sl.shape
        cap = cap if cap is not None else W * H
        out = np.empty(max(cap, 1), dtype)
        n = C.c_int()
        st = fn(self.h, _ptr(sl), _ptr(gl), _ptr(ml), len(ml), _ptr(sr), _ptr(gr), _ptr(mr), len(mr), W, H,
                C.byref(settings), _ptr(out), cap, C.byref(n))
        self._ck(st, allow=(E_CAPACITY,))
        return out[:min(n.value, cap)].copy(), n.value, st

    def rectified_match(self, L_img, R_img, settings, cap=None):
        """Forest::rectifiedMatch on (smooth, grad, mask) triples."""
        return self._match_pre(self.L.gpc_hip_rectified_match, SUPPORT_DTYPE, L_img, R_img, settings, cap)

    def stereo_match(self, L_img, R_img, settings, cap=None):
        """Forest::stereoMatch on (smooth, grad, mask) triples."""
        return self._match_pre(self.L.gpc_hip_stereo_match, CORR_DTYPE, L_img, R_img, settings, cap)

    def match_pair(self, rawL, rawR, settings, cap=None):
        rawL = np.ascontiguousarray(rawL, np.uint8)
        rawR = np.ascontiguousarray(rawR, np.uint8)
        H, W = rawL.shape
        cap = cap if cap is not None else W * H
        out = np.empty(max(cap, 1), SUPPORT_DTYPE)
        n, nl, nr = C.c_int(), C.c_int(), C.c_int()
        st = self.L.gpc_hip_match_pair(self.h, _ptr(rawL), _ptr(rawR), W, H, C.byref(settings), _ptr(out), cap,
                                       C.byref(n), C.byref(nl), C.byref(nr))
        self._ck(st, allow=(E_CAPACITY,))
        return out[:min(n.value, cap)].copy(), n.value, (nl.value, nr.value), st

    def match_batch(self, rawL, rawR, settings, cap, out=None):
        rawL = np.ascontiguousarray(rawL, np.uint8)
        rawR = np.ascontiguousarray(rawR, np.uint8)
        P, H, W = rawL.shape
        if out is None:
            out = np.empty((P, cap), SUPPORT_DTYPE)
        counts = np.empty(P, np.int32)
        ncand = np.empty((P, 2), np.int32)
        st = self.L.gpc_hip_match_batch(self.h, _ptr(rawL), _ptr(rawR), W, H, P, C.byref(settings), _ptr(out), cap,
                                        _ptr(counts), _ptr(ncand))
        self._ck(st, allow=(E_CAPACITY,))
        return out, counts, ncand, st

    def match_batch_packed(self, rawL, rawR, settings, cap, packed=None, rows=None):
        """Host images -> PACKED results in host memory (gpc_hip_match_batch_packed): words xL | xR << 16 [P][cap], per-row
        counts [P][H], true counts [P], candidate counts [P][2]; expand_packed() makes a pair's ndb::Support records."""
        rawL = np.ascontiguousarray(rawL, np.uint8)
        rawR = np.ascontiguousarray(rawR, np.uint8)
        P, H, W = rawL.shape
        if packed is None:
            packed = np.empty((P, cap), np.uint32)
        if rows is None:
            rows = np.empty((P, H), np.int32)
        counts = np.empty(P, np.int32)
        ncand = np.empty((P, 2), np.int32)
        st = self.L.gpc_hip_match_batch_packed(self.h, _ptr(rawL), _ptr(rawR), W, H, P, C.byref(settings), _ptr(packed), cap,
                                               _ptr(rows), _ptr(counts), _ptr(ncand))
        self._ck(st, allow=(E_CAPACITY,))
        return packed, rows, counts, ncand, st

    # ---- device-resident batch (pointers are integers, e.g. torch.Tensor.data_ptr())
    def match_batch_device(self, d_rawL, d_rawR, width, height, npairs, settings, d_out, cap_per_pair,
                           d_counts, d_ncand=0):
        self._ck(self.L.gpc_hip_match_batch_device(self.h, C.c_void_p(d_rawL), C.c_void_p(d_rawR), width, height,
                                                   npairs, C.byref(settings), C.c_void_p(d_out), cap_per_pair,
                                                   C.c_void_p(d_counts), C.c_void_p(d_ncand or 0)))

    def set_pipeline(self, lanes):
        """2: consecutive match_batch_device calls alternate between two lanes (gpc_hip_set_pipeline); 1: strict."""
        self._ck(self.L.gpc_hip_set_pipeline(self.h, int(lanes)))

    def pipeline_join(self):
        self._ck(self.L.gpc_hip_pipeline_join(self.h))

    def match_batch_device_packed(self, d_rawL, d_rawR, width, height, npairs, settings, d_packed, cap_per_pair,
                                  d_rows, d_counts, d_ncand=0):
        """Packed results (x | xR << 16 per support + per-row counts) left in HBM; epipolar sort-matcher only."""
        self._ck(self.L.gpc_hip_match_batch_device_packed(self.h, C.c_void_p(d_rawL), C.c_void_p(d_rawR), width, height,
                                                          npairs, C.byref(settings), C.c_void_p(d_packed), cap_per_pair,
                                                          C.c_void_p(d_rows), C.c_void_p(d_counts), C.c_void_p(d_ncand or 0)))

    # ---- fern training: the scoring loop
    def train_set(self, triplets):
        """Uploads (n, 3, 729) uint8 patch triplets (ref, pos, neg); returns a TrainSet."""
        return TrainSet(self, triplets)

    # ---- measurement
    def enable_kernel_timing(self, on=True, only=None):
        """HIP-event bracketing of kernel launches; `only` = iterable of kernel names to restrict it to."""
        mask = 0xFFFFFFFF
        if only is not None:
            names = [self.L.gpc_hip_kernel_name(i).decode() for i in range(self.L.gpc_hip_kernel_count())]
            mask = 0
            for n in only:
                mask |= 1 << names.index(n)
        self._ck(self.L.gpc_hip_set_kernel_timing_mask(self.h, mask))
        self._ck(self.L.gpc_hip_enable_kernel_timing(self.h, int(on)))

    def reset_kernel_timing(self):
        self._ck(self.L.gpc_hip_reset_kernel_timing(self.h))

    def kernel_launch_names(self):
        """{timing slot name: rocprofv3 name of the instantiation last launched there}"""
        return {self.L.gpc_hip_kernel_name(i).decode(): self.L.gpc_hip_kernel_launch_name(self.h, i).decode()
                for i in range(self.L.gpc_hip_kernel_count())}

    def kernel_times(self):
        """{kernel name: (total ms, launches)} since the last reset."""
        out = {}
        for i in range(self.L.gpc_hip_kernel_count()):
            ms, n = C.c_float(), C.c_int()
            self._ck(self.L.gpc_hip_kernel_time(self.h, i, C.byref(ms), C.byref(n)))
            out[self.L.gpc_hip_kernel_name(i).decode()] = (ms.value, n.value)
        return out


class TrainSet:
    """Device-resident training triplets of one Context (gpc_hip_train_set): Fern::evalSplit,
    Fern::markSplitSamples and Fern::train (with caller-supplied hyperplane samples) on the GPU."""

    def __init__(self, ctx, triplets):
        t = np.ascontiguousarray(triplets, np.uint8)
        if t.ndim != 3 or t.shape[1:] != (3, PATCH_BYTES):
            raise ValueError("triplets must have shape (n, 3, 729)")
        self.ctx = ctx
        self.n = len(t)
        h = C.c_void_p()
        ctx._ck(ctx.L.gpc_hip_train_set_create(ctx.h, _ptr(t), self.n, C.byref(h)))
        self.h = h

    def close(self):
        if self.h and self.ctx.h:
            self.ctx.L.gpc_hip_train_set_destroy(self.ctx.h, self.h)
        self.h = None

    def marks(self, new=None):
        """Reads (and optionally first replaces) the split marks: bit 0 pos.split, bit 1 neg.split."""
        out = np.empty(self.n, np.uint8)
        src = None if new is None else np.ascontiguousarray(new, np.uint8)
        self.ctx._ck(self.ctx.L.gpc_hip_train_set_marks(self.ctx.h, self.h, _ptr(src), _ptr(out)))
        return out

    def eval_split(self, params, score_until_level, w1):
        p = np.ascontiguousarray(params, SPLIT_DTYPE)
        st = np.zeros(1, STATS_DTYPE)
        self.ctx._ck(self.ctx.L.gpc_hip_train_eval_split(self.ctx.h, self.h, _ptr(p), int(score_until_level),
                                                         C.c_double(w1), _ptr(st)))
        return st[0]

    def mark_split_samples(self, params, num_params):
        p = np.ascontiguousarray(params, SPLIT_DTYPE)
        self.ctx._ck(self.ctx.L.gpc_hip_train_mark_split_samples(self.ctx.h, self.h, _ptr(p), int(num_params)))

    def train_fern(self, max_depth, cand, num_resamples, taulo, tauhi, only_score_non_split, w1):
        """cand[level * num_resamples + k] = k-th hyperplane sample of `level`.  Returns (params, stats)."""
        c = np.ascontiguousarray(cand, SPLIT_DTYPE)
        if len(c) < max_depth * num_resamples:
            raise ValueError("need max_depth * num_resamples hyperplane samples")
        fp = np.zeros(max_depth, SPLIT_DTYPE)
        st = np.zeros(max_depth, STATS_DTYPE)
        self.ctx._ck(self.ctx.L.gpc_hip_train_fern(self.ctx.h, self.h, int(max_depth), _ptr(c), int(num_resamples),
                                                   int(taulo), int(tauhi), int(bool(only_score_non_split)),
                                                   C.c_double(w1), _ptr(fp), _ptr(st)))
        return fp, st

    def begin_fern(self, reset_marks):
        self.ctx._ck(self.ctx.L.gpc_hip_train_begin_fern(self.ctx.h, self.h, int(bool(reset_marks))))

    def eval_level(self, cand, taulo, tauhi):
        """tp, fp of every (candidate, tau) of the current level and the number of samples that count."""
        c = np.ascontiguousarray(cand, SPLIT_DTYPE)
        ntau = tauhi - taulo
        tp = np.zeros((len(c), max(ntau, 0)), np.int32)
        fp = np.zeros_like(tp)
        tot = np.zeros(1, np.int32)
        self.ctx._ck(self.ctx.L.gpc_hip_train_eval_level(self.ctx.h, self.h, _ptr(c), len(c), int(taulo), int(tauhi),
                                                         _ptr(tp), _ptr(fp), _ptr(tot)))
        return tp, fp, int(tot[0])

    def commit_level(self, best, mark_split):
        b = np.ascontiguousarray(best, SPLIT_DTYPE).reshape(1)
        self.ctx._ck(self.ctx.L.gpc_hip_train_commit_level(self.ctx.h, self.h, _ptr(b), int(bool(mark_split))))
