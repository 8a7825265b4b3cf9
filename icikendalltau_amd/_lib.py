"""ctypes binding of the C ABI in include/icikt.h (libicikt_hip.so, built in-tree by build()).

There is no CPU implementation behind this module: if the shared library is missing, or no HIP
device is usable, every entry point raises.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
LIB_PATH = os.environ.get("ICIKT_LIB") or os.path.join(_PKG, "libicikt_hip.so")  # ICIKT_LIB: A/B of builds (tools)
SOURCES = [os.path.join(_PKG, "csrc", f) for f in ("icikt_kernels.hip", "icikt_capi.cpp", "icikt_multi.cpp")]
HEADERS = [os.path.join(_ROOT, "include", "icikt.h"), os.path.join(_PKG, "csrc", "icikt_device.h"),
           os.path.join(_PKG, "csrc", "icikt_host.h")]

# include/icikt.h
SUCCESS = 0
E_NO_DEVICE = -6
PERSPECTIVE = {"local": 0, "global": 1}
ALTERNATIVE = {"two.sided": 0, "less": 1, "greater": 2}
ALT_OTHER = 3
FLAG_EXACT_INT64 = 1
FLAG_TIMING = 2
FLAG_REUSE_COUNTS = 4
FLAG_BALANCE_COST = 16  # multi-device entries: pair blocks of equal cost (the streamed columns' tie structure) instead of equal length
FLAG_HOST_PINNED = 8     # the caller has page-locked the matrix and the result arrays of the call (pinned_empty)
CNT_FIELDS = ("n", "missing", "dis", "ntie", "xtie", "ytie", "x0", "x1", "y0", "y1", "tot")
K_PREPARE, K_PAIRS, K_EPILOGUE = 0, 1, 2
MASK_VALS = 32                # distinct finite global_na values of the device-side exclusion rule (icikt_device.h)
MAX_FEATURES = 65535          # the tuned kernels
MAX_FEATURES_WIDE = 262144    # the plain 32-bit path (exact integer arithmetic)
PREP_ARRAYS = 5  # order, rec, hirow, meta (bitsets + stats per column), tgroups
PREP_EXCHANGE = (0, 3)  # order and meta: the rest is rebuilt by expand_cols_dev()

REASON_OK, REASON_ALL_MISSING, REASON_SHORT, REASON_SINGLE_UNIQUE, REASON_TIES_EQ_TOTAL = range(5)
REASON_WARNINGS = {
    REASON_SHORT: "Warning: The vectors only have a single value, NA returned!",  # src/kendallc.cpp:225
    REASON_SINGLE_UNIQUE: "Warning: Either 'X' or 'Y' have only a single unique value, NA returned!",  # :238
    REASON_TIES_EQ_TOTAL: "Warning: Ties equal the total, NA returned!",  # :292
}

EXPORTS = (
    "icikt_version", "icikt_device_count", "icikt_ctx_create", "icikt_ctx_destroy", "icikt_last_error",
    "icikt_ctx_set_stream", "icikt_ctx_use_own_stream", "icikt_sync", "icikt_prepare_dev", "icikt_prepare_cols_dev", "icikt_prepare_cols_f64", "icikt_prep_arrays", "icikt_expand_cols_dev", "icikt_set_pairs", "icikt_set_pairs_combn",
    "icikt_num_pairs", "icikt_run_dev", "icikt_kernel_ms", "icikt_reset_timers", "icikt_pairs_f64",
    "icikt_pair_f64", "icikt_pairs_complete_f64", "icikt_missingness_f64", "icikt_selftest", "icikt_debug_set_plan",
    "icikt_multi_create", "icikt_multi_destroy", "icikt_multi_last_error", "icikt_multi_n_gpu", "icikt_multi_uses_rccl",
    "icikt_pairs_multi_f64", "icikt_multi_phase_ms", "icikt_multi_debug_set_plan",
    "icikt_matrix_f64", "icikt_matrix_multi_f64", "icikt_multi_rank_phase_ms", "icikt_multi_ranks_used",
    "icikt_debug_step_stats", "icikt_multi_comm_ranks", "icikt_multi_block_bounds", "icikt_cost_blocks",
)


class IciktError(RuntimeError):
    pass


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False, extra_flags=(), out: str | None = None) -> str:
    """Compile the HIP kernels + C ABI for gfx950 into icikendalltau_amd/libicikt_hip.so (or, for the development
    tools, a variant build with extra_flags into `out`: loaded through ICIKT_LIB)."""
    if out is None and not force and not needs_build():
        return LIB_PATH
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    hipcc = os.environ.get("HIPCC", os.path.join(rocm, "bin", "hipcc"))
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", *extra_flags,
           "-I", os.path.join(_ROOT, "include"), "-I", os.path.join(_PKG, "csrc"),
           "-o", out or LIB_PATH] + SOURCES + ["-L", os.path.join(rocm, "lib"), "-lrccl", "-pthread"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise IciktError("hipcc failed:\n" + res.stdout + res.stderr)
    if verbose:
        print(" ".join(cmd))
    return out or LIB_PATH


_lib = None


def lib():
    """Load the shared library (raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64.so.1.
    # Importing torch FIRST lets the dynamic linker satisfy this library's NEEDED sonames with torch's
    # copies; the other order loads /opt/rocm's runtime beside torch's and the second one finds no device.
    if os.environ.get("ICIKT_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    if not os.path.exists(LIB_PATH):
        raise IciktError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc); "
            "icikendalltau_amd has no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    c_int, c_i64, c_u32, c_vp = ctypes.c_int, ctypes.c_int64, ctypes.c_uint32, ctypes.c_void_p
    L.icikt_version.restype = c_int
    L.icikt_device_count.argtypes = [ctypes.POINTER(c_int)]
    L.icikt_ctx_create.argtypes = [c_int, ctypes.POINTER(c_vp)]
    L.icikt_ctx_destroy.argtypes = [c_vp]
    L.icikt_ctx_destroy.restype = None
    L.icikt_last_error.argtypes = [c_vp]
    L.icikt_last_error.restype = ctypes.c_char_p
    L.icikt_ctx_set_stream.argtypes = [c_vp, c_vp]
    L.icikt_ctx_use_own_stream.argtypes = [c_vp]
    L.icikt_sync.argtypes = [c_vp]
    L.icikt_prepare_dev.argtypes = [c_vp, c_vp, c_i64, c_i64, c_i64, c_u32]
    L.icikt_prepare_cols_dev.argtypes = [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_u32]
    L.icikt_prepare_cols_f64.argtypes = [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_u32]
    L.icikt_prep_arrays.argtypes = [c_vp, ctypes.POINTER(c_vp), ctypes.POINTER(c_i64)]
    L.icikt_expand_cols_dev.argtypes = [c_vp, c_i64, c_i64, c_u32]
    L.icikt_set_pairs.argtypes = [c_vp, c_vp, c_vp, c_i64]
    L.icikt_set_pairs_combn.argtypes = [c_vp, c_i64, c_i64, c_i64]
    L.icikt_num_pairs.argtypes = [c_vp]
    L.icikt_num_pairs.restype = c_i64
    L.icikt_run_dev.argtypes = [c_vp, c_int, c_int, c_int, c_u32, c_vp, c_vp, c_vp]
    L.icikt_kernel_ms.argtypes = [c_vp, c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_i64)]
    L.icikt_reset_timers.argtypes = [c_vp]
    L.icikt_pairs_f64.argtypes = [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_i64, c_int, c_int, c_int, c_u32,
                                  c_vp, c_vp, c_vp]
    L.icikt_pairs_complete_f64.argtypes = [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_i64, c_int, c_int, c_u32,
                                           c_vp, c_vp, c_vp]
    L.icikt_pair_f64.argtypes = [c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_int, c_u32, c_vp, c_vp, c_vp]
    L.icikt_missingness_f64.argtypes = [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_i64, c_vp]
    L.icikt_selftest.argtypes = [c_vp]
    L.icikt_debug_set_plan.argtypes = [c_vp, ctypes.c_char_p]
    L.icikt_multi_create.argtypes = [ctypes.POINTER(c_int), c_int, c_int, ctypes.POINTER(c_vp)]
    L.icikt_multi_destroy.argtypes = [c_vp]
    L.icikt_multi_destroy.restype = None
    L.icikt_multi_last_error.argtypes = [c_vp]
    L.icikt_multi_last_error.restype = ctypes.c_char_p
    L.icikt_multi_n_gpu.argtypes = [c_vp]
    L.icikt_multi_uses_rccl.argtypes = [c_vp]
    L.icikt_multi_comm_ranks.argtypes = [c_vp]
    L.icikt_multi_block_bounds.argtypes = [c_vp, ctypes.POINTER(c_i64)]
    L.icikt_cost_blocks.argtypes = [c_vp, c_i64, c_vp, c_i64, c_int, c_i64, c_vp]
    L.icikt_pairs_multi_f64.argtypes = L.icikt_pairs_f64.argtypes
    L.icikt_multi_phase_ms.argtypes = [c_vp, ctypes.POINTER(ctypes.c_double)]
    L.icikt_multi_debug_set_plan.argtypes = [c_vp, ctypes.c_char_p]
    L.icikt_matrix_f64.argtypes = [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_int, c_vp, c_vp, c_i64, c_int, c_int, c_int,
                                   c_u32, c_int, c_int, c_vp, c_vp, c_vp]
    L.icikt_matrix_multi_f64.argtypes = L.icikt_matrix_f64.argtypes
    L.icikt_multi_rank_phase_ms.argtypes = [c_vp, c_int, ctypes.POINTER(ctypes.c_double)]
    L.icikt_multi_ranks_used.argtypes = [c_vp]
    L.icikt_debug_step_stats.argtypes = [c_vp, c_vp, c_int]
    for name in EXPORTS:
        if getattr(L, name).restype is not None and name not in ("icikt_last_error", "icikt_num_pairs",
                                                                   "icikt_multi_last_error"):
            getattr(L, name).restype = c_int
    _lib = L
    return L


def device_count() -> int:
    n = ctypes.c_int(0)
    lib().icikt_device_count(ctypes.byref(n))
    return n.value


def _ptr(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def pinned_empty(shape, dtype=np.float64, order="C"):
    """A numpy array in PAGE-LOCKED host memory (hipHostMalloc through torch's pinned allocator; the array keeps the
    tensor alive).  Host entries called with FLAG_HOST_PINNED copy from / into such arrays directly."""
    import torch
    shape = (shape,) if np.isscalar(shape) else tuple(shape)
    n = int(np.prod(shape)) if shape else 1
    t = torch.empty(max(n, 1) * np.dtype(dtype).itemsize, dtype=torch.uint8, pin_memory=True)
    a = t.numpy()[:n * np.dtype(dtype).itemsize].view(dtype).reshape(shape, order=order)
    return a


def _matrix_call(fn, handle, chk, X, global_na, pi, pj, perspective, alternative, continuity, flags, scale_max,
                 diag_good, want_keep):
    """icikt_matrix_f64 / icikt_matrix_multi_f64: (out5 [5, S, S], keep [S, n_feat] bool or None, reason_counts [5])."""
    if not (isinstance(X, np.ndarray) and X.dtype == np.float64 and X.ndim == 2 and X.flags.f_contiguous):
        X = np.asfortranarray(X, dtype=np.float64)
    n_feat, n_samp = X.shape
    gna = np.ascontiguousarray([] if global_na is None else np.atleast_1d(global_na), dtype=np.float64)
    if pi is None:
        pi_a = pj_a = None
        P = 0
    else:
        pi_a = np.ascontiguousarray(pi, dtype=np.int32)
        pj_a = np.ascontiguousarray(pj, dtype=np.int32)
        P = pi_a.shape[0]
    out5 = np.empty((5, n_samp, n_samp), dtype=np.float64)
    keep = np.empty((n_samp, n_feat), dtype=np.uint8) if want_keep else None
    rc5 = np.zeros(5, dtype=np.int64)
    alt = ALTERNATIVE.get(alternative, ALT_OTHER)
    chk(fn(handle, _ptr(X), n_feat, n_samp, max(n_feat, 0), _ptr(gna) if gna.size else None, int(gna.size), _ptr(pi_a),
           _ptr(pj_a), P, PERSPECTIVE[perspective], alt, int(bool(continuity)), flags, int(bool(scale_max)),
           int(bool(diag_good)), _ptr(out5), _ptr(keep), _ptr(rc5)), fn.__name__)
    return out5, (keep.view(np.bool_) if keep is not None else None), rc5


class Context:
    """One HIP device + stream + workspaces (icikt_ctx)."""

    def __init__(self, device: int = 0):
        self._h = ctypes.c_void_p()
        rc = lib().icikt_ctx_create(int(device), ctypes.byref(self._h))
        if rc == E_NO_DEVICE:
            raise IciktError("no usable HIP device: icikendalltau_amd computes on an MI355X only (no CPU fallback)")
        if rc != SUCCESS:
            raise IciktError(f"icikt_ctx_create(device={device}) failed with code {rc}")
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().icikt_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int, what: str):
        if rc != SUCCESS:
            msg = lib().icikt_last_error(self._h)
            raise IciktError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")

    # -- device-resident path --------------------------------------------------------------------
    def set_stream(self, hip_stream: int | None):
        """Run on an existing hipStream_t (0 / None = HIP's default stream, e.g. torch's default)."""
        self._chk(lib().icikt_ctx_set_stream(self._h, ctypes.c_void_p(hip_stream or 0)), "icikt_ctx_set_stream")

    def use_own_stream(self):
        self._chk(lib().icikt_ctx_use_own_stream(self._h), "icikt_ctx_use_own_stream")

    def sync(self):
        self._chk(lib().icikt_sync(self._h), "icikt_sync")

    def prepare_dev(self, d_ptr: int, n_feat: int, n_samp: int, ld: int, flags: int = 0):
        self._chk(lib().icikt_prepare_dev(self._h, ctypes.c_void_p(d_ptr), n_feat, n_samp, ld, flags),
                  "icikt_prepare_dev")

    def prepare_cols_dev(self, d_ptr: int, n_feat: int, n_samp: int, ld: int, col_begin: int, col_end: int,
                         alloc_cols: int, flags: int = 0):
        """Pre-pass over columns [col_begin, col_end) only (multi-rank: all-gather prep_arrays() afterwards)."""
        self._chk(lib().icikt_prepare_cols_dev(self._h, ctypes.c_void_p(d_ptr), n_feat, n_samp, ld, col_begin,
                                               col_end, alloc_cols, flags), "icikt_prepare_cols_dev")

    def prepare_cols(self, X, col_begin: int, col_end: int, alloc_cols: int, flags: int = 0):
        """Pre-pass over columns [col_begin, col_end) of a HOST matrix (F-ordered float64, NaN = missing): only
        those columns cross PCIe."""
        if not (isinstance(X, np.ndarray) and X.dtype == np.float64 and X.flags.f_contiguous and X.ndim == 2):
            raise ValueError("prepare_cols needs a Fortran-ordered float64 matrix")
        n_feat, n_samp = X.shape
        self._chk(lib().icikt_prepare_cols_f64(self._h, _ptr(X), n_feat, n_samp, max(n_feat, 0), col_begin, col_end,
                                               alloc_cols, flags), "icikt_prepare_cols_f64")

    def prep_arrays(self):
        """[(device pointer, bytes per column)] of the prepared-state arrays."""
        ptrs = (ctypes.c_void_p * PREP_ARRAYS)()
        bpc = (ctypes.c_int64 * PREP_ARRAYS)()
        self._chk(lib().icikt_prep_arrays(self._h, ptrs, bpc), "icikt_prep_arrays")
        return [(int(ptrs[i] or 0), int(bpc[i])) for i in range(PREP_ARRAYS)]

    def expand_cols_dev(self, col_begin: int, col_end: int, flags: int = 0):
        """Rebuild rec / hirow / girow / tgroups and the tie program with its step records of columns [col_begin, col_end)
        from their (received) order and gflag."""
        self._chk(lib().icikt_expand_cols_dev(self._h, col_begin, col_end, flags), "icikt_expand_cols_dev")

    def set_pairs(self, pi, pj):
        pi = np.ascontiguousarray(pi, dtype=np.int32)
        pj = np.ascontiguousarray(pj, dtype=np.int32)
        if pi.shape != pj.shape or pi.ndim != 1:
            raise ValueError("pi and pj must be 1-D arrays of the same length")
        self._chk(lib().icikt_set_pairs(self._h, _ptr(pi), _ptr(pj), pi.shape[0]), "icikt_set_pairs")

    def set_pairs_combn(self, n_samp: int, begin: int, end: int):
        self._chk(lib().icikt_set_pairs_combn(self._h, n_samp, begin, end), "icikt_set_pairs_combn")

    def num_pairs(self) -> int:
        return int(lib().icikt_num_pairs(self._h))

    def run_dev(self, perspective: int, alternative: int, continuity: bool, flags: int, d_out4: int,
                d_counts: int | None = None, d_reasons: int | None = None):
        self._chk(lib().icikt_run_dev(self._h, perspective, alternative, int(bool(continuity)), flags,
                                      ctypes.c_void_p(d_out4), ctypes.c_void_p(d_counts or 0),
                                      ctypes.c_void_p(d_reasons or 0)), "icikt_run_dev")

    def kernel_ms(self, kernel: int):
        ms = ctypes.c_double(0)
        n = ctypes.c_int64(0)
        self._chk(lib().icikt_kernel_ms(self._h, kernel, ctypes.byref(ms), ctypes.byref(n)), "icikt_kernel_ms")
        return ms.value, n.value

    def reset_timers(self):
        self._chk(lib().icikt_reset_timers(self._h), "icikt_reset_timers")

    def selftest(self):
        self._chk(lib().icikt_selftest(self._h), "icikt_selftest")

    def step_stats(self, reset: bool = True):
        """Diagnostic builds only (-DICIKT_STEP_STATS): {kind: (steps, rows, wave cycles)} of the pair kernel."""
        buf = np.zeros(24, dtype=np.uint64)
        self._chk(lib().icikt_debug_step_stats(self._h, _ptr(buf), int(reset)), "icikt_debug_step_stats")
        kinds = ("hot_loop", "hot_in_main_or_solo", "mixed", "group_rest", "group_top_phaseA", "tail", "setup", "group_close_insert_rebuild")
        return {k: (int(buf[i]), int(buf[8 + i]), int(buf[16 + i])) for i, k in enumerate(kinds)}

    def debug_set_plan(self, spec: str | dict | None = None):
        """Test / experiment hook: override the pair kernel's launch plan ("np=1,pend=g,..." or a dict; None resets)."""
        if isinstance(spec, dict):
            spec = ",".join(f"{k}={v}" for k, v in spec.items() if v not in ("", None))
        self._chk(lib().icikt_debug_set_plan(self._h, (spec or "").encode()), "icikt_debug_set_plan")

    # -- host-buffer path ------------------------------------------------------------------------
    def pairs(self, X, pi=None, pj=None, perspective="global", alternative="two.sided", continuity=False,
              flags: int = 0, want_counts: bool = True):
        """ici_split() over a host matrix (n_feat x n_samp, NaN = missing); pairs 0-based or None = all."""
        Xf = np.asfortranarray(X, dtype=np.float64)
        if Xf.ndim != 2:
            raise ValueError("X must be 2-D (features x samples)")
        n_feat, n_samp = Xf.shape
        if pi is None:
            P = n_samp * (n_samp - 1) // 2
            pi_a = pj_a = None
        else:
            pi_a = np.ascontiguousarray(pi, dtype=np.int32)
            pj_a = np.ascontiguousarray(pj, dtype=np.int32)
            P = pi_a.shape[0]
        alloc = pinned_empty if (flags & FLAG_HOST_PINNED) else np.empty   # (X is the caller's: pinned by the caller)
        out = alloc((P, 4), dtype=np.float64)
        cnt = alloc((P, len(CNT_FIELDS)), dtype=np.int64) if want_counts else None
        rsn = alloc(P, dtype=np.int32)
        if cnt is not None:
            cnt[...] = 0
        rsn[...] = 0
        alt = ALTERNATIVE.get(alternative, ALT_OTHER)
        self._chk(lib().icikt_pairs_f64(self._h, _ptr(Xf), n_feat, n_samp, max(n_feat, 0), _ptr(pi_a), _ptr(pj_a), P,
                                        PERSPECTIVE[perspective], alt, int(bool(continuity)), flags, _ptr(out),
                                        _ptr(cnt), _ptr(rsn)), "icikt_pairs_f64")
        return out, cnt, rsn

    def matrix(self, X, global_na=None, pi=None, pj=None, perspective="global", alternative="two.sided",
               continuity=False, flags: int = 0, scale_max=True, diag_good=True, want_keep=True):
        """ici_kendalltau() below its argument checks in ONE call (icikt_matrix_f64): the exclusion rule, the pair
        kernels and scale_and_reshape all run on the device.  X: raw data (features x samples, F-ordered float64 is
        taken as is); global_na: the values setup_missing_matrix excludes (NaN = NA, Inf, finite values)."""
        return _matrix_call(lib().icikt_matrix_f64, self._h, self._chk, X, global_na, pi, pj, perspective, alternative,
                            continuity, flags, scale_max, diag_good, want_keep)

    def pairs_complete(self, X, pi, pj, alternative="two.sided", continuity=False, flags: int = 0,
                       want_counts: bool = False):
        """kt_fast(use = "pairwise.complete.obs"): per pair, rows with a missing value in either vector are dropped."""
        Xf = np.asfortranarray(X, dtype=np.float64)
        n_feat, n_samp = Xf.shape
        pi_a = np.ascontiguousarray(pi, dtype=np.int32)
        pj_a = np.ascontiguousarray(pj, dtype=np.int32)
        P = pi_a.shape[0]
        out = np.empty((P, 4), dtype=np.float64)
        cnt = np.zeros((P, len(CNT_FIELDS)), dtype=np.int64) if want_counts else None
        rsn = np.zeros(P, dtype=np.int32)
        alt = ALTERNATIVE.get(alternative, ALT_OTHER)
        self._chk(lib().icikt_pairs_complete_f64(self._h, _ptr(Xf), n_feat, n_samp, max(n_feat, 0), _ptr(pi_a), _ptr(pj_a),
                                                 P, alt, int(bool(continuity)), flags, _ptr(out), _ptr(cnt), _ptr(rsn)),
                  "icikt_pairs_complete_f64")
        return out, cnt, rsn

    def pair(self, x, y, perspective="local", alternative="two.sided", continuity=False, flags: int = 0):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        out = np.empty(4, dtype=np.float64)
        cnt = np.zeros(len(CNT_FIELDS), dtype=np.int64)
        rsn = np.zeros(1, dtype=np.int32)
        alt = ALTERNATIVE.get(alternative, ALT_OTHER)
        self._chk(lib().icikt_pair_f64(self._h, _ptr(x), _ptr(y), x.shape[0], PERSPECTIVE[perspective], alt,
                                       int(bool(continuity)), flags, _ptr(out), _ptr(cnt), _ptr(rsn)),
                  "icikt_pair_f64")
        return out, dict(zip(CNT_FIELDS, cnt.tolist())), int(rsn[0])

    def missingness(self, X, pi, pj):
        Xf = np.asfortranarray(X, dtype=np.float64)
        n_feat, n_samp = Xf.shape
        pi_a = np.ascontiguousarray(pi, dtype=np.int32)
        pj_a = np.ascontiguousarray(pj, dtype=np.int32)
        out = np.zeros(pi_a.shape[0], dtype=np.int64)
        self._chk(lib().icikt_missingness_f64(self._h, _ptr(Xf), n_feat, n_samp, max(n_feat, 0), _ptr(pi_a),
                                              _ptr(pj_a), pi_a.shape[0], _ptr(out)), "icikt_missingness_f64")
        return out


def cost_blocks(col_cost, n_blocks: int, pj=None, max_block: int = 0):
    """The cost-weighted cut of the pair list (icikt_cost_blocks; host arithmetic, no device): bounds[0 .. n_blocks]."""
    cost = np.ascontiguousarray(col_cost, dtype=np.uint32)
    pj_a = None if pj is None else np.ascontiguousarray(pj, dtype=np.int32)
    b = np.zeros(n_blocks + 1, dtype=np.int64)
    rc = lib().icikt_cost_blocks(_ptr(cost), cost.shape[0], _ptr(pj_a), -1 if pj_a is None else pj_a.shape[0], n_blocks,
                                 int(max_block), _ptr(b))
    if rc != SUCCESS:
        raise IciktError(f"icikt_cost_blocks failed (code {rc})")
    return b.tolist()


EXCHANGE = {"auto": 0, "rccl": 1, "copy": 2}
MULTI_PHASES = ("prepare", "exchange", "pairs", "gather")


class MultiContext:
    """Several MI355X behind one call (icikt_multi): one host thread per device inside the library, column-sharded
    pre-pass, RCCL all-gather / gather over xGMI."""

    def __init__(self, devices=None, n_gpu: int | None = None, exchange: str = "auto"):
        if devices is None:
            devices = list(range(int(n_gpu or 1)))
        self.devices = [int(d) for d in devices]
        arr = (ctypes.c_int * len(self.devices))(*self.devices)
        self._h = ctypes.c_void_p()
        rc = lib().icikt_multi_create(arr, len(self.devices), EXCHANGE[exchange], ctypes.byref(self._h))
        if rc == E_NO_DEVICE:
            raise IciktError("no usable HIP device: icikendalltau_amd computes on MI355X only (no CPU fallback)")
        if rc != SUCCESS:
            raise IciktError(f"icikt_multi_create(devices={self.devices}, exchange={exchange!r}) failed with code {rc}")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().icikt_multi_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def uses_rccl(self) -> bool:
        return bool(lib().icikt_multi_uses_rccl(self._h))

    def block_bounds(self) -> list:
        """Pair blocks of the last call: rank r ran pairs [b[r], b[r + 1]) of the list."""
        n = self.ranks_used + 1
        b = (ctypes.c_int64 * n)()
        self._chk(lib().icikt_multi_block_bounds(self._h, b), "icikt_multi_block_bounds")
        return list(b)

    @property
    def comm_ranks(self) -> int:
        """Ranks of the RCCL communicator as RCCL reports them (0: device copies, no communicator)."""
        return int(lib().icikt_multi_comm_ranks(self._h))

    def _chk(self, rc: int, what: str):
        if rc != SUCCESS:
            msg = lib().icikt_multi_last_error(self._h)
            raise IciktError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")

    def debug_set_plan(self, spec: str | dict | None = None):
        if isinstance(spec, dict):
            spec = ",".join(f"{k}={v}" for k, v in spec.items() if v not in ("", None))
        self._chk(lib().icikt_multi_debug_set_plan(self._h, (spec or "").encode()), "icikt_multi_debug_set_plan")

    def phase_ms(self) -> dict:
        """Phases of the last call, the maximum over the ranks."""
        ms = (ctypes.c_double * len(MULTI_PHASES))()
        self._chk(lib().icikt_multi_phase_ms(self._h, ms), "icikt_multi_phase_ms")
        return dict(zip(MULTI_PHASES, list(ms)))

    def rank_phase_ms(self) -> list:
        """Per rank: its own phases and the time it waited for the others at the barriers ("wait")."""
        out = []
        for r in range(len(self.devices)):
            ms = (ctypes.c_double * (len(MULTI_PHASES) + 1))()
            self._chk(lib().icikt_multi_rank_phase_ms(self._h, r, ms), "icikt_multi_rank_phase_ms")
            out.append(dict(zip(MULTI_PHASES + ("wait",), list(ms))))
        return out

    @property
    def ranks_used(self) -> int:
        """Ranks the last call really used (1: too small to split, or wide columns: the first device alone)."""
        return int(lib().icikt_multi_ranks_used(self._h))

    def matrix(self, X, global_na=None, pi=None, pj=None, perspective="global", alternative="two.sided",
               continuity=False, flags: int = 0, scale_max=True, diag_good=True, want_keep=True):
        """Same contract as Context.matrix()."""
        return _matrix_call(lib().icikt_matrix_multi_f64, self._h, self._chk, X, global_na, pi, pj, perspective,
                            alternative, continuity, flags, scale_max, diag_good, want_keep)

    def pairs(self, X, pi=None, pj=None, perspective="global", alternative="two.sided", continuity=False,
              flags: int = 0, want_counts: bool = True):
        """Same contract as Context.pairs()."""
        Xf = np.asfortranarray(X, dtype=np.float64)
        if Xf.ndim != 2:
            raise ValueError("X must be 2-D (features x samples)")
        n_feat, n_samp = Xf.shape
        if pi is None:
            P = n_samp * (n_samp - 1) // 2
            pi_a = pj_a = None
        else:
            pi_a = np.ascontiguousarray(pi, dtype=np.int32)
            pj_a = np.ascontiguousarray(pj, dtype=np.int32)
            P = pi_a.shape[0]
        alloc = pinned_empty if (flags & FLAG_HOST_PINNED) else np.empty   # (X is the caller's: pinned by the caller)
        out = alloc((P, 4), dtype=np.float64)
        cnt = alloc((P, len(CNT_FIELDS)), dtype=np.int64) if want_counts else None
        rsn = alloc(P, dtype=np.int32)
        if cnt is not None:
            cnt[...] = 0
        rsn[...] = 0
        alt = ALTERNATIVE.get(alternative, ALT_OTHER)
        self._chk(lib().icikt_pairs_multi_f64(self._h, _ptr(Xf), n_feat, n_samp, max(n_feat, 0), _ptr(pi_a), _ptr(pj_a),
                                              P, PERSPECTIVE[perspective], alt, int(bool(continuity)), flags, _ptr(out),
                                              _ptr(cnt), _ptr(rsn)), "icikt_pairs_multi_f64")
        return out, cnt, rsn


_default_ctx: dict[int, Context] = {}


def default_context(device: int | None = None) -> Context:
    if device is None:
        device = int(os.environ.get("ICIKT_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        n = device_count()
        if n > 0:
            device %= n
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]
