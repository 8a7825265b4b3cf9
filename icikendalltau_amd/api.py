"""Host-side mirror of the reference's R interface for the all-pairs ICI-Kendall-tau path.

R is not available in the build image, so the host logic the reference keeps in R
(R/kendalltau.R:96-308,357-421,563-629; R/utils.R:1-66) is mirrored here in Python with the same
function names, argument names, defaults, return shapes and error / warning texts, on top of the
C ABI (include/icikt.h).  The R glue a maintainer would use instead lives in icikendalltau_amd/r/.

Every number is computed by the HIP kernels through ``engine`` (default: the MI355X engine, which
raises when no GPU is usable -- there is no CPU fallback).  ``engine`` exists so the sharding /
gather / reshape logic can be exercised in CPU-only tests with a checker engine.
"""
from __future__ import annotations

import math
import time
import warnings
from typing import Sequence

import numpy as np

from . import _lib

try:  # pandas is the natural carrier of R's data.frame / dimnames; numpy-only callers still work
    import pandas as pd
except Exception:  # pragma: no cover
    pd = None


# --------------------------------------------------------------------------------------------------
# engines
# --------------------------------------------------------------------------------------------------
class HipEngine:
    """Runs pair lists on one MI355X through libicikt_hip.so."""

    name = "hip"

    def __init__(self, device: int | None = None, exact_int64: bool = False):
        self.ctx = _lib.default_context(device)
        self.flags = _lib.FLAG_EXACT_INT64 if exact_int64 else 0

    def pairs(self, X, pi, pj, perspective, alternative, continuity):
        out, _cnt, rsn = self.ctx.pairs(X, pi, pj, perspective, alternative, continuity, self.flags,
                                        want_counts=False)
        return out, rsn

    def pairs_counts(self, X, pi, pj, perspective, alternative, continuity):
        """pairs() plus the integer counts record (dict of arrays keyed by _lib.CNT_FIELDS)."""
        out, cnt, rsn = self.ctx.pairs(X, pi, pj, perspective, alternative, continuity, self.flags, want_counts=True)
        return out, rsn, {k: cnt[:, i] for i, k in enumerate(_lib.CNT_FIELDS)}

    def missingness(self, X, pi, pj):
        return self.ctx.missingness(X, pi, pj)

    def matrix(self, data_matrix, global_na, pi, pj, perspective, alternative, continuity, scale_max, diag_good):
        """The whole of ici_kendalltau() below its argument checks in one library call (icikt_matrix_f64 /
        icikt_matrix_multi_f64): exclusion rule, pair kernels and scale_and_reshape on the device.  Returns
        (out5 [5, S, S]: cor, raw, pvalue, taumax, completeness; keep [S, n_feat] bool; pairs per reason code [5])."""
        return self.ctx.matrix(data_matrix, global_na, pi, pj, perspective, alternative, continuity, self.flags,
                               scale_max, diag_good, want_keep=True)

    def pairs_complete(self, X, pi, pj):
        """kt_fast(use = "pairwise.complete.obs") on the device: (out4, reasons)."""
        out, _cnt, rsn = self.ctx.pairs_complete(X, pi, pj, "two.sided", False, self.flags)
        return out, rsn

    def pairs_block_dev(self, X, pi, pj, begin, end, n_each, perspective, alternative, continuity, dist, device,
                        via_host):
        """This rank's block [begin, end) of a pair list under torch.distributed: column-sharded pre-pass
        (sharding.ShardedPrepass; every rank falls back to the whole pre-pass together if it cannot be set up), then
        K1 + K2 into torch buffers on `device`, padded to n_each rows for the gather."""
        import torch
        from . import sharding
        ctx = self.ctx
        Xf = np.asfortranarray(X, dtype=np.float64)
        S = Xf.shape[1]
        with torch.cuda.device(device):
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)  # one stream orders kernels and collectives
            try:
                sp = sharding.ShardedPrepass(ctx, dist, device, via_host)
                if not sp.setup(S, lambda c0, c1, alloc, fl: ctx.prepare_cols(Xf, c0, c1, alloc, fl),
                                sync=torch.cuda.synchronize):
                    ctx.prepare_cols(Xf, 0, S, S, 0)
                self.pre_pass = sp.mode
                out_l = torch.full((n_each, 4), float("nan"), dtype=torch.float64, device=device)
                rsn_l = torch.zeros(n_each, dtype=torch.int32, device=device)
                ctx.set_pairs(pi[begin:end], pj[begin:end])
                if end > begin:
                    ctx.run_dev(_lib.PERSPECTIVE[perspective], _lib.ALTERNATIVE.get(alternative, _lib.ALT_OTHER),
                                continuity, self.flags, out_l.data_ptr(), None, rsn_l.data_ptr())
                torch.cuda.synchronize()
            finally:
                ctx.use_own_stream()
        return out_l, rsn_l


class MultiHipEngine(HipEngine):
    """Runs pair lists on several MI355X behind ONE library call (icikt_pairs_multi_f64): one host thread per
    GPU inside the library, column-sharded pre-pass, RCCL all-gather / gather over xGMI.  This is what an R
    caller's `n_gpu` argument selects (icikendalltau_amd/r/icikt_mi355x.R); with torch.distributed ranks use
    HipEngine per rank instead."""

    name = "hip-multi"

    def __init__(self, devices=None, n_gpu: int | None = None, exchange: str = "auto", exact_int64: bool = False):
        self.ctx = _lib.MultiContext(devices, n_gpu, exchange)
        self._single = None
        self.flags = _lib.FLAG_EXACT_INT64 if exact_int64 else 0

    def _one(self):
        if self._single is None:
            self._single = _lib.default_context(self.ctx.devices[0])
        return self._single

    def missingness(self, X, pi, pj):  # a bitset popcount: one device is plenty
        return self._one().missingness(X, pi, pj)

    def pairs_complete(self, X, pi, pj):  # kt_fast's per-pair masking path exists on one device only
        out, _cnt, rsn = self._one().pairs_complete(X, pi, pj, "two.sided", False, self.flags)
        return out, rsn


def _default_engine():
    return HipEngine()


_multi_engines: dict = {}


def _multi_engine(n_gpu, devices):
    """One MultiHipEngine per device list: RCCL communicators are expensive to create."""
    devs = tuple(int(d) for d in devices) if devices is not None else tuple(range(int(n_gpu)))
    if devs not in _multi_engines:
        _multi_engines[devs] = MultiHipEngine(devices=list(devs))
    return _multi_engines[devs]


# --------------------------------------------------------------------------------------------------
# ici_kt: one pair (R/RcppExports.R:62-64 -> src/kendallc.cpp:166)
# --------------------------------------------------------------------------------------------------
class IciKtResult(tuple):
    """Named numeric(4): c(tau, pvalue, tau_max, completeness) (src/kendallc.cpp:171-172)."""

    names = ("tau", "pvalue", "tau_max", "completeness")

    def __new__(cls, values):
        return super().__new__(cls, (float(v) for v in values))

    def __getitem__(self, key):
        if isinstance(key, str):
            return super().__getitem__(self.names.index(key))
        return super().__getitem__(key)

    tau = property(lambda self: self[0])
    pvalue = property(lambda self: self[1])
    tau_max = property(lambda self: self[2])
    completeness = property(lambda self: self[3])

    def __repr__(self):
        return "IciKtResult(" + ", ".join(f"{n}={v!r}" for n, v in zip(self.names, self)) + ")"


def _warn_reason(reason: int):
    msg = _lib.REASON_WARNINGS.get(int(reason))
    if msg:
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


def ici_kt(x, y, perspective="local", alternative="two.sided", continuity=False, output="simple", engine=None):
    """Information-content-informed Kendall tau-b of two vectors (NaN = missing).

    Same defaults as the reference (src/kendallc.cpp:166).  Any ``output`` other than "simple" prints the
    reference's report (src/kendallc.cpp:342-363: same labels, ``std::to_string`` formatting) from the counts
    record of the pair; ``ici_kt_counts`` returns those integers instead of printing them.
    Vectors of up to 65 535 rows (``_lib.MAX_FEATURES``) run the tuned kernels; up to 262 144
    (``_lib.MAX_FEATURES_WIDE``) a plain 32-bit path in exact integer arithmetic; longer ones are refused with an
    error that names the limit.
    """
    x = np.asarray(x, dtype=np.float64).ravel()
    y = np.asarray(y, dtype=np.float64).ravel()
    if x.shape[0] != y.shape[0]:
        raise ValueError("'X' and 'Y' are not the same length!")  # src/kendallc.cpp:168-170
    if perspective not in ("local", "global"):
        # the reference treats any string other than "local" as global (kendallc.cpp:180)
        perspective = "global"
    eng = engine or _default_engine()
    X = np.empty((x.shape[0], 2), dtype=np.float64, order="F")
    X[:, 0] = x
    X[:, 1] = y
    i0, i1 = np.array([0], np.int32), np.array([1], np.int32)
    if output != "simple" and hasattr(eng, "pairs_counts"):
        out, rsn, cnt = eng.pairs_counts(X, i0, i1, perspective, alternative, continuity)
        if rsn[0] == 0:  # the reference returns before its report in every NA case
            print(_ici_kt_report(x, y, perspective, continuity, out[0], {k: int(v[0]) for k, v in cnt.items()}), end="")
    else:
        out, rsn = eng.pairs(X, i0, i1, perspective, alternative, continuity)
    _warn_reason(rsn[0])
    return IciKtResult(out[0])


def _na_rm(a: np.ndarray) -> np.ndarray:
    """The non-missing values of `a`.  NOT np.nanmax / np.nanmin: R's NA_real_ -- the pattern the library returns for a
    pair without a result, and what a matrix read from an .rda file holds -- is a SIGNALLING NaN, and the C fmax / fmin
    behind numpy's NaN-skipping reductions answer NaN for one (the running maximum is lost wherever such a cell meets
    the scalar tail of the reduction)."""
    a = np.asarray(a)
    return a[~np.isnan(a)]


def _ici_kt_report(x, y, perspective, continuity, out4, k) -> str:
    """The text ici_kt(output != "simple") writes to Rcout (src/kendallc.cpp:342-363), rebuilt from the counts
    record: integers print as integers, doubles as std::to_string prints them ("%f")."""
    keep = ~(np.isnan(x) & np.isnan(y)) if perspective == "local" else np.ones(x.shape[0], bool)
    x2, y2 = x[keep].copy(), y[keep].copy()
    min_x, min_y = _na_rm(x2).min() - 0.1, _na_rm(y2).min() - 0.1    # :214-215
    x2[np.isnan(x2)] = min_x
    y2[np.isnan(y2)] = min_y
    sum_obs = len(np.unique(np.stack([x2, y2], axis=1), axis=0)) + 1  # joint runs + 1 (:261-263)
    n, tot = k["n"], k["tot"]
    m = n * (n - 1)
    ld = np.longdouble
    con_minus_dis = ld(tot) - k["xtie"] - k["ytie"] + k["ntie"] - 2 * k["dis"]
    var = (ld(m * (2 * n + 5)) - k["x1"] - k["y1"]) / 18 + ld(2.0 * k["xtie"] * k["ytie"]) / m + \
        ld(float(k["x0"]) * float(k["y0"])) / ld(9 * m * (n - 2)) if n > 2 else ld("nan")
    s_adj = ld(out4[0]) * np.sqrt(ld(m // 2 - k["xtie"]) * ld(m // 2 - k["ytie"]))
    if continuity:
        s_adj = np.sign(s_adj) * (abs(s_adj) - 1)
    z_b = s_adj / np.sqrt(var)
    f = lambda v: f"{float(v):f}"  # noqa: E731
    rows = [("min_x: ", f(min_x)), ("min_y: ", f(min_y)), ("n_entry: ", str(n)), ("missingness: ", str(k["missing"])),
            ("completeness: ", f(out4[3])), ("tot: ", str(tot)), ("sum_obs: ", str(sum_obs)), ("dis: ", str(k["dis"])),
            ("con_minus_dis (k_numerator): ", f(con_minus_dis)), ("n_tie: ", f(k["ntie"])), ("m: ", str(m)),
            ("x_tie: ", f(k["xtie"])), ("y_tie: ", f(k["ytie"])), ("s_adjusted: ", f(s_adj)), ("var: ", f(var)),
            ("z_b: ", f(z_b)), ("tau: ", f(out4[0])), ("tau_max:", f(out4[2])), ("pvalue: ", f(out4[1]))]
    return "".join(a + b + "\n" for a, b in rows)


def ici_kt_counts(x, y, perspective="local", device=None, exact_int64=False):
    """The integer counts ici_kt(..., output != "simple") prints (src/kendallc.cpp:342-363)."""
    ctx = _lib.default_context(device)
    out, cnt, rsn = ctx.pair(x, y, perspective, flags=_lib.FLAG_EXACT_INT64 if exact_int64 else 0)
    return cnt, rsn


# --------------------------------------------------------------------------------------------------
# input checks (R/utils.R:1-66)
# --------------------------------------------------------------------------------------------------
def _as_matrix(data_matrix, colnames, arg):
    """check_if_colnames_null / transform_to_matrix / check_if_numeric."""
    if pd is not None and isinstance(data_matrix, pd.DataFrame):
        print(f"i `{arg}` is a data.frame, converting to matrix ...")  # R/utils.R:53-57
        if colnames is None:
            colnames = [str(c) for c in data_matrix.columns]
        data_matrix = data_matrix.to_numpy()
    arr = np.asarray(data_matrix)
    if colnames is None:
        raise ValueError(f"Colnames of `{arg}` must be be specified.")  # R/utils.R:29-34 (sic)
    if arr.dtype.kind not in "fiu":
        raise TypeError(f"`{arg}` must be a numeric type.")  # R/utils.R:42-47
    if arr.ndim != 2:
        raise ValueError(f"`{arg}` must be a 2-D matrix (features x samples)")
    colnames = [str(c) for c in colnames]
    if len(colnames) != arr.shape[1]:
        raise ValueError("length of colnames does not match the number of columns")
    return np.asarray(arr, dtype=np.float64), colnames


def setup_missing_matrix(data_matrix: np.ndarray, global_na) -> np.ndarray:
    """Logical exclude_loc (R/utils.R:1-23), in the memory order of the matrix (no transposing copies)."""
    vals = [] if global_na is None else list(np.atleast_1d(np.asarray(global_na, dtype=np.float64)))
    has_nan = any(math.isnan(v) for v in vals)
    has_inf = any(math.isinf(v) for v in vals)
    vals = [v for v in vals if not (math.isnan(v) or math.isinf(v))]
    if has_nan and has_inf:
        exclude = ~np.isfinite(data_matrix)  # one pass for NA and Inf
    elif has_nan:
        exclude = np.isnan(data_matrix)
    elif has_inf:
        exclude = np.isinf(data_matrix)
    else:
        exclude = np.zeros_like(data_matrix, dtype=bool)
    for v in vals:
        with np.errstate(invalid="ignore"):
            exclude |= (data_matrix == v)
    return exclude


def _masked_fortran(data_matrix: np.ndarray, exclude_loc: np.ndarray) -> np.ndarray:
    """exclude_data[exclude_loc] = NA (R/kendalltau.R:120-121) as a column-major float64 matrix."""
    out = np.array(data_matrix, dtype=np.float64, order="F", copy=True)
    out[exclude_loc] = np.nan
    return out


# --------------------------------------------------------------------------------------------------
# setup_comparisons (R/kendalltau.R:181-278)
# --------------------------------------------------------------------------------------------------
def _is_vector_like(obj):
    if isinstance(obj, (str, int, float, np.integer, np.floating)):
        return True
    if isinstance(obj, np.ndarray):
        return obj.ndim <= 1
    if isinstance(obj, (list, tuple)):
        return all(isinstance(v, (str, int, float, np.integer, np.floating)) for v in obj)
    return False


def _r_str(v) -> str:
    if isinstance(v, (float, np.floating)) and float(v).is_integer():
        return str(int(v))  # as.character(1) == "1"
    return str(v)


def _recycle(a: Sequence, n: int):
    return [a[i % len(a)] for i in range(n)]


def setup_comparisons(samples, include_only=None, diag_good=True, ncore=1, include_arg="include_only"):
    """Pair list in utils::combn order with include_only filtering; returns (i, j, core) 0-based arrays."""
    n_sample = len(samples)
    iu, ju = np.triu_indices(n_sample, k=1)  # row-major upper triangle == combn(n, 2) order
    pi = iu.astype(np.int32)
    pj = ju.astype(np.int32)
    if not diag_good:  # self comparisons appended after all pairs (R/kendalltau.R:191-194)
        d = np.arange(n_sample, dtype=np.int32)
        pi = np.concatenate([pi, d])
        pj = np.concatenate([pj, d])
    names = np.asarray(samples, dtype=object)

    if include_only is not None:
        if pd is not None and isinstance(include_only, pd.DataFrame):
            include_only = [include_only[c].tolist() for c in include_only.columns]
        elif isinstance(include_only, dict):
            include_only = list(include_only.values())
        if _is_vector_like(include_only):
            inc = {_r_str(v) for v in np.atleast_1d(np.asarray(include_only, dtype=object))}
            col_in = np.fromiter((nm in inc for nm in names), dtype=bool, count=n_sample)   # per SAMPLE, not per pair
            keep = col_in[pi] | col_in[pj]  # R/kendalltau.R:210-212
        elif isinstance(include_only, (list, tuple)):
            if len(include_only) == 2:
                l1 = [_r_str(v) for v in np.atleast_1d(np.asarray(include_only[0], dtype=object))]
                l2 = [_r_str(v) for v in np.atleast_1d(np.asarray(include_only[1], dtype=object))]
                m = max(len(l1), len(l2))
                l1r, l2r = _recycle(l1, m), _recycle(l2, m)  # paste0 recycles
                # "a-b" strings of the reference (:213-229) as index pairs: a listed name that is no sample matches
                # nothing; names holding "-" could make two different name pairs paste to one string -- those lists
                # keep the reference's string comparison
                if any("-" in str(nm) for nm in names) or any("-" in v for v in l1r + l2r):
                    allowed = {f"{a}-{b}" for a, b in zip(l1r, l2r)} | {f"{b}-{a}" for a, b in zip(l1r, l2r)}
                    keep = np.fromiter((f"{names[i]}-{names[j]}" in allowed for i, j in zip(pi, pj)), dtype=bool,
                                       count=len(pi))
                else:
                    index = {str(nm): k for k, nm in reversed(list(enumerate(names)))}   # first occurrence wins
                    ia = np.array([index.get(a, -1) for a in l1r], dtype=np.int64)
                    ib = np.array([index.get(b, -1) for b in l2r], dtype=np.int64)
                    ok = (ia >= 0) & (ib >= 0)
                    ia, ib = ia[ok], ib[ok]
                    codes = np.concatenate([ia * n_sample + ib, ib * n_sample + ia])
                    keep = np.isin(pi.astype(np.int64) * n_sample + pj, codes)
            else:
                raise ValueError(
                    f"`{include_arg}` must be a vector, a data.frame with two columns, or list of two vectors. "
                    f"Currently, `length({include_arg})` returns {len(include_only)}")  # R/kendalltau.R:230-236
        else:
            raise ValueError(f"`{include_arg}` must be a vector, a data.frame with two columns, or list of two vectors.")
        pi, pj = pi[keep], pj[keep]

    n_todo = len(pi)
    if n_todo == 0:
        raise ValueError("No comparisons to do. Check the list of column names in "
                         f"`{include_arg}` vs those in the samples.")  # R/kendalltau.R:240-247
    n_each = int(math.ceil(n_todo / ncore))
    core = (np.arange(n_todo) // n_each + 1).astype(np.int32)  # rep(seq(1, ncore), each = n_each)
    return pi, pj, core


# --------------------------------------------------------------------------------------------------
# sharding over ranks: the reference's `core` chunks (R/kendalltau.R:250-255) become GPU ranks
# --------------------------------------------------------------------------------------------------
def _dist_info():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist, dist.get_rank(), dist.get_world_size()
    except Exception:
        pass
    return None, 0, 1


def _run_sharded(engine, X, pi, pj, core, perspective, alternative, continuity):
    """Each rank computes the pairs whose `core` is rank+1, then all ranks gather everything.

    The HIP engine runs the whole sharded flow on the device (icikendalltau_amd/sharding.py): a rank uploads and
    sorts only its share of the columns, `order` + `meta` are all-gathered, the pair block runs, and the padded
    result blocks are all-gathered -- every buffer on the ENGINE's device (LOCAL_RANK-derived), never on torch's
    ambient current device.  Any other engine (the checker of the CPU tests) computes its block through
    engine.pairs() and the same gather.
    """
    dist, rank, world = _dist_info()
    if world == 1:
        return engine.pairs(X, pi, pj, perspective, alternative, continuity)
    import torch
    from . import sharding
    begin, end, n_each = sharding.pair_block(len(pi), rank, world)
    assert np.array_equal(np.nonzero(core == rank + 1)[0], np.arange(begin, end)), "core chunks are consecutive blocks"
    via_host = dist.get_backend() != "nccl"
    if hasattr(engine, "pairs_block_dev"):
        device = torch.device("cuda", engine.ctx.device)
        out_l, rsn_l = engine.pairs_block_dev(X, pi, pj, begin, end, n_each, perspective, alternative, continuity,
                                              dist, device, via_host)
    else:
        device = torch.device("cpu")
        out_l = torch.full((n_each, 4), float("nan"), dtype=torch.float64)
        rsn_l = torch.zeros(n_each, dtype=torch.int32)
        if end > begin:
            o, r = engine.pairs(X, pi[begin:end], pj[begin:end], perspective, alternative, continuity)
            out_l[:end - begin] = torch.from_numpy(np.ascontiguousarray(o))
            rsn_l[:end - begin] = torch.from_numpy(np.ascontiguousarray(r, dtype=np.int32))
    g_out = sharding.gather_blocks(dist, out_l, n_each, device, via_host or device.type == "cpu", to_all=True)
    g_rsn = sharding.gather_blocks(dist, rsn_l, n_each, device, via_host or device.type == "cpu", to_all=True)
    out = sharding.assemble(g_out, len(pi), n_each).cpu().numpy()
    rsn = sharding.assemble(g_rsn, len(pi), n_each).cpu().numpy()
    return out, rsn


# --------------------------------------------------------------------------------------------------
# ici_kendalltau (R/kendalltau.R:96-179) + scale_and_reshape (:357-421)
# --------------------------------------------------------------------------------------------------
def _named_matrix(values: np.ndarray, names):
    if pd is not None:
        return pd.DataFrame(values, index=list(names), columns=list(names))
    return values


def ici_kendalltau(data_matrix, global_na=(float("nan"), float("inf"), 0), perspective="global", scale_max=True,
                   diag_good=True, include_only=None, alternative="two.sided", continuity=False,
                   check_timing=False, return_matrix=True, colnames=None, engine=None, n_gpu=1, devices=None):
    """All-pairs ICI-Kendall-tau between the COLUMNS (samples) of a features x samples matrix.

    ``n_gpu`` > 1 (or ``devices``) uses several MI355X behind one library call (icikt_pairs_multi_f64: a host
    thread per GPU, RCCL all-gather / gather over xGMI) -- the role furrr workers play in the reference
    (R/kendalltau.R:127,158): the `core` column then numbers the GPUs' pair blocks.  Under an initialised
    torch.distributed process group the ranks are the cores instead (one process per GPU).

    Mirrors R/kendalltau.R:96-179: same argument names and defaults; returns a dict with ``cor, raw,
    pvalue, taumax, completeness`` (samples x samples, un-computed cells 0), ``keep`` and ``run_time``,
    or with ``return_matrix=False`` ``{"cor": data.frame(s1, s2, core, raw, pvalue, taumax,
    completeness, cor), "run_time"}``.  Column names are required, as in the reference (``colnames=``
    for a bare ndarray).
    """
    data_matrix, names = _as_matrix(data_matrix, colnames, "data_matrix")
    n_sample = data_matrix.shape[1]

    _dist, _rank, world = _dist_info()
    ncore = world
    if engine is None and world == 1 and (n_gpu > 1 or devices is not None):
        engine = _multi_engine(n_gpu, devices)
        ncore = len(engine.ctx.devices)
    eng = engine or _default_engine()

    # the device-side exclusion rule holds up to MASK_VALS distinct finite values; the reference loops over any number
    # (R/utils.R:16-20): a longer list takes the host-masking route below
    n_finite = len({float(v) for v in np.atleast_1d(np.asarray([] if global_na is None else global_na, dtype=np.float64))
                    if math.isfinite(v)})
    if return_matrix and not check_timing and world == 1 and hasattr(eng, "matrix") and n_finite <= _lib.MASK_VALS:
        # One library call does everything below the argument checks on the device (icikt_matrix_f64): the exclusion
        # rule of setup_missing_matrix inside the pre-pass (no masked copy of the matrix on the host), the pair
        # kernels, scale_and_reshape, one copy of the five matrices back.  All pairs of the upper triangle need no
        # pair list at all; include_only / diag_good = FALSE hand over the filtered list.
        if include_only is None and diag_good:
            if n_sample < 2:
                raise ValueError("No comparisons to do. Check the list of column names in "
                                 "`include_only` vs those in the samples.")  # R/kendalltau.R:240-247
            pi = pj = None
        else:
            pi, pj, _core = setup_comparisons(names, include_only, diag_good, ncore=ncore)
        t1 = time.perf_counter()
        out5, keep, rcounts = eng.matrix(data_matrix, global_na, pi, pj, perspective, alternative, continuity,
                                         scale_max, diag_good)
        t_diff = time.perf_counter() - t1
        for code in (_lib.REASON_SHORT, _lib.REASON_SINGLE_UNIQUE, _lib.REASON_TIES_EQ_TOTAL):
            for _ in range(int(rcounts[code])):   # one warning per offending pair, as ici_split raises them
                _warn_reason(code)
        res = {key: _named_matrix(out5[k], names)
               for k, key in enumerate(("cor", "raw", "pvalue", "taumax", "completeness"))}
        res["keep"] = keep
        res["run_time"] = t_diff
        return res

    exclude_loc = setup_missing_matrix(data_matrix, global_na)
    exclude_data = _masked_fortran(data_matrix, exclude_loc)
    pi, pj, core = setup_comparisons(names, include_only, diag_good, ncore=ncore)
    n_todo = len(pi)

    if check_timing:  # R/kendalltau.R:141-148, 633-669
        rng = np.random.default_rng()
        first = np.nonzero(core == 1)[0]
        pick = rng.choice(first, size=min(5, len(first)), replace=False)
        t0 = time.perf_counter()
        for p in pick:
            eng.pairs(exclude_data, pi[p:p + 1], pj[p:p + 1], perspective, "two.sided", False)
        t_total = time.perf_counter() - t0
        n_comp = len(pick)
        t_each = t_total / n_comp
        t_theoretical = t_each * n_todo
        t_cores = t_theoretical / ncore
        which = ["n_tested", "n_todo", "time_tested", "time_single", "time_all", "time_across_cores",
                 "time_minutes", "time_hours", "time_days"]
        value = [n_comp, n_todo, t_total, t_each, t_theoretical, t_cores, t_cores / 60, t_cores / (60 * 60),
                 t_cores / (60 * 60 * 60)]  # (sic) the reference divides by 60^3 for days
        return pd.DataFrame({"which": which, "value": value}) if pd is not None else dict(zip(which, value))

    t1 = time.perf_counter()
    out, rsn = _run_sharded(eng, exclude_data, pi, pj, core, perspective, alternative, continuity)
    t_diff = time.perf_counter() - t1
    for r in rsn[rsn > 1]:
        _warn_reason(r)

    raw, pvalue, taumax, completeness = out[:, 0], out[:, 1], out[:, 2], out[:, 3]
    n_good = (exclude_loc.shape[0] - exclude_loc.sum(axis=0)).astype(np.float64)
    frac_complete = n_good / exclude_loc.shape[0]

    # scale_and_reshape
    if scale_max:
        have = _na_rm(taumax)                                           # max(taumax, na.rm = TRUE)
        max_cor = have.max() if have.size else -np.inf                  # max(numeric(0)) is -Inf in R (and in k_assemble)
        cor = raw / max_cor
    else:
        cor = raw.copy()
    s1, s2 = pi.copy(), pj.copy()
    core_col = core.astype(np.float64)
    if diag_good:
        d = np.arange(n_sample, dtype=np.int32)
        s1 = np.concatenate([s1, d])
        s2 = np.concatenate([s2, d])
        core_col = np.concatenate([core_col, np.zeros(n_sample)])
        diag = n_good / n_good.max()
        raw = np.concatenate([raw, diag])
        pvalue = np.concatenate([pvalue, np.zeros(n_sample)])
        taumax = np.concatenate([taumax, np.ones(n_sample)])
        completeness = np.concatenate([completeness, frac_complete])
        cor = np.concatenate([cor, diag])

    if return_matrix:
        res = {}
        for key, vals in (("cor", cor), ("raw", raw), ("pvalue", pvalue), ("taumax", taumax),
                          ("completeness", completeness)):
            m = np.zeros((n_sample, n_sample))
            m[s1, s2] = vals
            m[s2, s1] = vals
            res[key] = _named_matrix(m, names)
        res["keep"] = (~exclude_loc).T
        res["run_time"] = t_diff
        return res
    names_arr = np.asarray(names, dtype=object)
    cols = {"s1": names_arr[s1], "s2": names_arr[s2], "core": core_col, "raw": raw, "pvalue": pvalue,
            "taumax": taumax, "completeness": completeness, "cor": cor}
    return {"cor": pd.DataFrame(cols) if pd is not None else cols, "run_time": t_diff}


# --------------------------------------------------------------------------------------------------
# pairwise_completeness (R/kendalltau.R:563-629)
# --------------------------------------------------------------------------------------------------
def pairwise_completeness(data_matrix, global_na=(float("nan"), float("inf"), 0), include_only=None,
                          return_matrix=True, colnames=None, engine=None):
    data_matrix, names = _as_matrix(data_matrix, colnames, "data_matrix")
    exclude_loc = setup_missing_matrix(data_matrix, global_na)
    _dist, _rank, world = _dist_info()
    pi, pj, core = setup_comparisons(names, include_only, diag_good=False, ncore=world)
    eng = engine or _default_engine()
    masked = np.asfortranarray(np.where(exclude_loc, np.nan, 0.0))
    missingness = eng.missingness(masked, pi, pj).astype(np.float64)
    completeness = 1 - (missingness / exclude_loc.shape[0])
    if return_matrix:
        m = np.zeros((len(names), len(names)))
        m[pi, pj] = completeness
        m[pj, pi] = completeness
        return _named_matrix(m, names)
    names_arr = np.asarray(names, dtype=object)
    cols = {"s1": names_arr[pi], "s2": names_arr[pj], "core": core.astype(np.float64), "missingness": missingness,
            "completeness": completeness}
    return pd.DataFrame(cols) if pd is not None else cols


# --------------------------------------------------------------------------------------------------
# kt_fast (R/kendalltau.R:448-545) and kt_split (:310-354) -- SURVEY.md section 8(f) row 3
# --------------------------------------------------------------------------------------------------
_KT_USE = ("all.obs", "complete.obs", "pairwise.complete.obs", "everything", "na.or.complete")


def _match_use(use: str) -> str:
    """R's match.arg(): exact or unique partial match."""
    hits = [u for u in _KT_USE if u == use] or [u for u in _KT_USE if u.startswith(use)]
    if len(hits) != 1:
        raise ValueError("'arg' should be one of " + ", ".join(f"'{u}'" for u in _KT_USE))
    return hits[0]


def kt_fast(x, y=None, use="everything", alternative="two.sided", continuity=False, return_matrix=True,
            colnames=None, engine=None, max_pair_chunk=4096):
    """Kendall tau-b with stats::cor-like NA policies over the ici_kt kernel (R/kendalltau.R:448-545).

    As in the reference, ``alternative`` and ``continuity`` are accepted but NOT forwarded (kt_split calls
    ici_kt with its defaults, :342), and self comparisons are part of the result.  ``use``:
    "everything"/"all.obs": any NA anywhere in ``x`` -> every entry NA; "complete.obs": rows with an NA in
    any column are dropped first; "pairwise.complete.obs": per pair, rows with an NA in either vector are
    dropped.  The last one runs on the GPU as ici_kt(..., perspective = "local") of the two vectors with both
    entries of such rows set missing -- identical to dropping the rows, since "local" removes rows missing in
    both (src/kendallc.cpp:180-185) and nothing missing remains; the HIP engine masks, sorts and counts every
    pair on the device (icikt_pairs_complete_f64), other engines get the masked vectors from the host.
    """
    na_method = _match_use(use)
    if na_method == "na.or.complete":
        raise ValueError("'na.or.complete' is not a supported value for `use`. "
                         "Please use one of all.obs complete.obs pairwise.complete everthing.")  # R/utils.R:86-90 (sic)
    if y is None:
        if not (pd is not None and isinstance(x, pd.DataFrame)) and np.ndim(x) < 2:
            raise ValueError("`x` and `y` should both be provided as vectors, or `x` should be matrix-like.")
        X, names = _as_matrix(x, colnames, "x")
    else:
        if np.ndim(x) > 1 or np.ndim(y) > 1:
            raise ValueError("Both `x` and `y` must be vectors.")
        X = np.column_stack([np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)])
        names = list(colnames) if colnames is not None else ["x", "y"]
    eng = engine or _default_engine()
    _dist, _rank, world = _dist_info()
    pi, pj, core = setup_comparisons(names, None, diag_good=False, ncore=world)
    P = len(pi)
    tau = np.full(P, np.nan)
    pvalue = np.full(P, np.nan)
    na_vals = np.isnan(X)
    t_diff = 0.0
    do_computation = True
    if na_method in ("everything", "all.obs") and na_vals.any():
        do_computation = False
    if na_method == "complete.obs":
        keep = ~na_vals.any(axis=1)
        if keep.sum() == 0:
            do_computation = False
        else:
            X = X[keep]
    if do_computation:
        t1 = time.perf_counter()
        if na_method == "pairwise.complete.obs" and np.isnan(X).any() and hasattr(eng, "pairs_complete"):
            # masking, per-pair sorts and counting on the device (icikt_pairs_complete_f64)
            out, rsn = eng.pairs_complete(np.asfortranarray(X), pi, pj)
            for r in rsn[rsn > 1]:
                _warn_reason(r)
            tau, pvalue = out[:, 0].copy(), out[:, 1].copy()
        elif na_method == "pairwise.complete.obs" and np.isnan(X).any():
            na = np.isnan(X)
            for b in range(0, P, max_pair_chunk):
                sl = slice(b, min(P, b + max_pair_chunk))
                m = sl.stop - sl.start
                either = na[:, pi[sl]] | na[:, pj[sl]]
                Xp = np.empty((X.shape[0], 2 * m), dtype=np.float64, order="F")
                Xp[:, 0::2] = np.where(either, np.nan, X[:, pi[sl]])
                Xp[:, 1::2] = np.where(either, np.nan, X[:, pj[sl]])
                idx = np.arange(m, dtype=np.int32)
                out, rsn = eng.pairs(Xp, 2 * idx, 2 * idx + 1, "local", "two.sided", False)
                for r in rsn[rsn > 1]:
                    _warn_reason(r)
                tau[sl], pvalue[sl] = out[:, 0], out[:, 1]
        else:
            out, rsn = eng.pairs(np.asfortranarray(X), pi, pj, "local", "two.sided", False)
            for r in rsn[rsn > 1]:
                _warn_reason(r)
            tau, pvalue = out[:, 0].copy(), out[:, 1].copy()
        t_diff = time.perf_counter() - t1
    if return_matrix:
        S = len(names)
        tm, pm = np.zeros((S, S)), np.zeros((S, S))
        tm[pi, pj] = tau
        tm[pj, pi] = tau
        pm[pi, pj] = pvalue
        pm[pj, pi] = pvalue
        return {"tau": _named_matrix(tm, names), "pvalue": _named_matrix(pm, names), "run_time": t_diff}
    names_arr = np.asarray(names, dtype=object)
    cols = {"s1": names_arr[pi], "s2": names_arr[pj], "core": core.astype(np.float64), "tau": tau, "pvalue": pvalue}
    return {"tau": pd.DataFrame(cols) if pd is not None else cols, "run_time": t_diff}
