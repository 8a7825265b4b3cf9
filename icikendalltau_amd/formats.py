"""Input / output formats either side of the path.

* ``read_r_data(path)``: numeric matrices and vectors out of R's ``.rda`` / ``.RData`` / ``.rds`` files
  (XDR serialisation, gzip / bzip2 / xz / uncompressed), e.g. the reference's ``data/yeast_missing.rda``.
  A pure data reader: it understands the handful of SEXP types a numeric matrix with dimnames needs and
  evaluates nothing from the file.
* ``cor_matrix_2_long_df`` / ``long_df_2_cor_matrix``: the reference's converters between the square result
  matrices and the long data.frame storage form (R/reshaping.R:15-68).
"""
from __future__ import annotations

import bz2
import gzip
import lzma
import struct

import numpy as np

try:
    import pandas as pd
except Exception:  # pragma: no cover
    pd = None

# SEXP type codes of R's serialize.c
_NILVALUE, _REFSXP, _GLOBALENV, _EMPTYENV, _BASEENV, _MISSINGARG, _UNBOUND = 254, 255, 253, 242, 241, 251, 252
_SYMSXP, _LISTSXP, _CHARSXP, _LGLSXP, _INTSXP, _REALSXP, _STRSXP, _VECSXP = 1, 2, 9, 10, 13, 14, 16, 19
_ALTREP, _ATTRLISTSXP, _ATTRLANGSXP, _LANGSXP = 238, 239, 240, 6
_NA_INT = -2147483648


class RDataError(ValueError):
    pass


class _Reader:
    def __init__(self, buf: bytes):
        self.b = buf
        self.p = 0
        self.refs = []

    def take(self, n: int) -> bytes:
        if self.p + n > len(self.b):
            raise RDataError("truncated R data stream")
        out = self.b[self.p:self.p + n]
        self.p += n
        return out

    def i32(self) -> int:
        return struct.unpack(">i", self.take(4))[0]

    def length(self) -> int:
        n = self.i32()
        if n == -1:  # long vector: two more ints
            hi, lo = struct.unpack(">II", self.take(8))
            n = (hi << 32) | lo
        return n

    def item(self):
        flags = self.i32()
        t = flags & 0xFF
        has_attr = bool(flags & 0x200)
        has_tag = bool(flags & 0x400)
        if t in (_NILVALUE, _GLOBALENV, _EMPTYENV, _BASEENV, _MISSINGARG, _UNBOUND):
            return None
        if t == _REFSXP:
            idx = flags >> 8
            if idx == 0:
                idx = self.i32()
            return self.refs[idx - 1]
        if t == _SYMSXP:
            name = self.item()
            self.refs.append(name)
            return name
        if t == _CHARSXP:
            n = self.i32()
            return None if n == -1 else self.take(n).decode("utf-8", "replace")
        if t in (_LISTSXP, _LANGSXP, _ATTRLISTSXP, _ATTRLANGSXP):
            # pairlist: [attr] [tag] car cdr ; returned as a list of (tag, value)
            out = []
            while True:
                if t in (_ATTRLISTSXP, _ATTRLANGSXP) or has_attr:
                    self.item()
                tag = self.item() if has_tag else None
                out.append((tag, self.item()))
                flags = self.i32()
                t = flags & 0xFF
                has_attr = bool(flags & 0x200)
                has_tag = bool(flags & 0x400)
                if t == _NILVALUE:
                    return out
                if t not in (_LISTSXP, _LANGSXP, _ATTRLISTSXP, _ATTRLANGSXP):
                    raise RDataError(f"unsupported pairlist tail type {t}")
        if t in (_LGLSXP, _INTSXP):
            n = self.length()
            v = np.frombuffer(self.take(4 * n), dtype=">i4").astype(np.int32)
            return self._with_attr(v, has_attr, logical=(t == _LGLSXP))
        if t == _REALSXP:
            n = self.length()
            v = np.frombuffer(self.take(8 * n), dtype=">f8").astype(np.float64)
            return self._with_attr(v, has_attr)
        if t == _STRSXP:
            n = self.length()
            v = [self.item() for _ in range(n)]
            return self._with_attr(v, has_attr)
        if t == _VECSXP:
            n = self.length()
            v = [self.item() for _ in range(n)]
            return self._with_attr(v, has_attr)
        if t == _ALTREP:
            info = self.item()       # pairlist: class symbol, package symbol, type
            state = self.item()
            self.item()              # attributes
            cls = info[0][1] if info else None
            if cls in ("compact_intseq", "compact_realseq"):
                n, start, step = (float(x) for x in np.asarray(state)[:3])
                return start + step * np.arange(int(n))
            if cls in ("wrap_real", "wrap_integer", "wrap_logical", "wrap_string"):
                return state[0] if isinstance(state, list) else state
            raise RDataError(f"unsupported ALTREP class {cls!r}")
        raise RDataError(f"unsupported R object type {t}")

    def _with_attr(self, value, has_attr, logical=False):
        attrs = {}
        if has_attr:
            for tag, val in (self.item() or []):
                attrs[tag] = val
        return {"value": value, "attr": attrs, "logical": logical} if attrs or logical else value


def _decompress(raw: bytes) -> bytes:
    if raw[:2] == b"\x1f\x8b":
        return gzip.decompress(raw)
    if raw[:3] == b"BZh":
        return bz2.decompress(raw)
    if raw[:6] == b"\xfd7zXZ\x00":
        return lzma.decompress(raw)
    return raw


def _to_numpy(obj):
    """R object -> ndarray (matrix with names) where it is numeric; NA_integer_ / NA_logical_ -> NaN."""
    if isinstance(obj, dict):
        v, attrs = obj["value"], obj["attr"]
        if isinstance(v, np.ndarray):
            if v.dtype == np.int32:
                v = np.where(v == _NA_INT, np.nan, v.astype(np.float64))
            dim = attrs.get("dim")
            if dim is not None:
                d = np.asarray(dim["value"] if isinstance(dim, dict) else dim, dtype=np.int64)
                v = np.asfortranarray(v.reshape(tuple(int(x) for x in d), order="F"))
            names = attrs.get("dimnames") or attrs.get("names")
            return {"data": v, "dimnames": names}
        return {"data": v, "attr": attrs}
    if isinstance(obj, np.ndarray) and obj.dtype == np.int32:
        return {"data": np.where(obj == _NA_INT, np.nan, obj.astype(np.float64)), "dimnames": None}
    return {"data": obj, "dimnames": None}


def read_r_data(path: str) -> dict:
    """Objects of an .rda/.RData file (name -> {"data": ndarray, "dimnames": ...}) or the single object of
    an .rds file (under the key None)."""
    buf = _decompress(open(path, "rb").read())
    is_rda = buf[:4] in (b"RDX2", b"RDX3")
    if is_rda:
        buf = buf[5:]
    if buf[:2] != b"X\n":
        raise RDataError("only the XDR serialisation format is supported")
    r = _Reader(buf)
    r.take(2)
    version = r.i32()
    r.i32()
    r.i32()
    if version == 3:
        r.take(r.i32())  # native encoding name
    elif version != 2:
        raise RDataError(f"unsupported serialisation version {version}")
    top = r.item()
    if is_rda:
        return {tag: _to_numpy(val) for tag, val in top}
    return {None: _to_numpy(top)}


def read_r_matrix(path: str, name: str | None = None):
    """(matrix, rownames, colnames) of the (named or only) numeric matrix in an R data file."""
    objs = read_r_data(path)
    if name is None:
        cands = [k for k, v in objs.items() if isinstance(v.get("data"), np.ndarray) and v["data"].ndim == 2]
        if len(cands) != 1:
            raise RDataError(f"expected exactly one matrix, found {cands}")
        name = cands[0]
    obj = objs[name]
    dn = obj.get("dimnames")
    rn = cn = None
    if isinstance(dn, list) and len(dn) == 2:
        rn, cn = dn
        rn = rn["value"] if isinstance(rn, dict) else rn
        cn = cn["value"] if isinstance(cn, dict) else cn
    return obj["data"], rn, cn


# --------------------------------------------------------------------------------------------------
# R/reshaping.R
# --------------------------------------------------------------------------------------------------
def cor_matrix_2_long_df(in_matrix):
    """Square matrix (DataFrame with row / column names) -> long data.frame(s1, s2, cor), stacked column by
    column as utils::stack does (R/reshaping.R:15-32)."""
    if pd is None:
        raise RuntimeError("pandas is required")
    df = in_matrix if isinstance(in_matrix, pd.DataFrame) else pd.DataFrame(in_matrix)
    vals = df.to_numpy()
    rows = np.asarray([str(r) for r in df.index], dtype=object)
    cols = np.asarray([str(c) for c in df.columns], dtype=object)
    return pd.DataFrame({"s1": np.tile(rows, len(cols)), "s2": np.repeat(cols, len(rows)),
                         "cor": vals.reshape(-1, order="F")})


def long_df_2_cor_matrix(long_df, is_square=True):
    """long data.frame(s1, s2, cor) -> (possibly square) matrix; half-filled square input is mirrored
    (R/reshaping.R:44-68).  Levels are sorted, as factor() sorts them."""
    if pd is None:
        raise RuntimeError("pandas is required")
    if not all(c in long_df.columns for c in ("s1", "s2", "cor")):
        raise ValueError("The data.frame must contain the names 's1', 's2', and 'cor'.")
    s1 = long_df["s1"].astype(str).to_numpy()
    s2 = long_df["s2"].astype(str).to_numpy()
    if is_square:
        l1 = l2 = sorted(set(s1) | set(s2))
    else:
        l1, l2 = sorted(set(s1)), sorted(set(s2))
    i1 = {k: i for i, k in enumerate(l1)}
    i2 = {k: i for i, k in enumerate(l2)}
    m = np.full((len(l1), len(l2)), np.nan)
    r = np.fromiter((i1[a] for a in s1), dtype=np.int64, count=len(s1))
    c = np.fromiter((i2[b] for b in s2), dtype=np.int64, count=len(s2))
    vals = long_df["cor"].to_numpy(dtype=np.float64)
    m[r, c] = vals
    if len(long_df) != m.shape[0] * m.shape[1] and is_square:
        m[c, r] = vals
    return pd.DataFrame(m, index=l1, columns=l2)
