"""Checker engine for CPU-only tests of the host logic (sharding, gather, reshape).

It plugs the CPU oracle into icikendalltau_amd.api's `engine=` hook.  Test infrastructure only: the
product's default engine is the MI355X one and there is no CPU fallback in the package.
"""
import numpy as np

from oracle import oracle as O


class OracleEngine:
    name = "oracle"

    def pairs(self, X, pi, pj, perspective, alternative, continuity):
        out, _cnt, rsn = O.ici_pairs(X, pi, pj, perspective, alternative, continuity, want_counts=False)
        return out, rsn

    def pairs_counts(self, X, pi, pj, perspective, alternative, continuity):
        out, cnt, rsn = O.ici_pairs(X, pi, pj, perspective, alternative, continuity, want_counts=True)
        names = ("n", "missing", "dis", "ntie", "xtie", "ytie", "x0", "x1", "y0", "y1", "tot")
        return out, rsn, {k: cnt[:, i] for i, k in enumerate(names)}

    def missingness(self, X, pi, pj):
        m = np.isnan(np.asarray(X))
        return np.array([(m[:, i] | m[:, j]).sum() for i, j in zip(pi, pj)], dtype=np.int64)
