"""Numpy model of K0/K1 (development aid: same algorithm as icikt_kernels.hip, lane loops written out)."""
import numpy as np

def k0(col):
    n = len(col)
    na = np.isnan(col)
    fill = (np.min(col[~na]) if (~na).any() else np.inf) - 0.1
    v = np.where(na, fill, col) + 0.0
    asc = np.argsort(v, kind="stable")
    sv = v[asc]
    start = np.r_[True, sv[1:] != sv[:-1]]
    end = np.r_[sv[1:] != sv[:-1], True]
    lo = np.maximum.accumulate(np.where(start, np.arange(n), 0))
    hi = np.minimum.accumulate(np.where(end, np.arange(n), n - 1)[::-1])[::-1]
    q_row = np.empty(n, int); lo_row = np.empty(n, int); hi_row = np.empty(n, int)
    q_row[asc] = np.arange(n); lo_row[asc] = lo; hi_row[asc] = hi
    order = asc[::-1]
    gflag = end[::-1]
    return dict(order=order, gflag=gflag, q=q_row, lo=lo_row, hi=hi_row, na=na)

def k1(A, B):
    n = len(A["order"])
    seen = np.zeros(n + 64, bool); pend = np.zeros(n + 64, bool)
    dis = 0; ntie = 0
    nb = (n + 63) // 64
    for t in range(nb):
        ks = np.arange(t * 64, min(n, t * 64 + 64))
        F = np.zeros(64, bool); F[:len(ks)] = A["gflag"][ks]
        Fn = A["gflag"][t * 64 + 64] if t * 64 + 64 < n else True
        rows = A["order"][ks]
        q = B["q"][rows]; lo = B["lo"][rows]; hi = B["hi"][rows]
        dis += sum(int(seen[:l].sum()) for l in lo)
        xg = np.cumsum(F[:len(ks)])
        for l in range(len(ks)):
            for j in range(l):
                if xg[j] != xg[l]:
                    dis += int(q[j] < lo[l])
                else:
                    ntie += int(lo[j] == lo[l])
        first = int(np.argmax(F)) if F.any() else 64
        ol = np.arange(len(ks)) < first
        if not F[0]:
            for l in np.nonzero(ol)[0]:
                ntie += int(pend[lo[l]:hi[l] + 1].sum())
            # rows of later groups of this step also see the open group's earlier rows (still in pend)
            for l in np.nonzero(~ol)[0]:
                dis += int(pend[:lo[l]].sum())
        if not F.any():
            pend[q] = True
            if Fn:
                seen |= pend; pend[:] = False
        else:
            last = 63 - int(np.argmax(F[::-1]))
            tl = np.arange(len(ks)) >= last
            pend[q[ol]] = True
            sel = (~ol) & (Fn | ~tl)
            seen[q[sel]] = True
            if not F[0]:
                seen |= pend; pend[:] = False
            if not Fn:
                pend[q[tl]] = True
    return dis, ntie

if __name__ == "__main__":
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle as O
    for n in (5, 63, 64, 65, 129, 300, 1000):
        rng = np.random.default_rng(n)
        X = rng.standard_normal((n, 6))
        X[rng.random(X.shape) < 0.15] = np.nan
        if n >= 300:
            X[:, 0] = np.round(X[:, 0]); X[:, 1] = np.round(X[:, 1] * 2)
        cols = [k0(X[:, c]) for c in range(6)]
        for i in range(6):
            for j in range(i + 1, 6):
                d, t = k1(cols[j], cols[i])
                _, cnt, rsn = O.ici_kt(X[:, i], X[:, j], "global")
                bf = O.bruteforce(X[:, i], X[:, j])
                assert (d, t) == (bf["dis"], bf["ntie"]), (n, i, j, d, t, bf)
                if rsn == 0:
                    assert (d, t) == (cnt["dis"], cnt["ntie"]), (n, i, j, d, t, cnt)
    print("model ok")
