/*
 * icikt_oracle.c -- CPU restatement of the reference's ICI-Kendall-tau pair kernel.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under icikendalltau_amd/ may import, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and there only as the checker / reported baseline.
 *
 * What it restates (all file:line under /root/reference):
 *   ici_kt()              src/kendallc.cpp:166-366   -> icikt_oracle_pair()
 *   sortedIndex()         src/kendallc.cpp:5-12      -> stable_argsort()
 *   compare_self()+cumsum src/kendallc.cpp:14-31,251 -> dense_rank_sorted()
 *   compare_both()/which_notzero()/diff  :33-67,261-264 -> joint run lengths
 *   kendall_discordant()  src/kendallc.cpp:69-100    -> fenwick_discordant()
 *   count_rank_tie()      src/kendallc.cpp:102-118   -> count_rank_tie()
 *   ici_kt_pairs()        src/kendallc.cpp:369-549   -> icikt_oracle_bruteforce()
 *                         (only the O(n^2) counting idea, used as a cross-check)
 *   ici_split()           R/kendalltau.R:280-308     -> icikt_oracle_pairs()
 *   setup_missing_matrix  R/utils.R:1-23             -> (tests build the mask in numpy)
 *
 * Third-party arithmetic that is NOT in /root/reference and is restated from its
 * published algorithm:
 *   - R's pnorm (libR nmath, unpinned R version; reference DESCRIPTION has no R pin):
 *     W. J. Cody, "Rational Chebyshev approximations for the error function",
 *     Math. Comp. 23 (1969) 631-637, in the pnorm_both() arrangement -> pnorm_cody().
 *   - Rcpp sugar (LinkingTo: Rcpp, unpinned) integer semantics: element-wise int32
 *     products and an int32 running sum in count_rank_tie.  The
 *     NA_INTEGER-propagation rule can never fire for n < 70000 (no t makes a
 *     product equal INT_MIN; checked by enumeration in tests/test_oracle.py).
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against the
 * reference's own known answers (tests/testthat/test-kendall-tau.R:5-59), its
 * snapshot (tests/testthat/_snaps/kendall-tau.md:1-17), README.md:140-153,255-258
 * and the vignette numbers, via an exact emulation of R's RNG (oracle/rrng.py).
 * Unpinned by any reference fixture (code reading only): Q1 (t0/2), Q2 (int32 wrap),
 * perspective="local" on real data beyond completeness.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define ICIKT_PERSPECTIVE_LOCAL 0
#define ICIKT_PERSPECTIVE_GLOBAL 1
#define ICIKT_ALT_TWO_SIDED 0
#define ICIKT_ALT_LESS 1
#define ICIKT_ALT_GREATER 2
#define ICIKT_ALT_OTHER 3

/* per-pair reason codes (match include/icikt.h) */
#define ICIKT_OK 0
#define ICIKT_NA_ALL_MISSING 1   /* kendallc.cpp:190-199, silent */
#define ICIKT_NA_SHORT 2         /* kendallc.cpp:224-231, warning */
#define ICIKT_NA_SINGLE_UNIQUE 3 /* kendallc.cpp:234-244, warning */
#define ICIKT_NA_TIES_EQ_TOTAL 4 /* kendallc.cpp:291-298, warning */

/* layout of the int64 counts record (match include/icikt.h) */
enum {
  CNT_N = 0, CNT_MISSING, CNT_DIS, CNT_NTIE, CNT_XTIE, CNT_YTIE,
  CNT_X0, CNT_X1, CNT_Y0, CNT_Y1, CNT_TOT, CNT_SUMOBS, CNT_NFIELDS
};

static double na_real(void) {
  /* R's NA_REAL: a quiet NaN with payload 1954 (0x7A2). */
  union { uint64_t u; double d; } v;
  v.u = 0x7FF00000000007A2ULL;
  return v.d;
}

/* ---- R pnorm (Cody 1969), lower or upper tail, no log ---------------------- */
static void pnorm_both(double x, double *cum, double *ccum) {
  static const double a[5] = {2.2352520354606839287, 161.02823106855587881,
                              1067.6894854603709582, 18154.981253343561249,
                              0.065682337918207449113};
  static const double b[4] = {47.20258190468824187, 976.09855173777669322,
                              10260.932208618978205, 45507.789335026729956};
  static const double c[9] = {0.39894151208813466764, 8.8831497943883759412,
                              93.506656132177855979,  597.27027639480026226,
                              2494.5375852903726711,  6848.1904505362823326,
                              11602.651437647350124,  9842.7148383839780218,
                              1.0765576773720192317e-8};
  static const double d[8] = {22.266688044328115691, 235.38790178262499861,
                              1519.377599407554805,  6485.558298266760755,
                              18615.571640885098091, 34900.952721145977266,
                              38912.003286093271411, 19685.429676859990727};
  static const double p[6] = {0.21589853405795699,     0.1274011611602473639,
                              0.022235277870649807,    0.001421619193227893466,
                              2.9112874951168792e-5,   0.02307344176494017303};
  static const double q[5] = {1.28426009614491121,    0.468238212480865118,
                              0.0659881378689285515,  0.00378239633202758244,
                              7.29751555083966205e-5};
  const double one_over_sqrt_2pi = 0.398942280401432677939946059934;
  const double sqrt32 = 5.656854249492380195206754896838;
  double xden, xnum, temp, del, xsq, y;
  int i;

  if (isnan(x)) { *cum = *ccum = x; return; }
  y = fabs(x);
  if (y <= 0.67448975) {
    if (y > DBL_EPSILON * 0.5) {
      xsq = x * x;
      xnum = a[4] * xsq;
      xden = xsq;
      for (i = 0; i < 3; ++i) {
        xnum = (xnum + a[i]) * xsq;
        xden = (xden + b[i]) * xsq;
      }
    } else {
      xnum = xden = 0.0;
    }
    temp = x * (xnum + a[3]) / (xden + b[3]);
    *cum = 0.5 + temp;
    *ccum = 0.5 - temp;
  } else if (y <= sqrt32) {
    xnum = c[8] * y;
    xden = y;
    for (i = 0; i < 7; ++i) {
      xnum = (xnum + c[i]) * y;
      xden = (xden + d[i]) * y;
    }
    temp = (xnum + c[7]) / (xden + d[7]);
    xsq = trunc(y * 16) / 16;
    del = (y - xsq) * (y + xsq);
    *cum = exp(-xsq * xsq * 0.5) * exp(-del * 0.5) * temp;
    *ccum = 1.0 - *cum;
    if (x > 0.) { temp = *cum; *cum = *ccum; *ccum = temp; }
  } else if ((-37.5193 < x && x < 8.2924) || (-8.2924 < x && x < 37.5193)) {
    xsq = 1.0 / (x * x);
    xnum = p[5] * xsq;
    xden = xsq;
    for (i = 0; i < 4; ++i) {
      xnum = (xnum + p[i]) * xsq;
      xden = (xden + q[i]) * xsq;
    }
    temp = xsq * (xnum + p[4]) / (xden + q[4]);
    temp = (one_over_sqrt_2pi - temp) / y;
    xsq = trunc(x * 16) / 16;
    del = (x - xsq) * (x + xsq);
    *cum = exp(-xsq * xsq * 0.5) * exp(-del * 0.5) * temp;
    *ccum = 1.0 - *cum;
    if (x > 0.) { temp = *cum; *cum = *ccum; *ccum = temp; }
  } else {
    if (x > 0) { *cum = 1.; *ccum = 0.; }
    else       { *cum = 0.; *ccum = 1.; }
  }
}

double icikt_oracle_pnorm(double z, int lower_tail) {
  double cum, ccum;
  if (isinf(z)) return (z > 0) == (lower_tail != 0) ? 1.0 : 0.0;
  pnorm_both(z, &cum, &ccum);
  return lower_tail ? cum : ccum;
}

/* ---- sortedIndex (kendallc.cpp:5-12): stable argsort, comparator x[i] < x[j] - */
static void merge_sort_idx(const double *x, int32_t *idx, int32_t *tmp, int64_t n) {
  /* bottom-up stable merge sort; a stable sort's result is unique, so this is the
     same permutation std::stable_sort returns. */
  for (int64_t w = 1; w < n; w *= 2) {
    for (int64_t lo = 0; lo < n; lo += 2 * w) {
      int64_t mid = lo + w < n ? lo + w : n;
      int64_t hi = lo + 2 * w < n ? lo + 2 * w : n;
      int64_t i = lo, j = mid, k = lo;
      while (i < mid && j < hi) {
        if (x[idx[j]] < x[idx[i]]) tmp[k++] = idx[j++];
        else tmp[k++] = idx[i++];
      }
      while (i < mid) tmp[k++] = idx[i++];
      while (j < hi) tmp[k++] = idx[j++];
    }
    memcpy(idx, tmp, (size_t)n * sizeof(int32_t));
  }
}

/* ---- kendall_discordant (kendallc.cpp:69-100): Fenwick tree, int accumulator - */
static int32_t fenwick_discordant(const int32_t *x, const int32_t *y, int64_t n,
                                  int32_t *arr, int64_t sup) {
  /* arr has sup entries, zeroed by the caller.  `dis` is a C int in the
     reference (:78); it cannot overflow for n <= 65535. */
  int64_t i = 0, k = 0;
  int32_t dis = 0;
  while (i < n) {
    while (k < n && x[i] == x[k]) {
      dis = (int32_t)((double)dis + (double)i);
      int32_t idx = y[k];
      while (idx != 0) {
        dis -= arr[idx];
        idx = idx & (idx - 1);
      }
      k++;
    }
    while (i < k) {
      int64_t idx = y[i];
      while (idx < sup) {
        arr[idx] += 1;
        idx += idx & (-idx);
      }
      i++;
    }
  }
  return dis;
}

/* ---- count_rank_tie (kendallc.cpp:102-118) ------------------------------------
 * ranks: dense ranks 1..K.  int32_compat != 0 reproduces the reference's int32
 * element products and int32 running sum (two's-complement wrap, as gcc/x86-64
 * generates for the Rcpp sugar expression); 0 computes the sums in int64.       */
static int32_t wrap32(int64_t v) { return (int32_t)(uint32_t)(uint64_t)v; }

static void count_rank_tie(const int32_t *ranks, int64_t n, int32_t kmax,
                           int32_t *hist, int int32_compat, double out[3]) {
  memset(hist, 0, (size_t)(kmax + 1) * sizeof(int32_t));
  for (int64_t i = 0; i < n; ++i) hist[ranks[i]]++;
  if (int32_compat) {
    int32_t s0 = 0, s1 = 0, s2 = 0;
    for (int32_t r = 1; r <= kmax; ++r) {
      int32_t t = hist[r];
      if (t < 2) continue; /* table(ranks[duplicated]) only holds groups >= 2 */
      int32_t tt1 = wrap32((int64_t)t * (t - 1));
      int32_t e0 = tt1;
      int32_t e1 = wrap32((int64_t)tt1 * (t - 2));
      int32_t e2 = wrap32((int64_t)tt1 * wrap32(2 * (int64_t)t + 5));
      s0 = wrap32((int64_t)s0 + e0);
      s1 = wrap32((int64_t)s1 + e1);
      s2 = wrap32((int64_t)s2 + e2);
    }
    out[0] = (double)(s0 / 2); /* int / int, truncating */
    out[1] = (double)(s1 / 2);
    out[2] = (double)s2;
  } else {
    int64_t s0 = 0, s1 = 0, s2 = 0;
    for (int32_t r = 1; r <= kmax; ++r) {
      int64_t t = hist[r];
      if (t < 2) continue;
      s0 += t * (t - 1);
      s1 += t * (t - 1) * (t - 2);
      s2 += t * (t - 1) * (2 * t + 5);
    }
    out[0] = (double)(s0 / 2);
    out[1] = (double)(s1 / 2);
    out[2] = (double)s2;
  }
}

static double signC(double x) { return x > 0 ? 1.0 : (x == 0 ? 0.0 : -1.0); }

/*
 * One ici_kt() evaluation.  Returns 0, or -1 when allocation fails.
 * out4   = tau, pvalue, tau_max, completeness (NA_REAL x4 for degenerate input)
 * counts = CNT_NFIELDS int64 (may be NULL); only filled when reason == ICIKT_OK
 *          or ICIKT_NA_TIES_EQ_TOTAL.
 * reason = per-pair reason code.
 */
int icikt_oracle_pair(const double *xin, const double *yin, int64_t len,
                      int perspective, int alternative, int continuity,
                      int int32_compat, double *out4, int64_t *counts,
                      int32_t *reason) {
  const double NA = na_real();
  int rc = 0;
  out4[0] = out4[1] = out4[2] = out4[3] = NA;
  if (counts) memset(counts, 0, CNT_NFIELDS * sizeof(int64_t));
  *reason = ICIKT_OK;

  size_t cap = (size_t)(len > 0 ? len : 1);
  double *x = (double *)malloc(cap * sizeof(double));
  double *y = (double *)malloc(cap * sizeof(double));
  double *x2 = (double *)malloc(cap * sizeof(double));
  double *y2 = (double *)malloc(cap * sizeof(double));
  double *tx = (double *)malloc(cap * sizeof(double));
  int32_t *perm = (int32_t *)malloc(cap * sizeof(int32_t));
  int32_t *tmp = (int32_t *)malloc(cap * sizeof(int32_t));
  int32_t *x4 = (int32_t *)malloc(cap * sizeof(int32_t));
  int32_t *y4 = (int32_t *)malloc(cap * sizeof(int32_t));
  int32_t *y4b = (int32_t *)malloc(cap * sizeof(int32_t));
  int32_t *arr = (int32_t *)calloc(cap + 2, sizeof(int32_t));
  if (!x || !y || !x2 || !y2 || !tx || !perm || !tmp || !x4 || !y4 || !y4b || !arr) {
    rc = -1;
    goto done;
  }

  /* :180-185 local perspective drops rows where both are NA (is_na: NA or NaN) */
  int64_t n = 0;
  for (int64_t i = 0; i < len; ++i) {
    if (perspective == ICIKT_PERSPECTIVE_LOCAL && isnan(xin[i]) && isnan(yin[i])) continue;
    x[n] = xin[i];
    y[n] = yin[i];
    ++n;
  }

  /* :190-199 */
  int64_t n_na_x = 0, n_na_y = 0;
  for (int64_t i = 0; i < n; ++i) {
    n_na_x += isnan(x[i]) ? 1 : 0;
    n_na_y += isnan(y[i]) ? 1 : 0;
  }
  if (n_na_x == n || n_na_y == n) {
    *reason = ICIKT_NA_ALL_MISSING;
    goto done;
  }

  /* :204-212 completeness */
  int64_t missingness = 0;
  for (int64_t i = 0; i < n; ++i) missingness += (isnan(x[i]) || isnan(y[i])) ? 1 : 0;
  long double either_na_length = (long double)n;
  long double completeness = 1 - (missingness / either_na_length);

  /* :214-219 fill NA with min - 0.1 (computed in double) */
  double min_x = INFINITY, min_y = INFINITY;
  int have_x = 0, have_y = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (!isnan(x[i])) { if (!have_x || x[i] < min_x) min_x = x[i]; have_x = 1; }
    if (!isnan(y[i])) { if (!have_y || y[i] < min_y) min_y = y[i]; have_y = 1; }
  }
  min_x = min_x - 0.1;
  min_y = min_y - 0.1;
  for (int64_t i = 0; i < n; ++i) {
    x2[i] = isnan(x[i]) ? min_x : x[i];
    y2[i] = isnan(y[i]) ? min_y : y[i];
  }

  int64_t n_entry = n;
  /* :224-231 */
  if (n_entry < 2) {
    *reason = ICIKT_NA_SHORT;
    goto done;
  }
  /* :234-244 unique() */
  int all_same_x = 1, all_same_y = 1;
  for (int64_t i = 1; i < n; ++i) {
    if (x2[i] != x2[0]) all_same_x = 0;
    if (y2[i] != y2[0]) all_same_y = 0;
  }
  if (all_same_x || all_same_y) {
    *reason = ICIKT_NA_SINGLE_UNIQUE;
    goto done;
  }

  /* :247-251 sort by y, dense ranks y4 */
  for (int64_t i = 0; i < n; ++i) perm[i] = (int32_t)i;
  merge_sort_idx(y2, perm, tmp, n);
  for (int64_t i = 0; i < n; ++i) tx[i] = x2[perm[i]];
  memcpy(x2, tx, (size_t)n * sizeof(double));
  for (int64_t i = 0; i < n; ++i) tx[i] = y2[perm[i]];
  memcpy(y2, tx, (size_t)n * sizeof(double));
  {
    int32_t r = 0;
    for (int64_t i = 0; i < n; ++i) {
      if (i == 0 || y2[i] != y2[i - 1]) ++r;
      y4[i] = r;
    }
  }
  /* :254-258 sort by x (stable => lexicographic (x,y)), dense ranks x4 */
  for (int64_t i = 0; i < n; ++i) perm[i] = (int32_t)i;
  merge_sort_idx(x2, perm, tmp, n);
  for (int64_t i = 0; i < n; ++i) tx[i] = x2[perm[i]];
  memcpy(x2, tx, (size_t)n * sizeof(double));
  for (int64_t i = 0; i < n; ++i) y4b[i] = y4[perm[i]];
  memcpy(y4, y4b, (size_t)n * sizeof(int32_t));
  int32_t kx = 0, ky = 0;
  {
    int32_t r = 0;
    for (int64_t i = 0; i < n; ++i) {
      if (i == 0 || x2[i] != x2[i - 1]) ++r;
      x4[i] = r;
    }
    kx = r;
    for (int64_t i = 0; i < n; ++i) if (y4[i] > ky) ky = y4[i];
  }

  /* :261-267 joint runs -> cnt; ntie = sum((cnt*(cnt-1))/2) in int32 */
  int64_t sum_obs = 0;
  int32_t ntie_i32 = 0;
  int64_t ntie_i64 = 0;
  {
    int64_t run_start = 0;
    for (int64_t i = 1; i <= n; ++i) {
      int brk = (i == n) || (x4[i] != x4[i - 1]) || (y4[i] != y4[i - 1]);
      if (brk) {
        int64_t cnt = i - run_start;
        ++sum_obs;
        if (int32_compat) {
          int32_t e = wrap32(cnt * (cnt - 1)) / 2;
          ntie_i32 = wrap32((int64_t)ntie_i32 + e);
        }
        ntie_i64 += cnt * (cnt - 1) / 2;
        run_start = i;
      }
    }
    ++sum_obs; /* obs has a leading 1 and a pushed-back 1: sum = runs + 1 */
  }
  long double ntie = int32_compat ? (long double)ntie_i32 : (long double)ntie_i64;

  int64_t dis = fenwick_discordant(x4, y4, n, arr, (int64_t)ky + 1);

  /* :270-278 */
  double xc[3], yc[3];
  count_rank_tie(x4, n, kx, arr, int32_compat, xc);
  count_rank_tie(y4, n, ky, arr, int32_compat, yc);
  double xtie = xc[0], x0 = xc[1], x1 = xc[2];
  double ytie = yc[0], y0 = yc[1], y1 = yc[2];

  int64_t tot = (n_entry * (n_entry - 1)) / 2;

  if (counts) {
    counts[CNT_N] = n_entry;
    counts[CNT_MISSING] = missingness;
    counts[CNT_DIS] = dis;
    counts[CNT_NTIE] = (int64_t)ntie;
    counts[CNT_XTIE] = (int64_t)xtie;
    counts[CNT_YTIE] = (int64_t)ytie;
    counts[CNT_X0] = (int64_t)x0;
    counts[CNT_X1] = (int64_t)x1;
    counts[CNT_Y0] = (int64_t)y0;
    counts[CNT_Y1] = (int64_t)y1;
    counts[CNT_TOT] = tot;
    counts[CNT_SUMOBS] = sum_obs;
  }

  /* :291-298 */
  if (xtie == (double)tot || ytie == (double)tot) {
    *reason = ICIKT_NA_TIES_EQ_TOTAL;
    goto done;
  }

  /* :300-308 */
  long double con_minus_dis = tot - xtie - ytie + ntie - 2 * dis;
  long double tau = con_minus_dis / sqrt((tot - xtie) * (tot - ytie));
  long double con_plus_dis = tot - xtie - ytie + ntie;
  long double tau_max = con_plus_dis / sqrt((tot - xtie) * (tot - ytie));
  if (tau > 1) tau = 1;
  else if (tau < -1) tau = -1;

  /* :310-321 */
  int64_t m = n_entry * (n_entry - 1);
  long double var = ((m * (2 * n_entry + 5) - x1 - y1) / 18 +
                     (2 * xtie * ytie) / m + x0 * y0 / (9 * m * (n_entry - 2)));
  long double s_adjusted = tau * sqrt(((m / 2) - xtie) * ((m / 2) - ytie));
  if (continuity) {
    long double adj_s2 = signC((double)s_adjusted) * (fabsl(s_adjusted) - 1);
    s_adjusted = adj_s2;
  }
  double z_b = (double)(s_adjusted / sqrtl(var));

  /* :323-332 */
  double pval = 0.0;
  if (alternative == ICIKT_ALT_LESS) {
    pval = icikt_oracle_pnorm(z_b, 1);
  } else if (alternative == ICIKT_ALT_GREATER) {
    pval = icikt_oracle_pnorm(z_b, 0);
  } else if (alternative == ICIKT_ALT_TWO_SIDED) {
    double p0 = icikt_oracle_pnorm(z_b, 1);
    double p1 = icikt_oracle_pnorm(z_b, 0);
    /* Rcpp sugar min(): NaN in p0 propagates; p0 < p1 otherwise */
    double mn = p0;
    if (!isnan(p0)) { if (isnan(p1)) mn = p1; else if (p1 < mn) mn = p1; }
    pval = 2 * mn;
  }
  out4[0] = (double)tau;
  out4[1] = pval;
  out4[2] = (double)tau_max;
  out4[3] = (double)completeness;

done:
  free(x); free(y); free(x2); free(y2); free(tx); free(perm); free(tmp);
  free(x4); free(y4); free(y4b); free(arr);
  return rc;
}

/*
 * ici_split() (R/kendalltau.R:280-308): loop ici_kt over a pair list.
 * X: column-major n_feat x n_samp (ld = leading dimension), NaN/NA = missing.
 * pi/pj: 0-based column indices.  out4: P x 4 row-major.  counts: P x CNT_NFIELDS or NULL.
 */
int icikt_oracle_pairs(const double *X, int64_t n_feat, int64_t n_samp, int64_t ld,
                       const int32_t *pi, const int32_t *pj, int64_t n_pairs,
                       int perspective, int alternative, int continuity,
                       int int32_compat, double *out4, int64_t *counts,
                       int32_t *reasons) {
  for (int64_t p = 0; p < n_pairs; ++p) {
    if (pi[p] < 0 || pi[p] >= n_samp || pj[p] < 0 || pj[p] >= n_samp) return -2;
    int rc = icikt_oracle_pair(X + (int64_t)pi[p] * ld, X + (int64_t)pj[p] * ld, n_feat,
                               perspective, alternative, continuity, int32_compat,
                               out4 + 4 * p, counts ? counts + CNT_NFIELDS * p : NULL,
                               reasons + p);
    if (rc) return rc;
  }
  return 0;
}

/*
 * O(n^2) cross-check in the spirit of ici_kt_pairs (kendallc.cpp:447-452): counts
 * concordant / discordant / tie classes by enumerating all row pairs of the
 * NA-filled vectors.  cnt5 = {con, dis, xtie_only.., } see below.  Exact int64.
 *   cnt[0]=concordant  cnt[1]=discordant  cnt[2]=pairs tied in x (incl. joint)
 *   cnt[3]=pairs tied in y (incl. joint)  cnt[4]=pairs tied in both
 */
int icikt_oracle_bruteforce(const double *xin, const double *yin, int64_t len,
                            int perspective, int64_t *cnt) {
  double *x = (double *)malloc((size_t)(len > 0 ? len : 1) * sizeof(double));
  double *y = (double *)malloc((size_t)(len > 0 ? len : 1) * sizeof(double));
  if (!x || !y) { free(x); free(y); return -1; }
  int64_t n = 0;
  for (int64_t i = 0; i < len; ++i) {
    if (perspective == ICIKT_PERSPECTIVE_LOCAL && isnan(xin[i]) && isnan(yin[i])) continue;
    x[n] = xin[i]; y[n] = yin[i]; ++n;
  }
  double min_x = INFINITY, min_y = INFINITY;
  for (int64_t i = 0; i < n; ++i) {
    if (!isnan(x[i]) && x[i] < min_x) min_x = x[i];
    if (!isnan(y[i]) && y[i] < min_y) min_y = y[i];
  }
  min_x -= 0.1; min_y -= 0.1;
  for (int64_t i = 0; i < n; ++i) {
    if (isnan(x[i])) x[i] = min_x;
    if (isnan(y[i])) y[i] = min_y;
  }
  memset(cnt, 0, 5 * sizeof(int64_t));
  for (int64_t i = 0; i < n; ++i) {
    for (int64_t j = i + 1; j < n; ++j) {
      int sx = (x[i] < x[j]) - (x[i] > x[j]);
      int sy = (y[i] < y[j]) - (y[i] > y[j]);
      if (sx == 0) cnt[2]++;
      if (sy == 0) cnt[3]++;
      if (sx == 0 && sy == 0) cnt[4]++;
      if (sx * sy > 0) cnt[0]++;
      if (sx * sy < 0) cnt[1]++;
    }
  }
  free(x); free(y);
  return 0;
}

int icikt_oracle_counts_fields(void) { return CNT_NFIELDS; }
