/*
 * icikt_rglue.c -- R .Call glue over the C ABI of include/icikt.h.
 *
 * NOT compiled in the build container (Rinternals.h is absent there); it is what a maintainer of the
 * reference adds to src/ (or ships as a small companion package) to route ici_split()
 * (R/kendalltau.R:280-308) and ici_kt() (R/RcppExports.R:62-64) through the MI355X library:
 *
 *   R CMD SHLIB icikt_rglue.c -I<repo>/include -L<repo>/icikendalltau_amd -licikt_hip
 *
 * Registered routines (R_CallMethodDef, as src/RcppExports.cpp:113-128 does for the Rcpp path):
 *   .Call("icikt_R_pairs", exclude_data, pi, pj, perspective, alternative, continuity, device, n_gpu, want_counts)
 *       exclude_data  REALSXP matrix n_feat x n_samp (column-major, NA = missing; integer input is
 *                     coerced by the R wrapper, as Rcpp does for NumericVector)
 *       pi, pj        INTSXP, 1-based column indices (R convention); NULL = all combn pairs
 *       device        first HIP device; n_gpu devices device .. device + n_gpu - 1 are used
 *       n_gpu         1: icikt_pairs_f64 on one device.  > 1: icikt_pairs_multi_f64 -- ONE call, one host thread
 *                     per GPU inside the library, RCCL all-gather / gather over xGMI; replaces the furrr fan-out of
 *                     R/kendalltau.R:158 (HIP must not be driven from forked multicore workers)
 *       want_counts   TRUE: also the P x 11 integer counts record (the numbers src/kendallc.cpp:342-363 prints)
 *       returns list(raw, pvalue, taumax, completeness, reason[, counts]) of length-P vectors
 *   .Call("icikt_R_matrix", data_matrix, global_na, pi, pj, perspective, alternative, continuity, scale_max, diag_good,
 *         device, n_gpu)
 *       ici_kendalltau(return_matrix = TRUE) below its argument checks in ONE call (icikt_matrix_f64 /
 *       icikt_matrix_multi_f64): the exclusion rule of setup_missing_matrix (R/utils.R:1-23) inside the pre-pass,
 *       the pair kernels, scale_and_reshape (R/kendalltau.R:357-421) on the device, one copy of five S x S matrices
 *       back -- no 523 776-row data.frame, no name-indexed fill.  data_matrix is the RAW features x samples matrix
 *       (NOT exclude_data); global_na the values to exclude (NA, Inf, 0, ...); pi / pj NULL = all combn pairs.
 *       returns list(cor, raw, pvalue, taumax, completeness, keep, reason_counts): five S x S numeric matrices, the
 *       logical S x n_feat `keep` matrix of R/kendalltau.R:417 and the pairs per reason code 0..4
 *   .Call("icikt_R_missingness", exclude_data, pi, pj, device) -> numeric(P)
 * Errors become R errors (Rf_error), as BEGIN_RCPP/END_RCPP does (src/RcppExports.cpp:84,95);
 * per-pair degenerate cases are returned as NA_real_ x4 plus a reason code so that the R wrapper can
 * raise the reference's warning texts (src/kendallc.cpp:225,238,292).
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <string.h>

#include "icikt.h"

static icikt_ctx *g_ctx = NULL;
static int g_dev = -1;
static icikt_multi *g_multi = NULL; /* communicators are expensive: kept until (device, n_gpu) changes */
static int g_multi_dev = -1, g_multi_n = 0;

static icikt_ctx *get_ctx(int device) {
  if (g_ctx && g_dev == device) return g_ctx;
  if (g_ctx) { icikt_ctx_destroy(g_ctx); g_ctx = NULL; }
  int rc = icikt_ctx_create(device, &g_ctx);
  if (rc == ICIKT_E_NO_DEVICE) Rf_error("icikt: no usable HIP device (there is no CPU fallback)");
  if (rc != ICIKT_SUCCESS) Rf_error("icikt: icikt_ctx_create(%d) failed with code %d", device, rc);
  g_dev = device;
  return g_ctx;
}

static icikt_multi *get_multi(int device, int n_gpu) {
  if (g_multi && g_multi_dev == device && g_multi_n == n_gpu) return g_multi;
  if (g_multi) { icikt_multi_destroy(g_multi); g_multi = NULL; }
  int *devs = (int *)R_alloc(n_gpu, sizeof(int));
  for (int k = 0; k < n_gpu; ++k) devs[k] = device + k;
  int rc = icikt_multi_create(devs, n_gpu, ICIKT_MULTI_EXCHANGE_AUTO, &g_multi);
  if (rc == ICIKT_E_NO_DEVICE) Rf_error("icikt: no usable HIP device (there is no CPU fallback)");
  if (rc != ICIKT_SUCCESS) Rf_error("icikt: icikt_multi_create(%d devices from %d) failed with code %d", n_gpu, device, rc);
  g_multi_dev = device;
  g_multi_n = n_gpu;
  return g_multi;
}

static int perspective_code(SEXP s) {
  /* anything other than "local" is the global behaviour (src/kendallc.cpp:180) */
  return strcmp(CHAR(STRING_ELT(s, 0)), "local") == 0 ? ICIKT_PERSPECTIVE_LOCAL : ICIKT_PERSPECTIVE_GLOBAL;
}

static int alternative_code(SEXP s) {
  const char *a = CHAR(STRING_ELT(s, 0));
  if (strcmp(a, "two.sided") == 0) return ICIKT_ALT_TWO_SIDED;
  if (strcmp(a, "less") == 0) return ICIKT_ALT_LESS;
  if (strcmp(a, "greater") == 0) return ICIKT_ALT_GREATER;
  return ICIKT_ALT_OTHER; /* p-value stays 0, as in the reference (:323-332) */
}

SEXP icikt_R_pairs(SEXP x, SEXP pi, SEXP pj, SEXP perspective, SEXP alternative, SEXP continuity, SEXP device,
                   SEXP n_gpu, SEXP want_counts) {
  if (!Rf_isReal(x) || !Rf_isMatrix(x)) Rf_error("icikt: exclude_data must be a double matrix");
  const int64_t n_feat = Rf_nrows(x), n_samp = Rf_ncols(x);
  const int ngpu = Rf_asInteger(n_gpu) > 1 ? Rf_asInteger(n_gpu) : 1;
  const int with_counts = Rf_asLogical(want_counts) == TRUE;
  icikt_ctx *ctx = (ngpu == 1) ? get_ctx(Rf_asInteger(device)) : NULL;
  icikt_multi *multi = (ngpu > 1) ? get_multi(Rf_asInteger(device), ngpu) : NULL;
  int64_t P;
  int32_t *pi0 = NULL, *pj0 = NULL;
  if (Rf_isNull(pi)) {
    P = n_samp * (n_samp - 1) / 2;
  } else {
    P = XLENGTH(pi);
    if (XLENGTH(pj) != P) Rf_error("icikt: pi and pj differ in length");
    pi0 = (int32_t *)R_alloc(P > 0 ? P : 1, sizeof(int32_t));
    pj0 = (int32_t *)R_alloc(P > 0 ? P : 1, sizeof(int32_t));
    for (int64_t p = 0; p < P; ++p) { pi0[p] = INTEGER(pi)[p] - 1; pj0[p] = INTEGER(pj)[p] - 1; }
  }
  double *out4 = (double *)R_alloc(P > 0 ? 4 * P : 1, sizeof(double));
  int32_t *reasons = (int32_t *)R_alloc(P > 0 ? P : 1, sizeof(int32_t));
  int64_t *counts = with_counts ? (int64_t *)R_alloc(P > 0 ? P * ICIKT_CNT_FIELDS : 1, sizeof(int64_t)) : NULL;
  int rc;
  if (multi) {
    rc = icikt_pairs_multi_f64(multi, REAL(x), n_feat, n_samp, n_feat, pi0, pj0, P, perspective_code(perspective),
                               alternative_code(alternative), Rf_asLogical(continuity) ? 1 : 0, 0u, out4, counts, reasons);
    if (rc != ICIKT_SUCCESS) Rf_error("icikt: %s (code %d)", icikt_multi_last_error(multi), rc);
  } else {
    rc = icikt_pairs_f64(ctx, REAL(x), n_feat, n_samp, n_feat, pi0, pj0, P, perspective_code(perspective),
                         alternative_code(alternative), Rf_asLogical(continuity) ? 1 : 0, 0u, out4, counts, reasons);
    if (rc != ICIKT_SUCCESS) Rf_error("icikt: %s (code %d)", icikt_last_error(ctx), rc);
  }
  const char *nm[] = {"raw", "pvalue", "taumax", "completeness", "reason", "counts", ""};
  SEXP res = PROTECT(Rf_mkNamed(VECSXP, nm));
  for (int f = 0; f < 4; ++f) {
    SEXP v = PROTECT(Rf_allocVector(REALSXP, P));
    for (int64_t p = 0; p < P; ++p) REAL(v)[p] = out4[4 * p + f];
    SET_VECTOR_ELT(res, f, v);
    UNPROTECT(1);
  }
  SEXP r = PROTECT(Rf_allocVector(INTSXP, P));
  for (int64_t p = 0; p < P; ++p) INTEGER(r)[p] = reasons[p];
  SET_VECTOR_ELT(res, 4, r);
  if (with_counts) { /* P x 11, columns in the order of ICIKT_CNT_*; doubles hold every count exactly (< 2^53) */
    SEXP cm = PROTECT(Rf_allocMatrix(REALSXP, (int)P, ICIKT_CNT_FIELDS));
    for (int64_t p = 0; p < P; ++p)
      for (int f = 0; f < ICIKT_CNT_FIELDS; ++f) REAL(cm)[(int64_t)f * P + p] = (double)counts[p * ICIKT_CNT_FIELDS + f];
    SET_VECTOR_ELT(res, 5, cm);
    UNPROTECT(1);
  }
  UNPROTECT(2);
  return res;
}

SEXP icikt_R_matrix(SEXP x, SEXP global_na, SEXP pi, SEXP pj, SEXP perspective, SEXP alternative, SEXP continuity,
                    SEXP scale_max, SEXP diag_good, SEXP device, SEXP n_gpu) {
  if (!Rf_isReal(x) || !Rf_isMatrix(x)) Rf_error("icikt: data_matrix must be a double matrix");
  const int64_t n_feat = Rf_nrows(x), n_samp = Rf_ncols(x);
  const int ngpu = Rf_asInteger(n_gpu) > 1 ? Rf_asInteger(n_gpu) : 1;
  icikt_ctx *ctx = (ngpu == 1) ? get_ctx(Rf_asInteger(device)) : NULL;
  icikt_multi *multi = (ngpu > 1) ? get_multi(Rf_asInteger(device), ngpu) : NULL;
  /* global_na: NA_real_ / NaN select "missing", +-Inf "infinite", anything else is compared with == (R/utils.R:1-23) */
  const int n_na = Rf_isNull(global_na) ? 0 : (int)XLENGTH(global_na);
  double *gna = (double *)R_alloc(n_na > 0 ? n_na : 1, sizeof(double));
  for (int k = 0; k < n_na; ++k) gna[k] = Rf_isReal(global_na) ? REAL(global_na)[k] : Rf_asReal(global_na);
  int64_t P = 0;
  int32_t *pi0 = NULL, *pj0 = NULL;
  if (!Rf_isNull(pi)) {
    P = XLENGTH(pi);
    if (XLENGTH(pj) != P) Rf_error("icikt: pi and pj differ in length");
    pi0 = (int32_t *)R_alloc(P > 0 ? P : 1, sizeof(int32_t));
    pj0 = (int32_t *)R_alloc(P > 0 ? P : 1, sizeof(int32_t));
    for (int64_t p = 0; p < P; ++p) { pi0[p] = INTEGER(pi)[p] - 1; pj0[p] = INTEGER(pj)[p] - 1; }
  }
  const char *nm[] = {"cor", "raw", "pvalue", "taumax", "completeness", "keep", "reason_counts", ""};
  SEXP res = PROTECT(Rf_mkNamed(VECSXP, nm));
  /* the library writes the five matrices contiguously: one REALSXP of 5 S^2, split afterwards without a second pass
     over the host (the matrices are symmetric: R's column-major order is either order) */
  double *out5 = (double *)R_alloc((size_t)(5 * n_samp * n_samp > 0 ? 5 * n_samp * n_samp : 1), sizeof(double));
  uint8_t *keep = (uint8_t *)R_alloc((size_t)(n_samp * n_feat > 0 ? n_samp * n_feat : 1), 1);
  int64_t rc5[5] = {0, 0, 0, 0, 0};
  int rc;
  if (multi) {
    rc = icikt_matrix_multi_f64(multi, REAL(x), n_feat, n_samp, n_feat, gna, n_na, pi0, pj0, P, perspective_code(perspective),
                                alternative_code(alternative), Rf_asLogical(continuity) ? 1 : 0, 0u,
                                Rf_asLogical(scale_max) ? 1 : 0, Rf_asLogical(diag_good) ? 1 : 0, out5, keep, rc5);
    if (rc != ICIKT_SUCCESS) Rf_error("icikt: %s (code %d)", icikt_multi_last_error(multi), rc);
  } else {
    rc = icikt_matrix_f64(ctx, REAL(x), n_feat, n_samp, n_feat, gna, n_na, pi0, pj0, P, perspective_code(perspective),
                          alternative_code(alternative), Rf_asLogical(continuity) ? 1 : 0, 0u,
                          Rf_asLogical(scale_max) ? 1 : 0, Rf_asLogical(diag_good) ? 1 : 0, out5, keep, rc5);
    if (rc != ICIKT_SUCCESS) Rf_error("icikt: %s (code %d)", icikt_last_error(ctx), rc);
  }
  for (int f = 0; f < 5; ++f) {
    SEXP m = PROTECT(Rf_allocMatrix(REALSXP, (int)n_samp, (int)n_samp));
    memcpy(REAL(m), out5 + (size_t)f * n_samp * n_samp, (size_t)n_samp * n_samp * sizeof(double));
    SET_VECTOR_ELT(res, f, m);
    UNPROTECT(1);
  }
  SEXP k = PROTECT(Rf_allocMatrix(LGLSXP, (int)n_samp, (int)n_feat));   /* keep = t(!exclude_loc): samples x features */
  for (int64_t c = 0; c < n_samp; ++c)
    for (int64_t r = 0; r < n_feat; ++r) LOGICAL(k)[r * n_samp + c] = keep[c * n_feat + r] ? TRUE : FALSE;
  SET_VECTOR_ELT(res, 5, k);
  SEXP rcv = PROTECT(Rf_allocVector(REALSXP, 5));
  for (int f = 0; f < 5; ++f) REAL(rcv)[f] = (double)rc5[f];
  SET_VECTOR_ELT(res, 6, rcv);
  UNPROTECT(3);
  return res;
}

SEXP icikt_R_missingness(SEXP x, SEXP pi, SEXP pj, SEXP device) {
  if (!Rf_isReal(x) || !Rf_isMatrix(x)) Rf_error("icikt: exclude_data must be a double matrix");
  const int64_t n_feat = Rf_nrows(x), n_samp = Rf_ncols(x), P = XLENGTH(pi);
  icikt_ctx *ctx = get_ctx(Rf_asInteger(device));
  int32_t *pi0 = (int32_t *)R_alloc(P > 0 ? P : 1, sizeof(int32_t));
  int32_t *pj0 = (int32_t *)R_alloc(P > 0 ? P : 1, sizeof(int32_t));
  int64_t *m = (int64_t *)R_alloc(P > 0 ? P : 1, sizeof(int64_t));
  for (int64_t p = 0; p < P; ++p) { pi0[p] = INTEGER(pi)[p] - 1; pj0[p] = INTEGER(pj)[p] - 1; }
  int rc = icikt_missingness_f64(ctx, REAL(x), n_feat, n_samp, n_feat, pi0, pj0, P, m);
  if (rc != ICIKT_SUCCESS) Rf_error("icikt: %s (code %d)", icikt_last_error(ctx), rc);
  SEXP v = PROTECT(Rf_allocVector(REALSXP, P));
  for (int64_t p = 0; p < P; ++p) REAL(v)[p] = (double)m[p];
  UNPROTECT(1);
  return v;
}

static const R_CallMethodDef CallEntries[] = {
    {"icikt_R_pairs", (DL_FUNC)&icikt_R_pairs, 9},
    {"icikt_R_matrix", (DL_FUNC)&icikt_R_matrix, 11},
    {"icikt_R_missingness", (DL_FUNC)&icikt_R_missingness, 4},
    {NULL, NULL, 0}};

void R_init_icikt_rglue(DllInfo *dll) {
  R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
  R_useDynamicSymbols(dll, FALSE);
}

void R_unload_icikt_rglue(DllInfo *dll) {
  (void)dll;
  if (g_ctx) { icikt_ctx_destroy(g_ctx); g_ctx = NULL; }
  if (g_multi) { icikt_multi_destroy(g_multi); g_multi = NULL; }
}
