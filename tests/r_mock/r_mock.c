/* TEST DOUBLE of the subset of R's C API that icikendalltau_amd/r/icikt_rglue.c uses (declarations: Rinternals.h in
 * this directory) plus a small driver for tests/test_rglue_mock.py: build argument objects, look a routine up in the
 * table the glue registered, check its arity as .Call does, run it, read the result.  NOT R and not part of the
 * product: R is absent from the build container and from the GPU box, and this is how the glue gets compiled with
 * warnings as errors and exercised end to end anyway.
 *
 * Object model: one struct per SEXP, everything allocated from a list that mock_reset() frees (objects and R_alloc
 * memory alike -- R frees R_alloc memory when .Call returns, the test frees it after reading the result).
 * Rf_error formats its message and longjmps back into mock_dotcall, which then returns NULL. */
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "R.h"
#include "Rinternals.h"
#include "R_ext/Rdynload.h"

struct mock_sexp {
  SEXPTYPE type;
  R_xlen_t len;
  void *data;          /* int / double / SEXP / char payload */
  int nrow, ncol, is_matrix;
  const char **names;  /* VECSXP made by Rf_mkNamed */
};
struct mock_dllinfo { int dynamic_symbols; };

static struct mock_sexp nil_obj = {NILSXP, 0, NULL, 0, 0, 0, NULL};
SEXP R_NilValue = &nil_obj;
SEXP R_NamesSymbol = &nil_obj;

/* ---- allocations ------------------------------------------------------------------------------------------- */
struct chunk { struct chunk *next; };
static struct chunk *g_chunks = NULL;
static void *arena(size_t bytes) {
  struct chunk *c = (struct chunk *)calloc(1, sizeof(struct chunk) + (bytes ? bytes : 1) + 16);
  if (!c) abort();
  c->next = g_chunks;
  g_chunks = c;
  return (void *)(((uintptr_t)(c + 1) + 15u) & ~(uintptr_t)15u);
}
void mock_reset(void) {
  while (g_chunks) { struct chunk *n = g_chunks->next; free(g_chunks); g_chunks = n; }
}
char *R_alloc(size_t n, int size) { return (char *)arena(n * (size_t)size); }

/* ---- errors and the protect stack -------------------------------------------------------------------------- */
static jmp_buf g_jmp;
static int g_armed = 0, g_protect = 0, g_protect_max = 0;
static char g_err[1024];
void Rf_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  if (!g_armed) { fprintf(stderr, "Rf_error outside mock_dotcall: %s\n", g_err); abort(); }
  longjmp(g_jmp, 1);
}
const char *mock_last_error(void) { return g_err; }
SEXP Rf_protect(SEXP s) { if (++g_protect > g_protect_max) g_protect_max = g_protect; return s; }
void Rf_unprotect(int n) { g_protect -= n; }
int mock_protect_depth(void) { return g_protect; }
int mock_protect_max(void) { return g_protect_max; }

/* ---- objects ----------------------------------------------------------------------------------------------- */
static size_t elt_size(SEXPTYPE t) {
  switch (t) {
    case LGLSXP: case INTSXP: return sizeof(int);
    case REALSXP: return sizeof(double);
    case STRSXP: case VECSXP: return sizeof(SEXP);
    case CHARSXP: return 1;
    default: Rf_error("mock: unsupported SEXPTYPE %u", t);
  }
}
SEXP Rf_allocVector(SEXPTYPE type, R_xlen_t n) {
  if (n < 0) Rf_error("negative length vectors are not allowed");
  SEXP s = (SEXP)arena(sizeof(struct mock_sexp));
  s->type = type;
  s->len = n;
  s->data = arena((size_t)n * elt_size(type) + (type == CHARSXP ? 1 : 0));
  if (type == STRSXP || type == VECSXP)
    for (R_xlen_t i = 0; i < n; ++i) ((SEXP *)s->data)[i] = R_NilValue;
  return s;
}
SEXP Rf_allocMatrix(SEXPTYPE type, int nrow, int ncol) {
  if (nrow < 0 || ncol < 0) Rf_error("negative extents to matrix");
  SEXP s = Rf_allocVector(type, (R_xlen_t)nrow * ncol);
  s->nrow = nrow; s->ncol = ncol; s->is_matrix = 1;
  return s;
}
SEXP Rf_mkNamed(SEXPTYPE type, const char **names) {
  R_xlen_t n = 0;
  while (names[n][0] != '\0') ++n;
  SEXP s = Rf_allocVector(type, n);
  s->names = (const char **)arena((size_t)n * sizeof(char *));
  for (R_xlen_t i = 0; i < n; ++i) {
    char *c = (char *)arena(strlen(names[i]) + 1);
    strcpy(c, names[i]);
    s->names[i] = c;
  }
  return s;
}
Rboolean Rf_isReal(SEXP s) { return s->type == REALSXP ? TRUE : FALSE; }
Rboolean Rf_isMatrix(SEXP s) { return s->is_matrix ? TRUE : FALSE; }
Rboolean Rf_isNull(SEXP s) { return s->type == NILSXP ? TRUE : FALSE; }
int Rf_nrows(SEXP s) { if (!s->is_matrix) Rf_error("object is not a matrix"); return s->nrow; }
int Rf_ncols(SEXP s) { if (!s->is_matrix) Rf_error("object is not a matrix"); return s->ncol; }
R_xlen_t XLENGTH(SEXP s) { return s->len; }
static void want(SEXP s, SEXPTYPE a, SEXPTYPE b, const char *who) {
  if (s->type != a && s->type != b) Rf_error("%s() applied to an object of type %u", who, s->type);
}
int *INTEGER(SEXP s) { want(s, INTSXP, LGLSXP, "INTEGER"); return (int *)s->data; }
int *LOGICAL(SEXP s) { want(s, LGLSXP, LGLSXP, "LOGICAL"); return (int *)s->data; }
double *REAL(SEXP s) { want(s, REALSXP, REALSXP, "REAL"); return (double *)s->data; }
const char *CHAR(SEXP s) { want(s, CHARSXP, CHARSXP, "CHAR"); return (const char *)s->data; }
SEXP STRING_ELT(SEXP s, R_xlen_t i) {
  want(s, STRSXP, STRSXP, "STRING_ELT");
  if (i < 0 || i >= s->len) Rf_error("STRING_ELT: index %ld out of range", (long)i);
  return ((SEXP *)s->data)[i];
}
SEXP VECTOR_ELT(SEXP s, R_xlen_t i) {
  want(s, VECSXP, VECSXP, "VECTOR_ELT");
  if (i < 0 || i >= s->len) Rf_error("VECTOR_ELT: index %ld out of range", (long)i);
  return ((SEXP *)s->data)[i];
}
SEXP SET_VECTOR_ELT(SEXP s, R_xlen_t i, SEXP v) {
  want(s, VECSXP, VECSXP, "SET_VECTOR_ELT");
  if (i < 0 || i >= s->len) Rf_error("SET_VECTOR_ELT: index %ld out of range", (long)i);
  ((SEXP *)s->data)[i] = v;
  return v;
}
int Rf_asInteger(SEXP s) {
  if (s->len < 1) return NA_INTEGER;
  if (s->type == INTSXP || s->type == LGLSXP) return ((int *)s->data)[0];
  if (s->type == REALSXP) { double v = ((double *)s->data)[0]; return v != v ? NA_INTEGER : (int)v; }
  return NA_INTEGER;
}
int Rf_asLogical(SEXP s) {
  if (s->len < 1) return NA_LOGICAL;
  if (s->type == LGLSXP) return ((int *)s->data)[0];
  if (s->type == INTSXP) { int v = ((int *)s->data)[0]; return v == NA_INTEGER ? NA_LOGICAL : v != 0; }
  if (s->type == REALSXP) { double v = ((double *)s->data)[0]; return v != v ? NA_LOGICAL : v != 0.0; }
  return NA_LOGICAL;
}
double Rf_asReal(SEXP s) {
  if (s->len < 1) return 0.0 / 0.0;
  if (s->type == REALSXP) return ((double *)s->data)[0];
  if (s->type == INTSXP || s->type == LGLSXP) { int v = ((int *)s->data)[0]; return v == NA_INTEGER ? 0.0 / 0.0 : (double)v; }
  return 0.0 / 0.0;
}

/* ---- registration ------------------------------------------------------------------------------------------ */
static const R_CallMethodDef *g_calls = NULL;
static struct mock_dllinfo g_dll = {1};
int R_registerRoutines(DllInfo *info, const R_CMethodDef *c, const R_CallMethodDef *call, const R_FortranMethodDef *f,
                       const R_ExternalMethodDef *e) {
  (void)info; (void)c; (void)f; (void)e;
  g_calls = call;
  return 1;
}
Rboolean R_useDynamicSymbols(DllInfo *info, Rboolean value) {
  Rboolean old = info->dynamic_symbols ? TRUE : FALSE;
  info->dynamic_symbols = value;
  return old;
}

/* ---- the driver (called through ctypes) -------------------------------------------------------------------- */
void R_init_icikt_rglue(DllInfo *dll);
void R_unload_icikt_rglue(DllInfo *dll);
void mock_init(void) { R_init_icikt_rglue(&g_dll); }
void mock_unload(void) { R_unload_icikt_rglue(&g_dll); }
int mock_dynamic_symbols(void) { return g_dll.dynamic_symbols; }
int mock_n_routines(void) { int n = 0; while (g_calls && g_calls[n].name) ++n; return n; }
const char *mock_routine_name(int i) { return g_calls[i].name; }
int mock_routine_nargs(int i) { return g_calls[i].numArgs; }

SEXP mock_null(void) { return R_NilValue; }
SEXP mock_real_vector(const double *v, long n) {
  SEXP s = Rf_allocVector(REALSXP, n);
  if (n) memcpy(s->data, v, (size_t)n * sizeof(double));
  return s;
}
SEXP mock_real_matrix(const double *v, int nrow, int ncol) {   /* column-major, as R holds it */
  SEXP s = Rf_allocMatrix(REALSXP, nrow, ncol);
  if (s->len) memcpy(s->data, v, (size_t)s->len * sizeof(double));
  return s;
}
SEXP mock_int_vector(const int *v, long n) {
  SEXP s = Rf_allocVector(INTSXP, n);
  if (n) memcpy(s->data, v, (size_t)n * sizeof(int));
  return s;
}
SEXP mock_logical(int v) { SEXP s = Rf_allocVector(LGLSXP, 1); ((int *)s->data)[0] = v; return s; }
SEXP mock_string(const char *c) {
  SEXP ch = Rf_allocVector(CHARSXP, (R_xlen_t)strlen(c));
  strcpy((char *)ch->data, c);
  SEXP s = Rf_allocVector(STRSXP, 1);
  ((SEXP *)s->data)[0] = ch;
  return s;
}
int mock_type(SEXP s) { return (int)s->type; }
long mock_length(SEXP s) { return (long)s->len; }
int mock_is_matrix(SEXP s) { return s->is_matrix; }
int mock_nrow(SEXP s) { return s->nrow; }
int mock_ncol(SEXP s) { return s->ncol; }
void *mock_data(SEXP s) { return s->data; }
SEXP mock_list_elt(SEXP s, long i) { return ((SEXP *)s->data)[i]; }
const char *mock_list_name(SEXP s, long i) { return s->names ? s->names[i] : ""; }

typedef SEXP (*fn4)(SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn9)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn11)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);

/* .Call(name, args...): NULL after an R error (message: mock_last_error) */
SEXP mock_dotcall(const char *name, int nargs, SEXP *a) {
  g_err[0] = '\0';
  g_protect = g_protect_max = 0;
  const R_CallMethodDef *volatile d = NULL;
  for (int i = 0; g_calls && g_calls[i].name; ++i)
    if (strcmp(g_calls[i].name, name) == 0) d = &g_calls[i];
  if (!d) { snprintf(g_err, sizeof g_err, "C symbol name \"%s\" not in load table", name); return NULL; }
  if (d->numArgs != nargs) {
    snprintf(g_err, sizeof g_err, "Incorrect number of arguments (%d), expecting %d for '%s'", nargs, d->numArgs, name);
    return NULL;
  }
  SEXP volatile res = NULL;
  g_armed = 1;
  if (setjmp(g_jmp) == 0) {
    switch (nargs) {
      case 4: res = ((fn4)d->fun)(a[0], a[1], a[2], a[3]); break;
      case 9: res = ((fn9)d->fun)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8]); break;
      case 11: res = ((fn11)d->fun)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10]); break;
      default: snprintf(g_err, sizeof g_err, "mock: no dispatcher for %d arguments", nargs);
    }
  } else {
    res = NULL;
  }
  g_armed = 0;
  return res;
}
