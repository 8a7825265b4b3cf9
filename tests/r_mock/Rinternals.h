/* TEST DOUBLE of the subset of R's C API that icikendalltau_amd/r/icikt_rglue.c uses -- NOT R, not part of the product.
 * R itself (Rinternals.h, libR) is absent from the build container and the GPU box; with these declarations and the
 * tiny object model of r_mock.c the glue can be compiled with -Wall -Werror, loaded, and driven through its three
 * .Call entry points from a test (tests/test_rglue_mock.py).  Semantics follow "Writing R Extensions" for exactly the
 * calls the glue makes: column-major matrices with a dim attribute, 1-based nothing (the glue does that itself),
 * NA_LOGICAL = INT_MIN, R_alloc memory that lives until the call returns, Rf_error that does not return. */
#ifndef ICIKT_R_MOCK_RINTERNALS_H
#define ICIKT_R_MOCK_RINTERNALS_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mock_sexp *SEXP;
typedef ptrdiff_t R_xlen_t;
typedef unsigned int SEXPTYPE;
typedef enum { FALSE = 0, TRUE } Rboolean;

#define NILSXP 0
#define LGLSXP 10
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19
#define CHARSXP 9

extern SEXP R_NilValue;
extern SEXP R_NamesSymbol;

void Rf_error(const char *fmt, ...) __attribute__((noreturn, format(printf, 1, 2)));
Rboolean Rf_isReal(SEXP s);
Rboolean Rf_isMatrix(SEXP s);
Rboolean Rf_isNull(SEXP s);
int Rf_nrows(SEXP s);
int Rf_ncols(SEXP s);
int Rf_asInteger(SEXP s);
int Rf_asLogical(SEXP s);
double Rf_asReal(SEXP s);
R_xlen_t XLENGTH(SEXP s);
int *INTEGER(SEXP s);
int *LOGICAL(SEXP s);
double *REAL(SEXP s);
const char *CHAR(SEXP s);
SEXP STRING_ELT(SEXP s, R_xlen_t i);
SEXP VECTOR_ELT(SEXP s, R_xlen_t i);
SEXP SET_VECTOR_ELT(SEXP s, R_xlen_t i, SEXP v);
SEXP Rf_allocVector(SEXPTYPE type, R_xlen_t n);
SEXP Rf_allocMatrix(SEXPTYPE type, int nrow, int ncol);
SEXP Rf_mkNamed(SEXPTYPE type, const char **names);
SEXP Rf_protect(SEXP s);
void Rf_unprotect(int n);
#define PROTECT(s) Rf_protect(s)
#define UNPROTECT(n) Rf_unprotect(n)
char *R_alloc(size_t n, int size);

#ifdef __cplusplus
}
#endif
#endif
