/* TEST DOUBLE: see ../Rinternals.h. */
#ifndef ICIKT_R_MOCK_RDYNLOAD_H
#define ICIKT_R_MOCK_RDYNLOAD_H
#include "../Rinternals.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef void *(*DL_FUNC)(void);
typedef struct { const char *name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef struct { const char *name; DL_FUNC fun; int numArgs; } R_CMethodDef;
typedef R_CMethodDef R_FortranMethodDef;
typedef R_CallMethodDef R_ExternalMethodDef;
typedef struct mock_dllinfo DllInfo;
int R_registerRoutines(DllInfo *info, const R_CMethodDef *c, const R_CallMethodDef *call, const R_FortranMethodDef *f,
                       const R_ExternalMethodDef *e);
Rboolean R_useDynamicSymbols(DllInfo *info, Rboolean value);
#ifdef __cplusplus
}
#endif
#endif
