/* TEST DOUBLE: see Rinternals.h in this directory. */
#ifndef ICIKT_R_MOCK_R_H
#define ICIKT_R_MOCK_R_H
#include <stdlib.h>
#include <limits.h>
#define NA_LOGICAL INT_MIN
#define NA_INTEGER INT_MIN
#endif
