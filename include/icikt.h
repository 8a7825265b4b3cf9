/*
 * icikt.h -- C ABI of the MI355X (gfx950) ICI-Kendall-tau pair engine.
 *
 * This is the drop-in boundary for the reference's hot path
 *   ici_kendalltau() -> ici_split() -> ici_kt()   (R/kendalltau.R:96-308, src/kendallc.cpp:166-366).
 * The reference crosses its FFI once per pair:
 *   .Call('_ICIKendallTau_ici_kt', x, y, perspective, alternative, continuity, output)
 *     R/RcppExports.R:62-64, C symbol SEXP _ICIKendallTau_ici_kt(SEXP x6) src/RcppExports.cpp:83-96
 * This library moves the boundary up to the batch (ici_split, R/kendalltau.R:280-308): one call per
 * pair LIST.  Plain pointers and sizes only; no R, Rcpp or torch types.  The R-side .Call glue a
 * maintainer adds is in icikendalltau_amd/r/ and INTEGRATION.md.
 *
 * Conventions
 *   - X is column-major n_feat x n_samp with leading dimension ld (R matrix layout; columns are
 *     samples, R/kendalltau.R:6).  Missing = NaN of any payload (R's NA_real_ is a NaN;
 *     Rcpp is_na() is true for NA and NaN, src/kendallc.cpp:181).
 *   - Pair indices are 0-based column indices; pair p is ici_kt(x = X[, pi[p]], y = X[, pj[p]]).
 *   - out4 is P x 4 row-major: tau, pvalue, tau_max, completeness (src/kendallc.cpp:171-172).
 *     Degenerate pairs get R's NA_real_ bit pattern (0x7FF00000000007A2) in all four and a reason.
 *   - All functions return ICIKT_SUCCESS (0) or a negative ICIKT_E_* code; the message is kept in
 *     the context (icikt_last_error).  Nothing throws across the boundary.
 *   - n_feat <= ICIKT_MAX_FEATURES (65 535) rows per column run the tuned kernels (16-bit positions).  Longer
 *     columns, up to ICIKT_MAX_FEATURES_WIDE (262 144) rows, are accepted by every single-device entry and run a
 *     plain 32-bit path (about two orders of magnitude slower per pair) in EXACT integer arithmetic: the reference's
 *     int32 wrap-around (its `int dis`, src/kendallc.cpp:78, and the Rcpp-sugar tie sums) is not reproduced there,
 *     i.e. ICIKT_FLAG_EXACT_INT64 is implied.  Beyond that: ICIKT_E_TOO_LONG with a message that names the limit.
 *   - A context owns one HIP device, one stream and its workspaces.  Calls on one context must come
 *     from one thread at a time.  Must not be used in a fork()ed child of a process that has
 *     already created a context (R/utils.R:68-80 furrr multicore workers).
 */
#ifndef ICIKT_H
#define ICIKT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ICIKT_VERSION 400 /* 0.4.0: ICIKT_FLAG_HOST_PINNED (no page-locking of caller memory by the library), mask-only missingness pre-pass */

/* status codes */
#define ICIKT_SUCCESS 0
#define ICIKT_E_INVALID (-1)    /* bad argument (message says which) */
#define ICIKT_E_HIP (-2)        /* HIP runtime error */
#define ICIKT_E_NOMEM (-3)      /* host allocation failure */
#define ICIKT_E_TOO_LONG (-4)   /* n_feat > ICIKT_MAX_FEATURES_WIDE (or > ICIKT_MAX_FEATURES where wide columns are not supported) */
#define ICIKT_E_STATE (-5)      /* call order: prepare / set_pairs before run */
#define ICIKT_E_NO_DEVICE (-6)  /* no usable HIP device: the product path never falls back to CPU */

/* n_feat limit of the uint16 rank path.  The reference's `int dis` (src/kendallc.cpp:78) and
 * int32 tie sums are themselves only safe to n ~ 65 535. */
#define ICIKT_MAX_FEATURES 65535
#define ICIKT_MAX_FEATURES_WIDE 262144

/* perspective (src/kendallc.cpp:180) */
#define ICIKT_PERSPECTIVE_LOCAL 0
#define ICIKT_PERSPECTIVE_GLOBAL 1
/* alternative (src/kendallc.cpp:323-332); OTHER = any unrecognised string: p-value stays 0 */
#define ICIKT_ALT_TWO_SIDED 0
#define ICIKT_ALT_LESS 1
#define ICIKT_ALT_GREATER 2
#define ICIKT_ALT_OTHER 3

/* flags */
#define ICIKT_FLAG_EXACT_INT64 1u /* tie sums in int64 instead of the reference's wrapping int32
                                     (count_rank_tie, src/kendallc.cpp:112-114) */
#define ICIKT_FLAG_TIMING 2u      /* record HIP events around each kernel (icikt_kernel_ms) */
#define ICIKT_FLAG_REUSE_COUNTS 4u /* icikt_run_dev: if the pair kernel already ran for this prepared matrix and pair
                                     list, keep its integer counts and run the epilogue only -- "local" is derived
                                     from the same counts as "global", so the second perspective (or another
                                     alternative / continuity) costs microseconds */

#define ICIKT_FLAG_HOST_PINNED 8u  /* host entries: the caller states that the matrix and the result arrays of this call
                                     lie in memory it has page-locked itself (hipHostMalloc / hipHostRegister); they are
                                     then copied from and into directly.  Without it (default) every transfer of 256 KB
                                     or more crosses the library's own pinned buffers: the library never page-locks
                                     and never probes caller memory.  Setting it for pageable memory is a caller error
                                     with the consequences of an asynchronous copy from pageable memory. */

#define ICIKT_FLAG_BALANCE_COST 16u /* icikt_pairs_multi_f64 / icikt_matrix_multi_f64: cut the pair list into blocks of equal COST
                                     instead of equal length.  The reference's chunks are ceiling(n_todo / ncore) pairs each
                                     (R/kendalltau.R:250-255), and one ici_kt costs the same whatever the data; here a pair's
                                     cost follows the tie structure of the column it streams (up to 2.5x between columns), and
                                     the pre-pass leaves that cost with every column.  Blocks stay consecutive in list order
                                     (no block longer than twice the equal share); results are the same, in the same order. */

/* per-pair reason codes; the host wrapper raises the reference's warnings from them */
#define ICIKT_OK 0
#define ICIKT_NA_ALL_MISSING 1   /* src/kendallc.cpp:190-199: silent NA x4 */
#define ICIKT_NA_SHORT 2         /* :224-231 "The vectors only have a single value, NA returned!" */
#define ICIKT_NA_SINGLE_UNIQUE 3 /* :234-244 "Either 'X' or 'Y' have only a single unique value, NA returned!" */
#define ICIKT_NA_TIES_EQ_TOTAL 4 /* :291-298 "Ties equal the total, NA returned!" */

/* int64 counts record per pair (the integers src/kendallc.cpp:342-363 prints when output != "simple") */
#define ICIKT_CNT_N 0        /* n_entry after the perspective's row filter */
#define ICIKT_CNT_MISSING 1  /* rows with x or y missing (:208-211) */
#define ICIKT_CNT_DIS 2      /* kendall_discordant (:69-100) */
#define ICIKT_CNT_NTIE 3     /* joint ties (:267) */
#define ICIKT_CNT_XTIE 4     /* count_rank_tie(x): ntie, t0, t1 (:102-118) */
#define ICIKT_CNT_YTIE 5
#define ICIKT_CNT_X0 6
#define ICIKT_CNT_X1 7
#define ICIKT_CNT_Y0 8
#define ICIKT_CNT_Y1 9
#define ICIKT_CNT_TOT 10     /* n(n-1)/2 (:280) */
#define ICIKT_CNT_FIELDS 11

/* kernel ids for icikt_kernel_ms */
#define ICIKT_K_PREPARE 0  /* per-column rank pre-pass */
#define ICIKT_K_PAIRS 1    /* pair kernel (discordance / joint-tie counting) */
#define ICIKT_K_EPILOGUE 2 /* tau / p-value epilogue */
#define ICIKT_K_COUNT 3

typedef struct icikt_ctx icikt_ctx;

int icikt_version(void);
/* Number of visible HIP devices (0 and ICIKT_E_NO_DEVICE when none). */
int icikt_device_count(int *count);

int icikt_ctx_create(int device, icikt_ctx **ctx);
void icikt_ctx_destroy(icikt_ctx *ctx);
const char *icikt_last_error(const icikt_ctx *ctx);
/* Use an existing hipStream_t (e.g. torch's current stream) instead of the context's own;
 * NULL selects HIP's default (null) stream.  icikt_ctx_use_own_stream() switches back. */
int icikt_ctx_set_stream(icikt_ctx *ctx, void *hip_stream);
int icikt_ctx_use_own_stream(icikt_ctx *ctx);
int icikt_sync(icikt_ctx *ctx);

/* ---- device-resident path (what bench.py and the multi-GPU driver call) -------------------- */

/* Per-column pre-pass over a DEVICE matrix: NA bitsets, fill value min-0.1 (src/kendallc.cpp:214-219),
 * stable argsort + dense tie groups (sortedIndex/compare_self, :5-31), tie sums (count_rank_tie).
 * Asynchronous on the context's stream.  dX must stay valid until the stream reaches this point. */
int icikt_prepare_dev(icikt_ctx *ctx, const double *dX, int64_t n_feat, int64_t n_samp, int64_t ld,
                      uint32_t flags);

/* Column-sharded pre-pass for multi-rank use: the prepared state is allocated for alloc_cols >= n_samp
 * columns (so that every rank's slice has the same size) but only columns [col_begin, col_end) are computed.
  * The caller then all-gathers the column slices of the two arrays of icikt_prep_arrays() that carry the information
 * (see below) across ranks (RCCL all-gather) and rebuilds the rest for the received columns before icikt_run_dev().
 * Array i is alloc_cols * bytes_per_col[i] bytes; a rank's columns are one contiguous slice of it provided col_begin is
 * even and col_end is even or n_samp (some arrays interleave column pairs; other ranges are refused), so give
 * every rank an even number of columns. */
#define ICIKT_PREP_ARRAYS 5
int icikt_prepare_cols_dev(icikt_ctx *ctx, const double *dX, int64_t n_feat, int64_t n_samp, int64_t ld,
                           int64_t col_begin, int64_t col_end, int64_t alloc_cols, uint32_t flags);
/* The same from a HOST matrix: columns [col_begin, col_end) are copied to the device in chunks that overlap
 * their pre-pass (a rank uploads 1 / world of the matrix).  Returns when X has been read. */
int icikt_prepare_cols_f64(icikt_ctx *ctx, const double *X, int64_t n_feat, int64_t n_samp, int64_t ld,
                           int64_t col_begin, int64_t col_end, int64_t alloc_cols, uint32_t flags);
int icikt_prep_arrays(icikt_ctx *ctx, void **ptrs, int64_t *bytes_per_col);
/* ICIKT_PREP_ORDER (the descending permutation) and ICIKT_PREP_META (per column: missing-row, fill-group and
 * group-start bitsets + the column's statistics) carry the information: 24 KB per column of 10 000 rows, two
 * collectives.  Everything else -- ICIKT_PREP_REC, ICIKT_PREP_HIROW, ICIKT_PREP_TGROUPS (exposed for inspection: 80 KB
 * per column; a rec / hirow block has n_pad + 8 rows, row n_pad the guard row of the step records) and the state that is
 * not exposed (per-row tie-group indices, the tie program and its step records, the 32-bit copy of the order that long columns keep) -- is a function of a column's order
 * and group starts: icikt_expand_cols_dev() rebuilds it for the received columns [col_begin, col_end). */
#define ICIKT_PREP_ORDER 0
#define ICIKT_PREP_REC 1
#define ICIKT_PREP_HIROW 2
#define ICIKT_PREP_META 3
#define ICIKT_PREP_TGROUPS 4
int icikt_expand_cols_dev(icikt_ctx *ctx, int64_t col_begin, int64_t col_end, uint32_t flags);

/* Pair list (HOST arrays, copied).  Mirrors setup_comparisons' output order (R/kendalltau.R:181-278). */
int icikt_set_pairs(icikt_ctx *ctx, const int32_t *pi, const int32_t *pj, int64_t n_pairs);
/* Pairs [begin, end) of utils::combn(n_samp, 2) order: (0,1),(0,2)...(0,S-1),(1,2)... (R/kendalltau.R:189).
 * This is how ranks shard the upper triangle (reference: ceiling(n_todo/ncore) chunks, :250-255). */
int icikt_set_pairs_combn(icikt_ctx *ctx, int64_t n_samp, int64_t begin, int64_t end);
/* Number of pairs currently set. */
int64_t icikt_num_pairs(const icikt_ctx *ctx);

/* Pair kernel + epilogue into DEVICE buffers: d_out4 [P*4] doubles, d_counts [P*ICIKT_CNT_FIELDS]
 * int64 or NULL, d_reasons [P] int32 or NULL.  Asynchronous on the context's stream. */
int icikt_run_dev(icikt_ctx *ctx, int perspective, int alternative, int continuity, uint32_t flags,
                  double *d_out4, int64_t *d_counts, int32_t *d_reasons);

/* Accumulated HIP-event time (ms) and launch count of one kernel since the last reset; needs
 * ICIKT_FLAG_TIMING on the calls being measured.  Synchronises the stream. */
int icikt_kernel_ms(icikt_ctx *ctx, int kernel, double *ms, int64_t *launches);
int icikt_reset_timers(icikt_ctx *ctx);

/* ---- host-buffer path (what the R .Call glue binds) ----------------------------------------- */

/* ici_split() replacement (R/kendalltau.R:280-308): H2D, pre-pass, pair kernel, epilogue, D2H.
 * pi == NULL means all C(n_samp, 2) pairs in combn order.  counts / reasons may be NULL. */
int icikt_pairs_f64(icikt_ctx *ctx, const double *X, int64_t n_feat, int64_t n_samp, int64_t ld,
                    const int32_t *pi, const int32_t *pj, int64_t n_pairs, int perspective,
                    int alternative, int continuity, uint32_t flags, double *out4, int64_t *counts,
                    int32_t *reasons);

/* ici_kt() replacement for one pair of host vectors of equal length n (the caller checks lengths,
 * src/kendallc.cpp:168-170). */
int icikt_pair_f64(icikt_ctx *ctx, const double *x, const double *y, int64_t n, int perspective,
                   int alternative, int continuity, uint32_t flags, double *out4, int64_t *counts,
                   int32_t *reason);

/* ---- the whole matrix behind one call (what the R glue binds for ici_kendalltau(return_matrix = TRUE)) ----------
 *
 * ici_kendalltau() below its argument checks (R/kendalltau.R:117-176), with NOTHING left to the host:
 *   setup_missing_matrix (R/utils.R:1-23)     global_na is applied by the pre-pass while it reads X: a cell is
 *                                             excluded if it is NaN and global_na holds a NaN, infinite and global_na
 *                                             holds an Inf, or == any other global_na value (at most 32 distinct ones:
 *                                             ICIKT_E_INVALID beyond; the front-ends mask such a matrix themselves).  The
 *                                             masked copy `exclude_data` (R/kendalltau.R:119-121) never exists.  A
 *                                             NaN in X is missing for ici_kt whatever global_na says (Rcpp is_na,
 *                                             src/kendallc.cpp:181) but counts as excluded only under the rule, as in R.
 *   ici_split over the pair list (:158)       pi == NULL: all C(n_samp, 2) pairs in combn order; else the caller's list
 *                                             (setup_comparisons' include_only filter, self pairs when !diag_good)
 *   scale_and_reshape (:357-421)              cor = raw / max(taumax, na.rm = TRUE) over the computed pairs when
 *                                             scale_max; diag_good: raw = cor = n_good / max(n_good), pvalue 0,
 *                                             taumax 1, completeness = n_good / n_feat on the diagonal, n_good =
 *                                             colSums(!exclude_loc); symmetric fill, cells never computed stay 0
 * out5: [5][n_samp][n_samp] doubles -- cor, raw, pvalue, taumax, completeness (symmetric: either major order).
 * keep (optional): [n_samp][n_feat] bytes, 1 = not excluded -- the reference's `keep = t(!exclude_loc)` (:417).
 * reason_counts (optional): [5] pairs per reason code ICIKT_OK .. ICIKT_NA_TIES_EQ_TOTAL; the host raises the
 * reference's warning once per pair of codes 2..4, as ici_split does.  Same error contract as icikt_pairs_f64. */
int icikt_matrix_f64(icikt_ctx *ctx, const double *X, int64_t n_feat, int64_t n_samp, int64_t ld,
                     const double *global_na, int n_global_na, const int32_t *pi, const int32_t *pj, int64_t n_pairs,
                     int perspective, int alternative, int continuity, uint32_t flags, int scale_max, int diag_good,
                     double *out5, uint8_t *keep, int64_t *reason_counts);

/* ---- several GPUs behind one call (what the R glue binds when n_gpu > 1) ----------------------
 *
 * Replaces the reference's worker fan-out, computation$split_fun(split_comparisons, ici_split, ...)
 * (R/kendalltau.R:158; chunks = ceiling(n_todo / ncore) consecutive pairs, :250-255; workers from
 * R/utils.R:68-80): one host thread per device inside ONE call from the R main thread.  Rank r uploads and
 * sorts its share of the columns only, the ranks all-gather the prepared columns (RCCL over xGMI), each runs
 * the pair kernel over block r of the pair list, and the results are gathered to the first device (RCCL) and
 * copied out once.  Same arguments, outputs and errors as icikt_pairs_f64. */
typedef struct icikt_multi icikt_multi;

#define ICIKT_MULTI_EXCHANGE_AUTO 0 /* RCCL when the listed devices are distinct, device copies otherwise */
#define ICIKT_MULTI_EXCHANGE_RCCL 1 /* ncclAllGather / ncclGather over xGMI */
#define ICIKT_MULTI_EXCHANGE_COPY 2 /* hipMemcpyPeerAsync between the ranks' buffers; a device may be listed
                                       more than once (how the flow is rehearsed on a one-GPU box) */
/* devices: n_gpu HIP device indices (NULL = 0 .. n_gpu-1).  Creates one context per entry and, for RCCL,
 * the communicators (ncclCommInitAll).  Keep the handle: creating communicators costs far more than a call. */
int icikt_multi_create(const int *devices, int n_gpu, int exchange, icikt_multi **out);
void icikt_multi_destroy(icikt_multi *m);
const char *icikt_multi_last_error(const icikt_multi *m);
int icikt_multi_n_gpu(const icikt_multi *m);
int icikt_multi_uses_rccl(const icikt_multi *m);
/* Ranks of the handle's RCCL communicator as RCCL reports them (ncclCommCount); 0 when the handle exchanges by device
 * copies and has no communicator, -1 on an RCCL error. */
int icikt_multi_comm_ranks(const icikt_multi *m);
int icikt_pairs_multi_f64(icikt_multi *m, const double *X, int64_t n_feat, int64_t n_samp, int64_t ld,
                          const int32_t *pi, const int32_t *pj, int64_t n_pairs, int perspective,
                          int alternative, int continuity, uint32_t flags, double *out4, int64_t *counts,
                          int32_t *reasons);
/* icikt_matrix_f64 on several GPUs: the ranks apply the exclusion rule to their own columns and return their rows
 * of `keep`; the first device assembles the five matrices from the gathered pair results and copies them out once. */
int icikt_matrix_multi_f64(icikt_multi *m, const double *X, int64_t n_feat, int64_t n_samp, int64_t ld,
                           const double *global_na, int n_global_na, const int32_t *pi, const int32_t *pj,
                           int64_t n_pairs, int perspective, int alternative, int continuity, uint32_t flags,
                           int scale_max, int diag_good, double *out5, uint8_t *keep, int64_t *reason_counts);
/* Wall-clock milliseconds of the last call's phases, the MAXIMUM over the ranks: H2D + pre-pass of the rank's
 * columns, exchange (all-gather + local rebuild), pair kernel + epilogue, gather + D2H.  With ICIKT_FLAG_TIMING each
 * phase ends with a stream synchronisation, so the figures are device time; without it they are host-side
 * enqueue times except the last, which absorbs everything still in flight.  The time a rank waits for the others
 * at the barriers between the phases is kept apart (icikt_multi_rank_phase_ms). */
#define ICIKT_MULTI_PHASE_PREPARE 0
#define ICIKT_MULTI_PHASE_EXCHANGE 1
#define ICIKT_MULTI_PHASE_PAIRS 2
#define ICIKT_MULTI_PHASE_GATHER 3
#define ICIKT_MULTI_PHASES 4
int icikt_multi_phase_ms(const icikt_multi *m, double *ms);
/* One rank's figures of the last call: ms[0 .. ICIKT_MULTI_PHASES-1] its phases, ms[ICIKT_MULTI_PHASES] the time it
 * waited for the other ranks at the barriers (ICIKT_MULTI_PHASES + 1 doubles): a spread between the ranks is
 * the imbalance of the partition. */
int icikt_multi_rank_phase_ms(const icikt_multi *m, int rank, double *ms);
/* Ranks the last icikt_pairs_multi_f64 call really used: n_gpu, or 1 when the job was too small to split (fewer
 * than two columns or 64 pairs per rank, no rows) or had wide columns (n_feat > ICIKT_MAX_FEATURES) and ran on the
 * first device alone; 0 after a call that failed its argument checks. */
int icikt_multi_ranks_used(const icikt_multi *m);
/* The cut of ICIKT_FLAG_BALANCE_COST as a function of its own (host arithmetic only, no device): col_cost[n_samp] = what
 * streaming each column costs; pj = the list's second indices (n_pairs of them), or n_pairs = -1 for all C(n_samp, 2)
 * pairs in combn order; n_blocks consecutive blocks of equal cost, none longer than max_block pairs (<= 0: no limit):
 * bounds[0 .. n_blocks], bounds[0] = 0, bounds[n_blocks] = the number of pairs. */
int icikt_cost_blocks(const uint32_t *col_cost, int64_t n_samp, const int32_t *pj, int64_t n_pairs, int n_blocks,
                      int64_t max_block, int64_t *bounds);
/* The pair blocks of the last call: rank r ran pairs [bounds[r], bounds[r + 1]) of the list (ranks_used + 1 values:
 * the reference's `core` chunks, or the cost-weighted cut of ICIKT_FLAG_BALANCE_COST). */
int icikt_multi_block_bounds(const icikt_multi *m, int64_t *bounds);
/* icikt_debug_set_plan() on every rank's context. */
int icikt_multi_debug_set_plan(icikt_multi *m, const char *spec);

/* kt_fast(use = "pairwise.complete.obs") (R/kendalltau.R:310-354, 448-545): for every pair the rows with a missing
 * value in EITHER vector are dropped, then ici_kt(..., perspective = "local") of what remains.  The masking, the
 * per-pair sorts and the counting all run on the device (pairs in chunks); same outputs as icikt_pairs_f64. */
int icikt_pairs_complete_f64(icikt_ctx *ctx, const double *X, int64_t n_feat, int64_t n_samp, int64_t ld,
                             const int32_t *pi, const int32_t *pj, int64_t n_pairs, int alternative,
                             int continuity, uint32_t flags, double *out4, int64_t *counts, int32_t *reasons);

/* pairwise_completeness() arithmetic (R/kendalltau.R:611-629): missingness[p] = #rows missing in
 * either column, from a host matrix whose missing cells are NaN.  Self pairs allowed. */
int icikt_missingness_f64(icikt_ctx *ctx, const double *X, int64_t n_feat, int64_t n_samp, int64_t ld,
                          const int32_t *pi, const int32_t *pj, int64_t n_pairs, int64_t *missingness);

/* Device self-test of the wavefront primitives the pair kernel relies on (DPP scan / shift). */
int icikt_selftest(icikt_ctx *ctx);

/* Development / test hook (the product path reads no environment variable): "key=value,key=value" overrides of
 * the pair kernel's launch plan and of the host path's H2D mode on this context; NULL or "" restores the library's
 * choices.  Keys: np (pairs per wave: 1 | 2), pend (l | g: open-group bitset in LDS | global memory), wpb (waves
 * per workgroup), half (0 | 1), tgmax (joint ties of long tie groups by the gathered column's tie groups -- list or count
 * mode -- up to this many of them; -1 = per-row mode), list (list mode up to this many tie groups, <= 128: count mode
 * takes over above), solo (0: SOLO steps run as MIXED steps), waves (half-wave kernels: waves per CU down to which the
 * pairs' counter tables may cost the launch occupancy), split (1 | 2 | 4: segments a half-wave task is cut in, whatever the
 * launch's size), merge (0 | 1: the pipelined host entries' pairs in a launch per chunk | in one launch behind the last chunk), gridmult / gridcap (persistent grid of the long-column kernel: a
 * multiple of the resident workgroups / at most this many), pipe (0 | 1: the host entries' chunk pipeline), k0 (0 | 1: the
 * pre-pass always in its 1 024-thread / 256-thread shape), verbose (0 | 1: print the chosen plan to stderr). */
int icikt_debug_set_plan(icikt_ctx *ctx, const char *spec);
/* Development hook: per step kind of the pair kernel (hot loop, hot step in the main loop, MIXED, GROUP, general,
 * closed-form tail, set-up) the steps taken, their rows and the wave cycles spent, as out24 = [steps x 8 | rows x 8 |
 * cycles x 8] since the last reset.  Counted only by a diagnostic build of the library (-DICIKT_STEP_STATS,
 * tools/step_stats.py); the product build executes no stamp and answers ICIKT_E_STATE. */
int icikt_debug_step_stats(icikt_ctx *ctx, uint64_t *out24, int reset);

#ifdef __cplusplus
}
#endif
#endif /* ICIKT_H */
