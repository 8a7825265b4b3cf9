// icikt_device.h -- structures shared by the kernels (icikt_kernels.hip) and the C-ABI host side
// (icikt_capi.cpp).  Internal; the public boundary is include/icikt.h.
#ifndef ICIKT_DEVICE_H
#define ICIKT_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace icikt {

constexpr int ICIKT_PERSPECTIVE_LOCAL_ = 0;  // == ICIKT_PERSPECTIVE_LOCAL
constexpr int ICIKT_CNT_FIELDS_ = 11;        // == ICIKT_CNT_FIELDS

// Per-column result of the pre-pass (K0).
struct ColStats {
  int32_t nna;       // missing rows
  int32_t ngroups;   // distinct values after the fill (unique(), kendallc.cpp:234-235)
  int32_t tfill;     // size of the fill group (missing rows + rows equal to min-0.1); 0 if nna == 0
  int32_t maxgroup;  // largest tie group: size << 16 | its first ascending position
  uint32_t s0, s1, s2;  // count_rank_tie sums in wrapping int32: t(t-1), t(t-1)(t-2), t(t-1)(2t+5)
  uint32_t ntg;      // tie groups of size >= 2 (length of the column's tgroups list)
  long long e0, e1, e2;  // the same sums exactly
  double fill;           // min - 0.1
  int32_t nexcl;         // rows excluded by the caller's global_na rule (MaskSpec): exclude_loc of R/utils.R:1-23.
                         // == nna unless the data hold NaN and NaN is not in global_na
  int32_t flags;         // bit 0 COL_ODD_TIE: some tie group of >= 2 rows starts at an ODD ascending position;
                         // bits 8..31: what STREAMING the column costs a pair kernel task, in half hot steps (2 per 64-row hot
                         // step, 3 per MIXED / SOLO / GROUP step): the cost-weighted pair blocks of the multi-device driver
};
constexpr int COL_ODD_TIE = 1;
__host__ __device__ inline uint32_t col_stream_cost(const ColStats& st) { return (uint32_t)st.flags >> 8; }

// tie-program entry: rows of the step | kind << 7 | closes << 9 | n0 << 10 (MIXED: rows of its first sub-step)
constexpr uint32_t TPROG_KIND_HOT = 0u, TPROG_KIND_MIXED = 1u, TPROG_KIND_GROUP = 2u, TPROG_KIND_SOLO = 3u;
__host__ __device__ inline uint32_t tprog_rows(uint32_t e) { return e & 127u; }
__host__ __device__ inline uint32_t tprog_kind(uint32_t e) { return (e >> 7) & 3u; }
__host__ __device__ inline bool tprog_closes(uint32_t e) { return ((e >> 9) & 1u) != 0u; }
__host__ __device__ inline uint32_t tprog_n0(uint32_t e) { return (e >> 10) & 63u; }

// setup_missing_matrix (R/utils.R:1-23) on the device: which cells of the data matrix are excluded (become NA,
// R/kendalltau.R:119-121) before the pre-pass.  The pre-pass applies it while it reads the matrix, so the masked copy
// `exclude_data` never exists.  A NaN in the data is missing for ici_kt whether or not NaN is in global_na
// (Rcpp is_na, src/kendallc.cpp:181), but counts as excluded (n_good, keep) only when it is.
constexpr int ICIKT_MASK_VALS = 32;   // distinct finite global_na values the device rule holds (the struct travels as a kernel
                                      // argument); a longer list is masked by the host front-end before the call (api.py, icikt_mi355x.R)
struct MaskSpec {
  int mask_nan;                  // global_na holds NA
  int mask_inf;                  // global_na holds Inf: is.infinite(), both signs
  int n_vals;                    // the other global_na values, compared with ==
  double vals[ICIKT_MASK_VALS];
};
__host__ __device__ inline bool mask_excluded(const MaskSpec& ms, double v) {
  bool ex = (ms.mask_nan && v != v) || (ms.mask_inf && (v - v != 0.0) && v == v);   // v - v: NaN for +-Inf, 0 otherwise
  for (int k = 0; k < ms.n_vals; ++k) ex = ex || (v == ms.vals[k]);   // (n_vals <= ICIKT_MASK_VALS, wave-uniform)
  return ex;
}

// Per-pair integers produced by K1 for the global perspective.
struct PairRaw {
  unsigned long long dis;   // #{(i,j): x_i < x_j, y_i > y_j}
  unsigned long long ntie;  // #{i<j: x_i == x_j, y_i == y_j}
  uint32_t c_both;          // rows missing in both columns
  uint32_t g;               // rows in both fill groups (== c_both unless fl(min-0.1) == min)
};

constexpr int K1_CNT_MIN_GROUPS = 8;        // columns too long for the half-wave kernels: girow is written, and count mode used, above this many tie groups
constexpr uint16_t GIROW_NONE = 0xFFFFu;   // PrepView::girow: the row is its own tie group
// bytes of a counter of count mode (k1_pairs): 2 -- two to a dword, twice the tie groups per LDS byte -- or 4 (a build
// option for measurements: -DICIKT_CNT_BYTES=4)
#ifndef ICIKT_CNT_BYTES
#define ICIKT_CNT_BYTES 2
#endif
static_assert(ICIKT_CNT_BYTES == 2 || ICIKT_CNT_BYTES == 4, "counter width");

// Device pointers + sizes of the prepared matrix (HBM layout).
static_assert(sizeof(ColStats) == 72, "ColStats is 9 words of a column's meta record");

struct PrepView {
  int n;       // n_feat (rows per column)
  int n_pad;   // n rounded up to a multiple of 64
  int n_ord;   // n_pad + 256: stride of `order`, zero padded so that K1 can load rows three steps ahead
  int W;       // ceil(n / 64) bitset words
  int Wp;      // W + 1 (one zero guard word)
  int npow2;   // sort scratch length per column
  int n_samp;
  // per column, stride n_pad
  uint16_t* order;   // [S][n_ord]  row at processing position k (descending value)
  uint32_t* order_w; // [S][n_ord]  the same as 32-bit words, for columns too long for the half-wave kernels (else nullptr): the whole-wave
                     //              kernels reload their ring of rows from it after a step of fewer than 64 rows (k1_pairs)
  int rec_rows;      // rows of a rec / hirow block: n_pad + 8.  Row n_pad is the GUARD ROW of every column: q = n_pad (a
                     // position in the guard word of a pair kernel's bitset), lo = 0, hi = 0 -- the empty lanes of a step
                     // of the tie program name it (srow below), so that they gather "a row that never counts, is never
                     // counted, queries 0 and inserts a bit no query reaches" without an instruction of their own
  uint32_t* rec;     // [S/2][rec_rows][2]  per row: q | lo << 16  (ascending stable position, group start),
                     // the columns 2a and 2a+1 interleaved: one 8-byte gather per row serves both
  uint16_t* hirow;   // [S/2][rec_rows][2]  per row: last ascending position of its tie group, the columns 2a and 2a+1
                     // interleaved like rec (a half-wave GROUP step reads both pairs' ends with one 4-byte gather per row)
  uint16_t* girow;   // [S/2][rec_rows][2]  per row: the place of its tie group in the column's list `tgroups` (groups of
                     // >= 2 rows, ascending), GIROW_NONE for a row that is its own group; interleaved like hirow.  A half-
                     // wave GROUP step counts the streamed group's rows per tie group of the gathered column in a table of
                     // counters indexed by it (count mode, k1_pairs)
  // per column, stride Wp
  // per column one record of mstride = 3 * Wp + 9 words (one array: one collective moves it between ranks):
  //   [Wp] mask      missing rows
  //   [Wp] fillmask  rows in the fill group
  //   [Wp] gflag     bit k: processing position k starts a tie group
  //   [9]  ColStats
  unsigned long long* meta;
  int mstride;
  __host__ __device__ unsigned long long* col_mask(int c) const { return meta + (int64_t)c * mstride; }
  __host__ __device__ unsigned long long* col_fillmask(int c) const { return meta + (int64_t)c * mstride + Wp; }
  __host__ __device__ unsigned long long* col_gflag(int c) const { return meta + (int64_t)c * mstride + 2 * Wp; }
  __host__ __device__ ColStats* col_stats(int c) const {
    return reinterpret_cast<ColStats*>(meta + (int64_t)c * mstride + 3 * Wp);
  }
  uint32_t* tgroups;             // [S][tg_stride] tie groups (size >= 2), ascending: lo | hi << 16
  int tg_stride;                 // n_pad / 2 + 1
  // The tie program of a column as the STREAMED side of a half-wave kernel (n <= 30 656): the steps the pair kernel
  // takes from where the column's tie groups begin, cut and classified ONCE per column by the pre-pass instead of by
  // every one of its S - 1 pairs (k0_tie_program; entry layout: TPROG_*).  Round 4, second half: a step comes with its RECORD --
  //   srow   the rows of the step in the LANE LAYOUT the pair kernel runs it in, 64 entries whatever the step holds
  //          (HOT / GROUP: lane = row of the step; MIXED: lanes 0..31 the first sub-step, lanes 32..63 the second);
  //          an empty lane names the guard row (rec_rows above).  The pair kernel reads them three steps ahead like the
  //          rows of the singleton region: no step re-lays its rows out, rotates its ring or selects guard values;
  //   smask  MIXED steps: per lane l = 0..31 which flags of the pair kernel's two in-step compare vectors (x: vector 1,
  //          y: vector 2; both sub-steps of the lane) belong to pairs INSIDE a tie group.
  // Three guard steps (guard rows only) follow a column's last step: what the pair kernel loads ahead.  All of it is a
  // function of order and gflag: rebuilt, not exchanged, between ranks.
  uint32_t* tprog;               // [S][tp_stride]; tp_stride = n_pad + 2 (a step takes at least one row; 0 ends the list)
  int tp_stride;
  int sr_steps;                  // records per column: n_pad / 17 + 8 (two consecutive steps hold >= 34 rows, see k0_tie_program)
  uint16_t* srow;                // [S][sr_steps][64]
  uint2* smask;                  // [S][sr_steps][32]
  // sort scratch (per column of the current chunk)
  unsigned long long* sort_keys;  // [chunk][npow2]
  uint32_t* sort_idx;             // [chunk][npow2]
  // ---- wide columns (65 535 < n <= ICIKT_MAX_FEATURES_WIDE): 32-bit positions, separate arrays, a plain pair
  // kernel (k1_wide).  order / rec / hirow / tgroups are not allocated then.
  int wide;
  uint32_t* order32;              // [S][n_pad]  row at processing position k (descending value)
  uint32_t* q32;                  // [S][n_pad]  per row: ascending stable position
  uint32_t* lo32;                 // [S][n_pad]  per row: first position of its tie group
  uint32_t* hi32;                 // [S][n_pad]  per row: last position of its tie group
  unsigned long long* k0_bits;    // [chunk][2][Wp + 1]  K0's group-start and fill bitsets (LDS holds them up to 65 535 rows)
};

// one pair per wave: bytes of the counts of `seen` (loc 64 x 16 x u8, lb 64 x u32, hist 64 x u32, locg 64 x 4 x u16)
constexpr int K1_TL_BYTES = 1024 + 256 + 256 + 512 + 64;   // (+ 64: the closed-form tail reuses the area as a u16 prefix over Wp <= 1 025 words)

// half-wave kernels: bytes of a pair's prefix slots (32 lanes x one aligned slot of 8 u16 -- 16 u16 above 8 words per
// lane)
__host__ __device__ inline int k1_half_pre(int half_items) { return 32 * (half_items > 8 ? 16 : 8) * 2; }

// half-wave K1 kernels exist for 1..7, 9, 11, 13 and 15 words per lane of a half's prefix rebuild: n <= 30 656 (the packed
// in-step compares of those kernels need positions -- the guard position 64 W included -- below 2^15).  An EVEN number of
// words per lane from 8 on puts the lanes' words at a stride of 64 / 128 bytes, an 8- / 16-way LDS bank conflict per read
// (8 words measured at half the speed of one pair per wave), so such columns run the next odd kernel.  (Round 2 measured
// 11 and 13 words per lane at 2 waves per SIMD -- the LDS state with `pend` allowed no more -- and did not keep them;
// without `pend` they run 4 waves per SIMD.)
constexpr int ICIKT_HALF_ITEMS_MAX = 15;
__host__ __device__ inline int k1_half_items(int Wp) {
  const int hi = (Wp + 31) >> 5;
  return (hi >= 8 && (hi & 1) == 0) ? hi + 1 : hi;
}

// Stride (in 64-bit words) of a pair's LDS / pend arrays in K1, shared by the kernel and the host plan.  The
// arrays are padded so that the hot steps' prefix rebuilds run without predicates:
//  * half-wave kernels: 32 lanes x half_items words;
//  * one pair per wave: the Wp words, rounded up to a multiple of 8 (two words are read at a time, and the u16 array
//    `ppre` of that length must end on a 16-byte boundary; an owner of the two-level counts has 16 words whatever the
//    length, and the owners past the last word own nothing).
__host__ __device__ inline int k1_lds_stride(int Wp, int half_items) {
  if (half_items > 0) return 32 * half_items;
  return (Wp + 7) & ~7;
}

// ms: cells to exclude while reading dX (nullptr: NaN = missing, nothing else); keep: optional [n_samp][n] bytes,
// 1 = not excluded (the reference's `keep = t(!exclude_loc)`, R/kendalltau.R:417)
// small_shape: 256-thread workgroups (4 waves per column, same results) that fit a CU beside a running pair kernel
hipError_t launch_k0(const PrepView& pv, const double* dX, int64_t ld, int col_begin, int ncols, const MaskSpec* ms,
                     uint8_t* keep, int small_shape, hipStream_t s);
// Full-matrix assembly (scale_and_reshape, R/kendalltau.R:357-421) on the device.
//   launch_out_stats: red[0] = max(taumax) over the pairs, NaN skipped, as a sortable key (0 = none); red[1..5] =
//     pairs per reason code; red[6] = max(n_good)
//   launch_assemble:  five S x S matrices (cor, raw, pvalue, taumax, completeness), symmetric, diagonal rows
// pi / pj == nullptr: pair p is pair `first + p` of combn(S, 2) order.  n_good == nullptr: n - ColStats::nexcl.
hipError_t launch_out_stats(const PrepView& pv, const double* out4, const int32_t* reasons, int64_t n_pairs,
                            const int64_t* n_good, unsigned long long* red, hipStream_t s);
hipError_t launch_assemble(const PrepView& pv, const double* out4, const int32_t* pi, const int32_t* pj, int64_t n_pairs,
                           const int64_t* n_good, const unsigned long long* red, int scale_max, int diag_good,
                           double* out5, hipStream_t s);
// wide columns: one wave per pair, grid of `blocks` single-wave workgroups that fetch pairs from *task_ctr
hipError_t launch_k1_wide(const PrepView& pv, const int32_t* pi, const int32_t* pj, PairRaw* raw, int64_t n_pairs,
                          int blocks, size_t lds_bytes, int* task_ctr, hipStream_t s);
hipError_t k1_wide_blocks_per_cu(size_t lds_bytes, int* out);
inline size_t k1_wide_lds_bytes(int Wp) { return (size_t)(Wp + 1) * (8 + 8 + 4 + 4); }
hipError_t launch_k0_expand(const PrepView& pv, int col_begin, int ncols, hipStream_t s);
// task_ctr: nullptr = the grid covers the task list; else 8 zeroed counters, the waves fetch their tasks (whole-wave kernels)
hipError_t launch_zero_raw(const int32_t* tasks, int n_tasks, PairRaw* raw, hipStream_t s);
hipError_t launch_k1(const PrepView& pv, const int32_t* tasks, int n_tasks, const int32_t* pi,
                     const int32_t* pj, PairRaw* raw, int np, int half_items, int wpb, int blocks,
                     size_t lds_bytes, int perpair_bytes, int* task_ctr, int opts, hipStream_t s);
hipError_t k1_blocks_per_cu(int np, int half_items, int wpb, size_t lds_bytes, int* out);
hipError_t launch_k2(const PrepView& pv, const int32_t* pi, const int32_t* pj, const PairRaw* raw,
                     int64_t n_pairs, int perspective, int alternative, int continuity, int exact64,
                     double* out4, int64_t* counts, int32_t* reasons, hipStream_t s);
// the mask-only pre-pass of pairwise_completeness: missing-row bitsets (NaN = missing) of columns [col_begin, col_begin + ncols)
hipError_t launch_k0_mask(const PrepView& pv, const double* dX, int64_t ld, int col_begin, int ncols, hipStream_t s);
hipError_t launch_missingness(const PrepView& pv, const int32_t* pi, const int32_t* pj, int64_t n_pairs,
                              int64_t* missing, hipStream_t s);
hipError_t launch_mask_pairs(const double* dX, int64_t ld, int n, const int32_t* pi, const int32_t* pj, int64_t first,
                             int64_t npairs, double* dXp, hipStream_t s);
// pi / pj of pairs [begin, begin + count) of combn(S, 2) order, computed on the device
hipError_t launch_fill_combn(int32_t* pi, int32_t* pj, int64_t S, int64_t begin, int64_t count, hipStream_t s);
hipError_t launch_selftest(uint32_t* d_out, hipStream_t s);
hipError_t read_step_stats(unsigned long long* out24, int reset);

}  // namespace icikt
#endif
