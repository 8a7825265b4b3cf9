// icikt_host.h -- host-side internals shared by icikt_capi.cpp (one device) and icikt_multi.cpp (several
// devices, RCCL).  Internal; the public boundary is include/icikt.h.
#ifndef ICIKT_HOST_H
#define ICIKT_HOST_H

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <functional>
#include <vector>

#include "icikt.h"
#include "icikt_device.h"

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;  // elements
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = n + n / 8 + 64;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), want * sizeof(T));
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct icikt_ctx {
  int device = -1;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;
  hipDeviceProp_t prop{};

  // prepared matrix
  bool prepared = false;
  icikt::PrepView pv{};
  DevBuf<uint16_t> order, hirow, girow, srow;
  DevBuf<uint32_t> wide32;                 // wide columns: order32 | q32 | lo32 | hi32, S x n_pad each
  DevBuf<uint32_t> order_w;                // columns of more than 30 656 rows: `order` as 32-bit words (PrepView::order_w)
  DevBuf<unsigned long long> k0_bits;      // wide columns: K0's phase-3 bitsets, per column of a sort chunk
  DevBuf<uint32_t> rec, tgroups, tprog;
  DevBuf<uint2> smask;
  DevBuf<unsigned long long> meta, sort_keys;
  DevBuf<uint32_t> sort_idx;
  int sort_chunk = 0;
  int64_t alloc_cols = 0;  // columns the prepared-state arrays are allocated for (>= n_samp)

  // pair list
  int64_t n_pairs = -1;
  int64_t pairs_nsamp = -1;  // largest column index + 1 seen in the list
  int n_units = 0;
  int wpb = 0;  // pairs per wave (np) the tasks were built for; 0 = not built
  int group_hint = 0;      // ... and the size of the tie group a random tied pair of their rows sits in (fill groups apart)
  int ntg_hint = -1;       // the most tie groups among the columns whose statistics matrix_tied() read back (-1: not read)
  int tied_state = -1;     // 15 200 .. 30 656 rows (and longer columns: pairs per wave): do the prepared columns hold many tie groups (1), not (0), not asked yet (-1)
  bool raw_valid = false;  // d_raw holds the pair kernel's counts for the current prepared matrix and pair list
  DevBuf<int32_t> d_pi, d_pj, d_unit_start;
  DevBuf<icikt::PairRaw> d_raw;
  DevBuf<int> d_task_ctr;  // persistent pair kernel: one task counter per XCD group, zeroed before every launch
  std::vector<int32_t> h_pi, h_pj, h_units;
  // a combn range [combn_begin, combn_end) of combn(combn_S, 2) set by icikt_set_pairs_combn: the device arrays are
  // filled by a kernel at once, the host copies (h_pi / h_pj: only the host-built task lists read them) on demand
  int64_t combn_S = -1, combn_begin = 0, combn_end = 0;
  bool h_pairs_valid = false;
  void* copy_pool = nullptr;      // worker threads of the staged transfers' host-side copies (icikt_capi.cpp: CopyPool)
  void* pinned_tasks = nullptr;   // pinned staging of the task list the pipelined host path generates chunk by chunk
  size_t pinned_tasks_bytes = 0;
  bool units_dirty = false;   // h_units has been rebuilt on the host and not uploaded yet
  // result copies of 256 KB or more cross a pinned buffer of the library's (one per array of a call, kept from call
  // to call) and are moved to the caller's array by finish_downloads()
  struct Bounce { void* pinned; void* dst; size_t bytes; hipEvent_t ev; };   // ev: recorded behind the piece's D2H copy (may be null)
  struct PinnedSlot { void* p = nullptr; size_t bytes = 0; };
  std::vector<PinnedSlot> out_pinned;   // result downloads: one pinned buffer per array of a call, kept from call to call
  std::vector<Bounce> bounced_out;
  std::vector<hipEvent_t> ev_out;         // events of the result pieces in flight (a pool, reused from call to call)
  size_t ev_out_used = 0;

  // host-path staging: a second stream for H2D copies that run ahead of K0 by column chunks
  hipStream_t copy_stream = nullptr;
  // the pipelined host path (icikt_pairs_f64 / icikt_matrix_f64 on a matrix of several chunks): the pre-pass of chunk
  // k runs on its own stream as soon as the chunk has arrived, and the pair kernel -- one launch per chunk, over the
  // tasks whose LAST column lies in it -- follows on the context's stream while later chunks still cross PCIe
  hipStream_t prep_stream = nullptr;
  std::vector<hipEvent_t> ev_chunk;        // pre-pass of chunk k done
  std::vector<int64_t> chunk_col_end;      // columns [.., chunk_col_end[k]) have arrived with chunk k (this call)
  int pipe_mode = -1;                      // -1: the library's choice; 0 / 1: off / on whenever possible (debug plan)
  hipEvent_t ev_copy[4] = {};
  void* pinned = nullptr;   // pinned staging area: the matrix in column chunks (two halves), pair / task lists in 8 MB pieces
  size_t pinned_bytes = 0;
  bool host_pinned = false; // for the duration of a host entry called with ICIKT_FLAG_HOST_PINNED: the caller has page-locked
                            // the matrix and the result arrays, they are copied from / into directly (PinnedScope)
  DevBuf<double> d_X, d_out4, d_Xp;  // d_Xp: masked column pairs of icikt_pairs_complete_f64
  // full-matrix entry (icikt_matrix_f64): the exclusion rule the pre-pass applies while it reads the matrix and the
  // optional keep bytes it writes (both only for the duration of that call), the assembled matrices, the reduction
  DevBuf<double> d_out5;
  DevBuf<uint8_t> d_keep;
  DevBuf<unsigned long long> d_red;
  DevBuf<int32_t> d_pi_all, d_pj_all;     // icikt_matrix_multi_f64, explicit pair list: the whole list on the first device
  bool k0_small = false;    // (no caller sets it: the 256-thread shape of the pre-pass is reached through the plan key only)
  int k0_shape = -1;        // debug plan key "k0": -1 the library's choice (the large shape), 0 the large shape, 1 the small one
  const icikt::MaskSpec* k0_mask = nullptr;
  uint8_t* k0_keep = nullptr;
  DevBuf<int64_t> d_counts;
  DevBuf<int32_t> d_reasons;
  DevBuf<uint32_t> d_self;

  // launch-plan overrides of the pair kernel (icikt_debug_set_plan; -1 = the library's choice)
  struct PlanOverride {
    int np = -1, pend = -1, wpb = -1, half = -1, grid_mult = -1, grid_cap = -1, hyb = -1;
    bool has_tgmax = false;
    int tgmax = 0;
    int waves = -1;     // half-wave kernels: waves per CU the counter tables may cost the launch down to (default: none)
    int merge = -1;     // pipelined host entries: 1 = the pairs in ONE launch behind the last chunk, 0 = a launch per chunk, whatever the pair count
    int split = -1;     // half-wave kernels: segments per task (1 | 2 | 4), whatever the launch's size
    int solo = -1;      // 0: SOLO steps of the tie program run as MIXED steps (with the in-step chains)
    int list = -1;      // list mode (range counts per listed tie group) up to this many tie groups: count mode takes over above
    bool verbose = false;
  } plan_ov;

  // timing
  // per kernel id a pool of HIP event pairs: several timed launches per step accumulate without a host stall
  // and are folded into ms[] when the figures are read (or when the pool is full)
  struct EvPair { hipEvent_t a = nullptr, b = nullptr; };
  std::vector<EvPair> ev_pool[ICIKT_K_COUNT];
  size_t ev_used[ICIKT_K_COUNT] = {};
  bool ev_open[ICIKT_K_COUNT] = {};
  double ms[ICIKT_K_COUNT] = {};
  int64_t launches[ICIKT_K_COUNT] = {};
};


namespace icikt {
namespace host {

int fail(icikt_ctx* c, int code, const std::string& msg);
int use_device(icikt_ctx* c);

#define HIPCHK(c, call)                                                                               \
  do {                                                                                                \
    hipError_t e__ = (call);                                                                          \
    if (e__ != hipSuccess)                                                                            \
      return icikt::host::fail((c), ICIKT_E_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
  } while (0)

// Allocate the prepared state of an n_feat x n_samp matrix for alloc_cols >= n_samp columns (sort scratch for
// sort_cols columns at a time) and set c->pv.  No kernel is launched.
int prepare_alloc(icikt_ctx* c, int64_t n_feat, int64_t n_samp, int64_t alloc_cols, int64_t sort_cols);
// K0 over columns [col_begin, col_end) of the device matrix dX (leading dimension ld) on c->stream.
int prepare_launch(icikt_ctx* c, const double* dX, int64_t ld, int64_t col_begin, int64_t col_end, hipStream_t stream = nullptr);
// Host matrix -> device (columns [col_begin, col_end) only) overlapped with K0 by column chunks; the device copy
// keeps the full n_feat x n_samp layout (leading dimension n_feat) in c->d_X.  prepare_alloc() must have run.
// pipelined: the pre-pass launches go to c->prep_stream, one event per chunk in c->ev_chunk / c->chunk_col_end; the
// caller makes c->stream wait for them (per chunk, or for the last one), and synchronises c->copy_stream before it
// returns to ITS caller.  prepass: what runs over a chunk once it has arrived -- the full pre-pass (K0), the
// mask-only one of pairwise_completeness (k0_mask: meta alone must be allocated, mask_alloc), or nothing.
enum { kPrepassNone = 0, kPrepassFull = 1, kPrepassMask = 2 };
// Allocate c->pv for the missing-row bitsets alone (meta; no order / rec / sort scratch).  Leaves the context unprepared.
int mask_alloc(icikt_ctx* c, int64_t n_feat, int64_t n_samp);
int upload_and_prepare(icikt_ctx* c, const double* X, int64_t n_feat, int64_t n_samp, int64_t ld, int64_t col_begin,
                       int64_t col_end, uint32_t flags, bool pipelined = false,
                       const std::function<int(size_t, int64_t)>* on_chunk = nullptr,   // pipelined: called per chunk (index, columns arrived)
                       int prepass = kPrepassFull);
// H2D + pre-pass + pair kernel of the host entries: pipelined by chunks when the matrix has several, else in sequence.
// Leaves the pair kernel's counts in c->d_raw (raw_valid): the caller runs the epilogue (icikt_run_dev with
// ICIKT_FLAG_REUSE_COUNTS).
int upload_prepare_pairs(icikt_ctx* c, const double* X, int64_t n_feat, int64_t n_samp, int64_t ld, uint32_t flags);
// global_na values (NaN = NA, +-Inf = Inf, anything else compared with ==) -> the pre-pass's exclusion rule
int make_mask_spec(icikt_ctx* c, const double* global_na, int n_global_na, icikt::MaskSpec* ms);
// Build the pair kernel's task list on the host now (prepare_alloc and a pair list must be in place).
void prebuild_units(icikt_ctx* c);
// Host <-> device copies of the caller's memory.  THE LIBRARY NEVER PAGE-LOCKS CALLER MEMORY (no hipHostRegister /
// hipHostUnregister anywhere in it, since round 4) and never hands pageable memory of 256 KB or more to an asynchronous
// copy: rounds 2 and 3 each saw one GPU memory-access fault at a host heap address inside a host entry, with
// per-call registrations of heap ranges that Python frees and reuses; DESIGN.md section 6 lists what a reading of
// that code found (registrations of neighbouring, non-page-aligned heap ranges set up and torn down independently
// while copies from a neighbour were in flight) and why the mode was deleted rather than repaired.
//   * the matrix: double-buffered column chunks through the library's pinned buffer (hipHostMalloc, kept from call to
//     call), host-side copies on a few threads, each chunk's pre-pass and pair-kernel launches enqueued before the
//     host stages the next chunk;
//   * pair lists: a bounce buffer in 8 MB pieces (upload_sync); results: one pinned buffer per array (download /
//     finish_downloads);
//   * ICIKT_FLAG_HOST_PINNED: the caller states that the matrix and the result arrays of THIS call lie in memory it
//     has page-locked itself (hipHostMalloc / hipHostRegister): they are copied from and into directly.  The library
//     does not probe the caller's memory (hipPointerGetAttributes logs an error for every pageable pointer).
// Copies below kLockMin take the runtime's staging path, which does not touch the caller's pages from the GPU.
constexpr size_t kLockMin = (size_t)256 << 10;
struct PinnedScope {
  icikt_ctx* c;
  PinnedScope(icikt_ctx* ctx, uint32_t flags) : c(ctx) { c->host_pinned = (flags & ICIKT_FLAG_HOST_PINNED) != 0; }
  ~PinnedScope() { c->host_pinned = false; }
  PinnedScope(const PinnedScope&) = delete;
  PinnedScope& operator=(const PinnedScope&) = delete;
};
// grow c->pinned to at least `need` bytes (no copy may be in flight from or into it)
int ensure_bounce(icikt_ctx* c, size_t need);
void destroy_copy_pool(void* pool);
// H2D on c->stream, complete (and the host range released) on return
int upload_sync(icikt_ctx* c, void* dst, const void* src, size_t bytes);
// after the stream that carries download() copies has been synchronised: deliver the bounced ones
void finish_downloads(icikt_ctx* c, bool ok = true);
// The end of a host entry: waits for the stream AND delivers the bounced results piece by piece as their copies
// complete (a piece is moved to the caller's array while the next one still crosses PCIe).  ok: nothing failed so far;
// returns the stream's status.  On an error the caller's arrays may hold part of the results: the call fails.
hipError_t finish_stream(icikt_ctx* c, bool ok);
// D2H of a result array into the caller's buffer on c->stream (not synchronised; finish_downloads afterwards)
int download(icikt_ctx* c, void* dst, const void* src, size_t bytes);

}  // namespace host
}  // namespace icikt
#endif
