// icikt_multi.cpp -- several MI355X behind ONE call of the C ABI (include/icikt.h, icikt_multi_*).
//
// The reference fans its pair chunks out to worker processes (computation$split_fun over `core` chunks,
// R/kendalltau.R:158, chunks from :250-255, workers from R/utils.R:68-80) and ships the whole matrix to every
// worker.  Here one host thread drives each GPU (HIP and fork() do not mix, so the GPUs cannot be furrr
// workers), all inside one call from the R main thread:
//
//   rank r:  H2D of ITS columns only -> K0 on them              (the matrix crosses PCIe once, 1/G per link)
//            RCCL all-gather of `order` and `meta` over xGMI    (24 KB per column of 10 000 rows)
//            k0_expand: rec / hirow / tgroups of the received columns rebuilt locally
//            K1 + K2 over block r of the pair list: ceiling(P / G) consecutive pairs = the reference's `core` chunk
//            RCCL gather of the P/G x 4 results to rank 0 -> one D2H into the caller's buffers
//
// Exchange "copy" replaces the two collectives by device-to-device copies between the ranks' buffers
// (hipMemcpyPeerAsync): it is what runs when a device is listed more than once -- RCCL refuses duplicate
// devices -- which is how the whole flow is rehearsed on a one-GPU box.  There is no CPU path.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "icikt.h"
#include "icikt_device.h"
#include "icikt_host.h"

namespace {

// reusable barrier for the rank threads of one call
class Barrier {
 public:
  explicit Barrier(int n) : n_(n) {}
  void wait() {
    std::unique_lock<std::mutex> lk(mu_);
    if (aborted_) return;
    const int gen = gen_;
    if (++count_ == n_) {
      count_ = 0;
      ++gen_;
      cv_.notify_all();
    } else {
      cv_.wait(lk, [&] { return gen_ != gen || aborted_; });
    }
  }
  // releases every waiter, now and later (a rank thread could not be started: the others must not wait for it)
  void abort() {
    std::lock_guard<std::mutex> lk(mu_);
    aborted_ = true;
    cv_.notify_all();
  }

 private:
  std::mutex mu_;
  std::condition_variable cv_;
  int n_, count_ = 0, gen_ = 0;
  bool aborted_ = false;
};

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

struct icikt_multi {
  int n = 0;
  std::vector<int> devices;
  std::vector<icikt_ctx*> ctx;
  bool rccl = false;
  bool rccl_asked = false;                            // exchange = RCCL was asked for by name (a one-rank handle then runs the whole flow)
  std::vector<ncclComm_t> comms;
  std::string err;
  double phase_ms[ICIKT_MULTI_PHASES] = {};          // of the last call: maximum over the ranks
  std::vector<double> rank_ms;                        // [rank][ICIKT_MULTI_PHASES + 1]: its phases, then its barrier waits
  int ranks_used = 0;                                 // of the last call: n, or 1 when it ran on the first device alone
  std::vector<int64_t> bounds;                        // of the last call: rank r ran pairs [bounds[r], bounds[r + 1]) of the list
  void* root_compact = nullptr;                       // cost-weighted blocks, matrix entry: the gathered slots made contiguous
  size_t root_compact_bytes = 0;
  // rank 0's gather targets
  void* root_out4 = nullptr;
  void* root_counts = nullptr;
  void* root_reasons = nullptr;
  size_t root_out4_bytes = 0, root_counts_bytes = 0, root_reasons_bytes = 0;
};

extern "C" int icikt_cost_blocks(const uint32_t* col_cost, int64_t n_samp, const int32_t* pj, int64_t n_pairs, int n_blocks,
                                 int64_t max_block, int64_t* bounds) {
  // integer arithmetic throughout: every rank must cut at exactly the same pairs
  if (!col_cost || !bounds || n_samp < 0 || n_blocks < 1 || (n_pairs >= 0 && n_pairs > 0 && !pj)) return ICIKT_E_INVALID;
  const int64_t S = n_samp;
  const int G = n_blocks;
  const bool list = n_pairs >= 0;
  const int64_t P = list ? n_pairs : S * (S - 1) / 2;
  if (max_block <= 0) max_block = P;
  if ((__int128)max_block * G < (__int128)P) return ICIKT_E_INVALID;   // the blocks could not hold the list
  std::vector<uint64_t> cost((size_t)std::max<int64_t>(S, 1)), pre((size_t)S + 1, 0);   // pre[j] = cost of columns 0 .. j-1
  for (int64_t j = 0; j < S; ++j) {
    cost[(size_t)j] = std::max<uint32_t>(1u, col_cost[(size_t)j]);
    pre[(size_t)j + 1] = pre[(size_t)j] + cost[(size_t)j];
  }
  for (int q = 0; q <= G; ++q) bounds[q] = 0;
  // pair p of the list "starts at" the cost of the pairs before it; block k begins with the first pair that starts at
  // or beyond total * k / G
  int k = 1;
  if (list) {
    uint64_t total = 0, run = 0;
    for (int64_t p = 0; p < P; ++p) {
      if (pj[p] < 0 || pj[p] >= S) return ICIKT_E_INVALID;
      total += cost[(size_t)pj[p]];
    }
    for (int64_t p = 0; p < P && k < G; ++p) {
      while (k < G && run * (uint64_t)G >= total * (uint64_t)k) bounds[k++] = p;
      run += cost[(size_t)pj[p]];
    }
  } else if (S >= 2) {
    // combn order: row i holds the pairs (i, i + 1 .. S - 1), each streaming its second column; a row costs
    // pre[S] - pre[i + 1], so whole rows are skipped and a cut is a binary search inside one row
    uint64_t total = 0, run = 0;
    for (int64_t j = 1; j < S; ++j) total += cost[(size_t)j] * (uint64_t)j;
    int64_t p_row = 0;
    for (int64_t i = 0; i + 1 < S && k < G; ++i) {
      const uint64_t base = pre[(size_t)i + 1], row_cost = pre[(size_t)S] - base;
      while (k < G) {
        const uint64_t target = (total * (uint64_t)k + (uint64_t)G - 1) / (uint64_t)G;   // run' >= total k / G
        if (target > run + row_cost - cost[(size_t)S - 1]) break;                      // beyond the row's last pair
        // first j in (i, S) with run + pre[j] - base >= target
        const uint64_t want = (target > run) ? target - run + base : base;
        const int64_t j = std::max<int64_t>(i + 1, std::lower_bound(pre.begin() + (i + 1), pre.begin() + S, want) - pre.begin());
        bounds[k++] = p_row + (j - i - 1);
      }
      run += row_cost;
      p_row += S - 1 - i;
    }
  }
  while (k <= G) bounds[k++] = P;
  for (int q = 1; q <= G; ++q) {   // a block holds at most max_block pairs, and what is left must fit the blocks behind it
    bounds[q] = std::min(bounds[q], bounds[q - 1] + max_block);
    bounds[q] = std::max(bounds[q], P - (int64_t)(G - q) * max_block);
    bounds[q] = std::max(bounds[q], bounds[q - 1]);
  }
  bounds[G] = P;
  return ICIKT_SUCCESS;
}

namespace {

int mfail(icikt_multi* m, int code, const std::string& msg) {
  if (m) m->err = msg;
  return code;
}

struct Call {
  icikt_multi* m;
  const double* X;
  int64_t n_feat, n_samp, ld;
  const int32_t *pi, *pj;
  int64_t P, n_each;
  int64_t slot;                   // pairs a rank's result buffers hold (= n_each; 2 n_each with cost-weighted blocks)
  bool balance;                   // ICIKT_FLAG_BALANCE_COST: block boundaries by the streamed columns' cost, not by count
  std::vector<int64_t> bounds;    // [G + 1]; equal counts: known up front; cost-weighted: every rank computes the same in phase B
  int perspective, alternative, continuity;
  uint32_t flags;
  double* out4;
  int64_t* counts;
  int32_t* reasons;
  int64_t cols_per, alloc_cols;
  Barrier* bar;
  std::vector<int> rc;            // per rank
  std::vector<std::string> msg;   // per rank
  bool timing;
  // what the peers read from each other in "copy" exchange
  std::vector<char*> order_base, meta_base;
  std::vector<double*> out4_dev;
  std::vector<int64_t*> counts_dev;
  std::vector<int32_t*> reasons_dev;
  // icikt_matrix_multi_f64: the pre-pass applies the exclusion rule, every rank returns the keep bytes of ITS columns,
  // the first rank assembles the five matrices from the gathered results
  bool matrix = false;
  icikt::MaskSpec mask{};
  uint8_t* keep = nullptr;
  double* out5 = nullptr;
  int scale_max = 1, diag_good = 1;
  unsigned long long red[8] = {};
};

bool all_ok(Call& a) {
  for (int r : a.rc)
    if (r != ICIKT_SUCCESS) return false;
  return true;
}

// device-to-device copy between two ranks' buffers on the destination rank's stream
hipError_t peer_copy(void* dst, int dst_dev, const void* src, int src_dev, size_t bytes, hipStream_t s) {
  if (bytes == 0) return hipSuccess;
  if (dst_dev == src_dev) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s);
  return hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, s);
}

hipError_t grow(void** p, size_t* cap, size_t want) {
  if (want <= *cap) return hipSuccess;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *cap = 0;
  hipError_t e = hipMalloc(p, want);
  if (e == hipSuccess) *cap = want;
  return e;
}

#define RANKCHK_HIP(call)                                                                        \
  do {                                                                                           \
    hipError_t e__ = (call);                                                                     \
    if (e__ != hipSuccess) {                                                                     \
      a.rc[r] = ICIKT_E_HIP;                                                                     \
      a.msg[r] = std::string(#call) + ": " + hipGetErrorString(e__);                             \
      return;                                                                                    \
    }                                                                                            \
  } while (0)
#define RANKCHK_NCCL(call)                                                                       \
  do {                                                                                           \
    ncclResult_t e__ = (call);                                                                   \
    if (e__ != ncclSuccess) {                                                                    \
      a.rc[r] = ICIKT_E_HIP;                                                                     \
      a.msg[r] = std::string(#call) + ": " + ncclGetErrorString(e__);                            \
      return;                                                                                    \
    }                                                                                            \
  } while (0)
#define RANKCHK(call)                                                                            \
  do {                                                                                           \
    int rc__ = (call);                                                                           \
    if (rc__ != ICIKT_SUCCESS) {                                                                 \
      a.rc[r] = rc__;                                                                            \
      a.msg[r] = icikt_last_error(c);                                                            \
      return;                                                                                    \
    }                                                                                            \
  } while (0)

// One rank = one host thread = one device.  Every rank passes every barrier, whatever happened to it: a rank
// that failed skips the work of the later phases, and a failure anywhere makes ALL ranks skip the collectives
// (they are entered by all ranks or by none).
void rank_main(Call& a, int r) {
  icikt_multi* m = a.m;
  icikt_ctx* c = m->ctx[(size_t)r];
  const int G = m->n;
  const int64_t S = a.n_samp;
  const int64_t c0 = std::min(S, (int64_t)r * a.cols_per), c1 = std::min(S, (int64_t)(r + 1) * a.cols_per);
  int64_t begin = a.bounds[(size_t)r], end = a.bounds[(size_t)r + 1];   // (cost-weighted blocks: set in phase B)
  int64_t P_local = end - begin;
  double t_prev = now_ms();
  double* my_ms = m->rank_ms.data() + (size_t)r * (ICIKT_MULTI_PHASES + 1);
  // every rank keeps the wall clock of ITS phases (with ICIKT_FLAG_TIMING: after a stream synchronisation, so the
  // figures are device time) and, apart from them, the time it spent waiting for the other ranks at the barriers:
  // an imbalance between the ranks shows as a spread of the per-rank figures, not as a long next phase of rank 0
  auto mark = [&](int phase) {
    if (a.timing) (void)hipStreamSynchronize(c->stream);
    const double t = now_ms();
    my_ms[phase] += t - t_prev;
    t_prev = t;
  };
  auto rendezvous = [&]() {
    a.bar->wait();
    const double t = now_ms();
    my_ms[ICIKT_MULTI_PHASES] += t - t_prev;
    t_prev = t;
  };
  size_t order_slice = 0, meta_slice = 0;

  // ---- phase A: this rank's pair block, its share of the columns: H2D + K0 ---------------------------------
  auto phase_a = [&]() {
    RANKCHK(icikt::host::use_device(c));
    if (!a.balance) {
      if (a.pi) RANKCHK(icikt_set_pairs(c, a.pi + begin, a.pj + begin, P_local));
      else RANKCHK(icikt_set_pairs_combn(c, S, begin, end));
    }
    RANKCHK(icikt::host::prepare_alloc(c, a.n_feat, S, a.alloc_cols, std::max<int64_t>(c1 - c0, 1)));
    if (a.matrix) {
      c->k0_mask = &a.mask;
      if (a.keep) {
        RANKCHK_HIP(c->d_keep.reserve((size_t)S * (size_t)a.n_feat));
        c->k0_keep = c->d_keep.p;
      }
    }
    const int rc_up = icikt::host::upload_and_prepare(c, a.X, a.n_feat, S, a.ld, c0, c1, a.flags & ~ICIKT_FLAG_TIMING);
    c->k0_mask = nullptr;
    c->k0_keep = nullptr;
    RANKCHK(rc_up);
    if (!a.balance) icikt::host::prebuild_units(c);   // the task list of this rank's pair block, while its columns are copied
    RANKCHK_HIP(c->d_out4.reserve((size_t)std::max<int64_t>(a.slot, 1) * 4));
    if (a.counts) RANKCHK_HIP(c->d_counts.reserve((size_t)std::max<int64_t>(a.slot, 1) * ICIKT_CNT_FIELDS));
    if (a.reasons) RANKCHK_HIP(c->d_reasons.reserve((size_t)std::max<int64_t>(a.slot, 1)));
    if (a.matrix && !a.reasons) RANKCHK_HIP(c->d_reasons.reserve((size_t)std::max<int64_t>(a.slot, 1)));
    if (r == 0 && a.matrix) {
      RANKCHK_HIP(grow(&m->root_reasons, &m->root_reasons_bytes, (size_t)G * a.slot * sizeof(int32_t)));
      RANKCHK_HIP(c->d_out5.reserve(5 * (size_t)S * (size_t)S));
      RANKCHK_HIP(c->d_red.reserve(8));
      if (a.pi) {   // the assembly needs the WHOLE pair list on the first device (combn order is computed)
        RANKCHK_HIP(c->d_pi_all.reserve((size_t)a.P));
        RANKCHK_HIP(c->d_pj_all.reserve((size_t)a.P));
        RANKCHK(icikt::host::upload_sync(c, c->d_pi_all.p, a.pi, (size_t)a.P * sizeof(int32_t)));
        RANKCHK(icikt::host::upload_sync(c, c->d_pj_all.p, a.pj, (size_t)a.P * sizeof(int32_t)));
      }
    }
    if (r == 0 && a.balance && a.matrix)
      RANKCHK_HIP(grow(&m->root_compact, &m->root_compact_bytes, (size_t)a.P * 4 * sizeof(double) + (size_t)a.P * sizeof(int32_t) + 64));
    if (r == 0) {
      RANKCHK_HIP(grow(&m->root_out4, &m->root_out4_bytes, (size_t)G * a.slot * 4 * sizeof(double)));
      if (a.counts)
        RANKCHK_HIP(grow(&m->root_counts, &m->root_counts_bytes, (size_t)G * a.slot * ICIKT_CNT_FIELDS * sizeof(int64_t)));
      if (a.reasons) RANKCHK_HIP(grow(&m->root_reasons, &m->root_reasons_bytes, (size_t)G * a.slot * sizeof(int32_t)));
    }
    order_slice = (size_t)a.cols_per * (size_t)c->pv.n_ord * sizeof(uint16_t);
    meta_slice = (size_t)a.cols_per * (size_t)c->pv.mstride * sizeof(unsigned long long);
    a.order_base[(size_t)r] = reinterpret_cast<char*>(c->pv.order);
    a.meta_base[(size_t)r] = reinterpret_cast<char*>(c->pv.meta);
    a.out4_dev[(size_t)r] = c->d_out4.p;
    a.counts_dev[(size_t)r] = c->d_counts.p;
    a.reasons_dev[(size_t)r] = c->d_reasons.p;
    if (!m->rccl) RANKCHK_HIP(hipStreamSynchronize(c->stream));  // peers read this rank's slices after the barrier
  };
  phase_a();
  mark(ICIKT_MULTI_PHASE_PREPARE);
  rendezvous();
  if (!all_ok(a)) {
    (void)hipStreamSynchronize(c->stream);
    return;
  }

  // ---- phase B: all-gather of order + meta, local rebuild of the rest, pair kernel + epilogue ---------------
  auto phase_b = [&]() {
    if (m->rccl) {
      // in place: rank r's slice already sits at offset r * slice of its own array
      RANKCHK_NCCL(ncclAllGather(a.order_base[(size_t)r] + (size_t)r * order_slice, a.order_base[(size_t)r], order_slice,
                                 ncclUint8, m->comms[(size_t)r], c->stream));
      RANKCHK_NCCL(ncclAllGather(a.meta_base[(size_t)r] + (size_t)r * meta_slice, a.meta_base[(size_t)r], meta_slice,
                                 ncclUint8, m->comms[(size_t)r], c->stream));
    } else {
      for (int p = 0; p < G; ++p) {
        if (p == r) continue;
        RANKCHK_HIP(peer_copy(a.order_base[(size_t)r] + (size_t)p * order_slice, m->devices[(size_t)r],
                              a.order_base[(size_t)p] + (size_t)p * order_slice, m->devices[(size_t)p], order_slice, c->stream));
        RANKCHK_HIP(peer_copy(a.meta_base[(size_t)r] + (size_t)p * meta_slice, m->devices[(size_t)r],
                              a.meta_base[(size_t)p] + (size_t)p * meta_slice, m->devices[(size_t)p], meta_slice, c->stream));
      }
    }
    c->prepared = true;
    if (c0 > 0) RANKCHK(icikt_expand_cols_dev(c, 0, c0, 0));
    if (c1 < S) RANKCHK(icikt_expand_cols_dev(c, c1, S, 0));
    if (a.balance) {
      // Cost-weighted blocks: the pre-pass leaves with every column what STREAMING it costs a pair-kernel task
      // (ColStats::flags, bits 8..: hot / MIXED / GROUP steps weighted by their instructions); a pair (i, j) streams j.
      // Every rank holds every column's statistics now and cuts the list at the same places: consecutive blocks of
      // equal COST, no block longer than twice the equal share (the result buffers' size).
      std::vector<int32_t> fl((size_t)S);
      RANKCHK_HIP(hipMemcpy2DAsync(fl.data(), sizeof(int32_t), &c->pv.col_stats(0)->flags,
                                   (size_t)c->pv.mstride * sizeof(unsigned long long), sizeof(int32_t), (size_t)S,
                                   hipMemcpyDeviceToHost, c->stream));
      RANKCHK_HIP(hipStreamSynchronize(c->stream));
      std::vector<int64_t> b((size_t)G + 1, 0);
      {
        std::vector<uint32_t> cost((size_t)S);
        for (int64_t j = 0; j < S; ++j) cost[(size_t)j] = (uint32_t)fl[(size_t)j] >> 8;
        const int rcc = icikt_cost_blocks(cost.data(), S, a.pj, a.pi ? a.P : -1, G, a.slot, b.data());
        if (rcc != ICIKT_SUCCESS) { a.rc[r] = rcc; a.msg[r] = "cost-weighted blocks: bad arguments"; return; }
      }
      begin = b[(size_t)r]; end = b[(size_t)r + 1]; P_local = end - begin;
      if (r == 0) a.bounds = b;   // (the ranks agree; rank 0's copy is the call's record and what phase C reads)
      if (a.pi) RANKCHK(icikt_set_pairs(c, a.pi + begin, a.pj + begin, P_local));
      else RANKCHK(icikt_set_pairs_combn(c, S, begin, end));
    }
    if (a.timing) mark(ICIKT_MULTI_PHASE_EXCHANGE);   // (without the flag the exchange is only enqueued: counted with the pairs)
    RANKCHK(icikt_run_dev(c, a.perspective, a.alternative, a.continuity, a.flags & ~ICIKT_FLAG_TIMING, c->d_out4.p,
                          a.counts ? c->d_counts.p : nullptr, (a.reasons || a.matrix) ? c->d_reasons.p : nullptr));
    if (!m->rccl) RANKCHK_HIP(hipStreamSynchronize(c->stream));  // rank 0 reads the results after the barrier
  };
  phase_b();
  mark(ICIKT_MULTI_PHASE_PAIRS);
  rendezvous();
  // NB: a rank that failed inside phase B may have left its peers inside a collective that it never entered;
  // everything that can fail for reasons of its own (allocation, argument checks) happens in phase A for that
  // reason, and phase B failures are launch failures that hit every rank alike.
  if (!all_ok(a)) {
    (void)hipStreamSynchronize(c->stream);
    return;
  }

  // ---- phase C: gather to rank 0, one D2H -------------------------------------------------------------------
  auto phase_c = [&]() {
    const size_t n4 = (size_t)a.slot * 4;
    if (m->rccl) {
      RANKCHK_NCCL(ncclGather(c->d_out4.p, m->root_out4, n4, ncclDouble, 0, m->comms[(size_t)r], c->stream));
      if (a.counts)
        RANKCHK_NCCL(ncclGather(c->d_counts.p, m->root_counts, (size_t)a.slot * ICIKT_CNT_FIELDS, ncclInt64, 0,
                                m->comms[(size_t)r], c->stream));
      if (a.reasons || a.matrix)
        RANKCHK_NCCL(ncclGather(c->d_reasons.p, m->root_reasons, (size_t)a.slot, ncclInt32, 0, m->comms[(size_t)r], c->stream));
    } else if (r == 0) {
      for (int p = 0; p < G; ++p) {
        RANKCHK_HIP(peer_copy(static_cast<double*>(m->root_out4) + (size_t)p * n4, m->devices[0], a.out4_dev[(size_t)p],
                              m->devices[(size_t)p], n4 * sizeof(double), c->stream));
        if (a.counts)
          RANKCHK_HIP(peer_copy(static_cast<int64_t*>(m->root_counts) + (size_t)p * a.slot * ICIKT_CNT_FIELDS, m->devices[0],
                                a.counts_dev[(size_t)p], m->devices[(size_t)p],
                                (size_t)a.slot * ICIKT_CNT_FIELDS * sizeof(int64_t), c->stream));
        if (a.reasons || a.matrix)
          RANKCHK_HIP(peer_copy(static_cast<int32_t*>(m->root_reasons) + (size_t)p * a.slot, m->devices[0],
                                a.reasons_dev[(size_t)p], m->devices[(size_t)p], (size_t)a.slot * sizeof(int32_t), c->stream));
      }
    }
    if (a.matrix) {
      // the keep bytes of this rank's columns; on the first rank the assembly over the gathered results (the blocks
      // are consecutive and only the last one is short: the first P records ARE the pair list's results, in order)
      if (a.keep && c1 > c0)
        RANKCHK(icikt::host::download(c, a.keep + (size_t)c0 * (size_t)a.n_feat, c->d_keep.p + (size_t)c0 * (size_t)a.n_feat,
                                      (size_t)(c1 - c0) * (size_t)a.n_feat));
      if (r == 0) {
        const double* all4 = static_cast<const double*>(m->root_out4);
        const int32_t* allr = static_cast<const int32_t*>(m->root_reasons);
        if (a.balance) {
          // cost-weighted blocks have different lengths: the ranks' slots are made one contiguous list, in list order
          double* c4 = static_cast<double*>(m->root_compact);
          int32_t* cr = reinterpret_cast<int32_t*>(c4 + (size_t)a.P * 4);
          for (int p = 0; p < G; ++p) {
            const size_t len = (size_t)(a.bounds[(size_t)p + 1] - a.bounds[(size_t)p]);
            if (len == 0) continue;
            RANKCHK_HIP(hipMemcpyAsync(c4 + (size_t)a.bounds[(size_t)p] * 4, all4 + (size_t)p * n4, len * 4 * sizeof(double),
                                       hipMemcpyDeviceToDevice, c->stream));
            RANKCHK_HIP(hipMemcpyAsync(cr + (size_t)a.bounds[(size_t)p], allr + (size_t)p * (size_t)a.slot, len * sizeof(int32_t),
                                       hipMemcpyDeviceToDevice, c->stream));
          }
          all4 = c4; allr = cr;
        }
        RANKCHK_HIP(icikt::launch_out_stats(c->pv, all4, allr, a.P, nullptr, c->d_red.p, c->stream));
        RANKCHK_HIP(icikt::launch_assemble(c->pv, all4, a.pi ? c->d_pi_all.p : nullptr,
                                           a.pi ? c->d_pj_all.p : nullptr, a.P, nullptr, c->d_red.p, a.scale_max, a.diag_good,
                                           c->d_out5.p, c->stream));
        RANKCHK(icikt::host::download(c, a.out5, c->d_out5.p, 5 * (size_t)S * (size_t)S * sizeof(double)));
        RANKCHK(icikt::host::download(c, a.red, c->d_red.p, sizeof(a.red)));
      }
    } else if (r == 0 && !a.balance) {
      // blocks are consecutive and only the last one is short: the first P records of the gathered array
      RANKCHK(icikt::host::download(c, a.out4, m->root_out4, (size_t)a.P * 4 * sizeof(double)));
      if (a.counts) RANKCHK(icikt::host::download(c, a.counts, m->root_counts, (size_t)a.P * ICIKT_CNT_FIELDS * sizeof(int64_t)));
      if (a.reasons) RANKCHK(icikt::host::download(c, a.reasons, m->root_reasons, (size_t)a.P * sizeof(int32_t)));
    } else if (r == 0) {
      // cost-weighted blocks: every rank's slot holds a block of its own length -> one copy per rank and array
      for (int p = 0; p < G; ++p) {
        const size_t b0 = (size_t)a.bounds[(size_t)p], len = (size_t)(a.bounds[(size_t)p + 1] - a.bounds[(size_t)p]);
        if (len == 0) continue;
        RANKCHK(icikt::host::download(c, a.out4 + b0 * 4, static_cast<const double*>(m->root_out4) + (size_t)p * n4,
                                      len * 4 * sizeof(double)));
        if (a.counts)
          RANKCHK(icikt::host::download(c, a.counts + b0 * ICIKT_CNT_FIELDS,
                                        static_cast<const int64_t*>(m->root_counts) + (size_t)p * (size_t)a.slot * ICIKT_CNT_FIELDS,
                                        len * ICIKT_CNT_FIELDS * sizeof(int64_t)));
        if (a.reasons)
          RANKCHK(icikt::host::download(c, a.reasons + b0, static_cast<const int32_t*>(m->root_reasons) + (size_t)p * (size_t)a.slot,
                                        len * sizeof(int32_t)));
      }
    }
    const hipError_t es = icikt::host::finish_stream(c, true);   // (delivers the bounced results piece by piece as they arrive)
    RANKCHK_HIP(es);
  };
  phase_c();
  if (!c->bounced_out.empty()) {   // phase C left early: nothing may still write the caller's arrays
    (void)hipStreamSynchronize(c->stream);
    icikt::host::finish_downloads(c, false);
  }
  mark(ICIKT_MULTI_PHASE_GATHER);
  rendezvous();  // "copy" exchange: nobody returns (and lets its buffers be reused) while rank 0 still reads them
}

}  // namespace

extern "C" {

int icikt_multi_create(const int* devices, int n_gpu, int exchange, icikt_multi** out) {
  if (!out) return ICIKT_E_INVALID;
  *out = nullptr;
  if (n_gpu < 1 || n_gpu > 64) return ICIKT_E_INVALID;
  if (exchange != ICIKT_MULTI_EXCHANGE_AUTO && exchange != ICIKT_MULTI_EXCHANGE_RCCL && exchange != ICIKT_MULTI_EXCHANGE_COPY)
    return ICIKT_E_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ICIKT_E_NO_DEVICE;
  icikt_multi* m = new (std::nothrow) icikt_multi();
  if (!m) return ICIKT_E_NOMEM;
  m->n = n_gpu;
  bool distinct = true;
  for (int r = 0; r < n_gpu; ++r) {
    const int d = devices ? devices[r] : r;
    if (d < 0 || d >= ndev) {
      icikt_multi_destroy(m);
      return ICIKT_E_INVALID;
    }
    for (int q : m->devices) distinct = distinct && (q != d);
    m->devices.push_back(d);
  }
  if (exchange == ICIKT_MULTI_EXCHANGE_RCCL && !distinct) {  // RCCL refuses a device listed twice
    icikt_multi_destroy(m);
    return ICIKT_E_INVALID;
  }
  m->rccl = (exchange == ICIKT_MULTI_EXCHANGE_RCCL) || (exchange == ICIKT_MULTI_EXCHANGE_AUTO && distinct);
  m->rccl_asked = exchange == ICIKT_MULTI_EXCHANGE_RCCL;
  for (int r = 0; r < n_gpu; ++r) {
    icikt_ctx* c = nullptr;
    const int rc = icikt_ctx_create(m->devices[(size_t)r], &c);
    if (rc != ICIKT_SUCCESS) {
      icikt_multi_destroy(m);
      return rc;
    }
    m->ctx.push_back(c);
  }
  if (m->rccl) {
    m->comms.assign((size_t)n_gpu, nullptr);
    if (ncclCommInitAll(m->comms.data(), n_gpu, m->devices.data()) != ncclSuccess) {
      m->comms.clear();
      icikt_multi_destroy(m);
      return ICIKT_E_HIP;
    }
  }
  *out = m;
  return ICIKT_SUCCESS;
}

void icikt_multi_destroy(icikt_multi* m) {
  if (!m) return;
  for (size_t r = 0; r < m->comms.size(); ++r)
    if (m->comms[r]) {
      (void)hipSetDevice(m->devices[r]);
      (void)ncclCommDestroy(m->comms[r]);
    }
  if (!m->devices.empty()) (void)hipSetDevice(m->devices[0]);
  if (m->root_out4) (void)hipFree(m->root_out4);
  if (m->root_counts) (void)hipFree(m->root_counts);
  if (m->root_reasons) (void)hipFree(m->root_reasons);
  if (m->root_compact) (void)hipFree(m->root_compact);
  for (icikt_ctx* c : m->ctx) icikt_ctx_destroy(c);
  delete m;
}

const char* icikt_multi_last_error(const icikt_multi* m) { return m ? m->err.c_str() : "null handle"; }
int icikt_multi_n_gpu(const icikt_multi* m) { return m ? m->n : 0; }
int icikt_multi_uses_rccl(const icikt_multi* m) { return (m && m->rccl) ? 1 : 0; }
int icikt_multi_comm_ranks(const icikt_multi* m) {
  if (!m || !m->rccl || m->comms.empty() || !m->comms[0]) return 0;
  int n = 0;
  return ncclCommCount(m->comms[0], &n) == ncclSuccess ? n : -1;
}

int icikt_multi_phase_ms(const icikt_multi* m, double* ms) {
  if (!m || !ms) return ICIKT_E_INVALID;
  for (int k = 0; k < ICIKT_MULTI_PHASES; ++k) ms[k] = m->phase_ms[k];
  return ICIKT_SUCCESS;
}

int icikt_multi_rank_phase_ms(const icikt_multi* m, int rank, double* ms) {
  if (!m || !ms || rank < 0 || rank >= m->n) return ICIKT_E_INVALID;
  for (int k = 0; k <= ICIKT_MULTI_PHASES; ++k)
    ms[k] = (rank < m->ranks_used) ? m->rank_ms[(size_t)rank * (ICIKT_MULTI_PHASES + 1) + (size_t)k] : 0.0;
  return ICIKT_SUCCESS;
}

int icikt_multi_ranks_used(const icikt_multi* m) { return m ? m->ranks_used : 0; }

int icikt_multi_block_bounds(const icikt_multi* m, int64_t* bounds) {
  if (!m || !bounds) return ICIKT_E_INVALID;
  if ((int)m->bounds.size() != m->ranks_used + 1) return ICIKT_E_STATE;
  for (size_t k = 0; k < m->bounds.size(); ++k) bounds[k] = m->bounds[k];
  return ICIKT_SUCCESS;
}

int icikt_multi_debug_set_plan(icikt_multi* m, const char* spec) {
  if (!m) return ICIKT_E_INVALID;
  for (icikt_ctx* c : m->ctx) {
    const int rc = icikt_debug_set_plan(c, spec);
    if (rc) return mfail(m, rc, icikt_last_error(c));
  }
  return ICIKT_SUCCESS;
}

}  // extern "C"

namespace {

// what icikt_matrix_multi_f64 adds to the arguments of icikt_pairs_multi_f64
struct MatrixArgs {
  const double* global_na;
  int n_global_na, scale_max, diag_good;
  double* out5;
  uint8_t* keep;
  int64_t* reason_counts;
};

int multi_impl(icikt_multi* m, const double* X, int64_t n_feat, int64_t n_samp, int64_t ld,
               const int32_t* pi, const int32_t* pj, int64_t n_pairs, int perspective, int alternative,
               int continuity, uint32_t flags, double* out4, int64_t* counts, int32_t* reasons, const MatrixArgs* mx) {
  if (!m) return ICIKT_E_INVALID;
  for (double& v : m->phase_ms) v = 0.0;
  m->rank_ms.assign((size_t)m->n * (ICIKT_MULTI_PHASES + 1), 0.0);
  m->ranks_used = 0;
  m->bounds.clear();
  if (n_feat < 0 || n_samp < 0 || ld < n_feat) return mfail(m, ICIKT_E_INVALID, "pairs_multi: bad matrix shape");
  if (pi == nullptr) {
    if (pj != nullptr) return mfail(m, ICIKT_E_INVALID, "pairs_multi: pi is null but pj is not");
    n_pairs = n_samp * (n_samp - 1) / 2;
  }
  const int G = m->n;
  // Too little work to split (fewer than two columns per rank, or a handful of pairs): rank 0's single-device
  // path, which also owns every argument check.  A handle of ONE device takes it too -- it is the pipelined host path
  // (matrix chunks, pre-pass and pair kernel overlapped: icikt_pairs_f64), so the N = 1 point of an in-library scaling
  // curve IS the single-device figure (round 3: 14.4 against 12.2 ms on c4, the rank flow uploads, then sorts, then
  // counts) -- unless RCCL was asked for by name: then the whole rank flow runs on the one-rank communicator (tests).
  // Wide columns (n_feat > ICIKT_MAX_FEATURES) run on rank 0 alone as well: the ranks exchange the 16-bit prepared
  // state of the tuned kernels, which wide columns do not have.
  if (n_samp < 2 * (int64_t)G || n_pairs < 64 * (int64_t)G || n_feat == 0 || n_feat > ICIKT_MAX_FEATURES ||
      (G == 1 && !m->rccl_asked)) {
    icikt_ctx* c = m->ctx[0];
    const double t0 = now_ms();
    const int rc = mx ? icikt_matrix_f64(c, X, n_feat, n_samp, ld, mx->global_na, mx->n_global_na, pi, pj, n_pairs, perspective,
                                         alternative, continuity, flags, mx->scale_max, mx->diag_good, mx->out5, mx->keep,
                                         mx->reason_counts)
                      : icikt_pairs_f64(c, X, n_feat, n_samp, ld, pi, pj, n_pairs, perspective, alternative, continuity, flags,
                                        out4, counts, reasons);
    if (rc) return mfail(m, rc, icikt_last_error(c));
    m->ranks_used = 1;   // the caller can tell (icikt_multi_ranks_used); the whole call is booked as the pairs phase
    m->bounds.assign(2, 0);
    m->bounds[1] = n_pairs;
    m->phase_ms[ICIKT_MULTI_PHASE_PAIRS] = m->rank_ms[ICIKT_MULTI_PHASE_PAIRS] = now_ms() - t0;
    return ICIKT_SUCCESS;
  }
  // argument checks (the ranks run unchecked): same texts as the single-device path
  if (!X) return mfail(m, ICIKT_E_INVALID, "pairs_multi: null matrix");
  if (n_pairs < 0 || (pi && !pj)) return mfail(m, ICIKT_E_INVALID, "pairs_multi: bad pair list");
  if (pi)
    for (int64_t p = 0; p < n_pairs; ++p)
      if (pi[p] < 0 || pi[p] >= n_samp || pj[p] < 0 || pj[p] >= n_samp)
        return mfail(m, ICIKT_E_INVALID, "pairs_multi: column index out of range");
  if (mx ? !mx->out5 : !out4) return mfail(m, ICIKT_E_INVALID, "pairs_multi: null output");
  if (perspective != ICIKT_PERSPECTIVE_LOCAL && perspective != ICIKT_PERSPECTIVE_GLOBAL)
    return mfail(m, ICIKT_E_INVALID, "pairs_multi: perspective must be local (0) or global (1)");
  if (alternative < 0 || alternative > ICIKT_ALT_OTHER) return mfail(m, ICIKT_E_INVALID, "pairs_multi: bad alternative code");

  Call a{};
  a.m = m; a.X = X; a.n_feat = n_feat; a.n_samp = n_samp; a.ld = ld; a.pi = pi; a.pj = pj; a.P = n_pairs;
  a.n_each = (n_pairs + G - 1) / G;  // ceiling(n_todo / ncore), R/kendalltau.R:250
  a.balance = (flags & ICIKT_FLAG_BALANCE_COST) != 0;
  a.slot = a.balance ? std::min<int64_t>(n_pairs, 2 * a.n_each) : a.n_each;
  a.bounds.assign((size_t)G + 1, 0);
  for (int r = 0; r <= G; ++r) a.bounds[(size_t)r] = std::min<int64_t>(n_pairs, (int64_t)r * a.n_each);   // (cost-weighted: replaced in phase B)
  a.perspective = perspective; a.alternative = alternative; a.continuity = continuity; a.flags = flags;
  a.out4 = out4; a.counts = counts; a.reasons = reasons;
  a.cols_per = 2 * ((n_samp + 2 * G - 1) / (2 * G));  // even: the rec table interleaves column pairs
  a.alloc_cols = a.cols_per * G;
  a.timing = (flags & ICIKT_FLAG_TIMING) != 0;
  if (mx) {
    a.matrix = true;
    const int rcm = icikt::host::make_mask_spec(m->ctx[0], mx->global_na, mx->n_global_na, &a.mask);
    if (rcm) return mfail(m, rcm, icikt_last_error(m->ctx[0]));
    a.keep = mx->keep; a.out5 = mx->out5; a.scale_max = mx->scale_max ? 1 : 0; a.diag_good = mx->diag_good ? 1 : 0;
    if (mx->reason_counts) for (int k = 0; k < 5; ++k) mx->reason_counts[k] = 0;
  }
  Barrier bar(G);
  a.bar = &bar;
  a.rc.assign((size_t)G, ICIKT_SUCCESS);
  a.msg.assign((size_t)G, std::string());
  a.order_base.assign((size_t)G, nullptr); a.meta_base.assign((size_t)G, nullptr);
  a.out4_dev.assign((size_t)G, nullptr); a.counts_dev.assign((size_t)G, nullptr); a.reasons_dev.assign((size_t)G, nullptr);

  // Every rank stages its columns through its own pinned buffer (icikt_host.h: the library never page-locks the caller's
  // memory); with ICIKT_FLAG_HOST_PINNED the caller has page-locked the matrix and the result arrays (portably, for
  // several devices: hipHostMallocPortable / hipHostRegisterPortable) and every rank DMAs its columns straight out of it.
  for (icikt_ctx* c : m->ctx) c->host_pinned = (flags & ICIKT_FLAG_HOST_PINNED) != 0;

  std::vector<std::thread> th;
  bool started = true;
  try {
    for (int r = 1; r < G; ++r) th.emplace_back(rank_main, std::ref(a), r);
  } catch (...) {
    // a rank without a thread would leave the others at the first barrier: fail every rank (nobody enters a
    // collective then) and release the barrier for good
    started = false;
    for (int& v : a.rc) v = ICIKT_E_NOMEM;
    for (auto& t : a.msg) t = "could not start the rank threads";
    bar.abort();
  }
  if (started) rank_main(a, 0);  // the calling thread is rank 0
  for (auto& t : th) t.join();
  for (icikt_ctx* c : m->ctx) c->host_pinned = false;
  m->ranks_used = G;
  m->bounds = a.bounds;
  for (int r = 0; r < G; ++r)
    for (int k = 0; k < ICIKT_MULTI_PHASES; ++k)
      m->phase_ms[k] = std::max(m->phase_ms[k], m->rank_ms[(size_t)r * (ICIKT_MULTI_PHASES + 1) + (size_t)k]);
  for (int r = 0; r < G; ++r)
    if (a.rc[(size_t)r] != ICIKT_SUCCESS)
      return mfail(m, a.rc[(size_t)r], "rank " + std::to_string(r) + " (device " + std::to_string(m->devices[(size_t)r]) +
                                           "): " + a.msg[(size_t)r]);
  if (mx && mx->reason_counts) for (int k = 0; k < 5; ++k) mx->reason_counts[k] = (int64_t)a.red[1 + k];
  return ICIKT_SUCCESS;
}

}  // namespace

extern "C" {

int icikt_pairs_multi_f64(icikt_multi* m, const double* X, int64_t n_feat, int64_t n_samp, int64_t ld,
                          const int32_t* pi, const int32_t* pj, int64_t n_pairs, int perspective, int alternative,
                          int continuity, uint32_t flags, double* out4, int64_t* counts, int32_t* reasons) {
  return multi_impl(m, X, n_feat, n_samp, ld, pi, pj, n_pairs, perspective, alternative, continuity, flags, out4, counts,
                    reasons, nullptr);
}

int icikt_matrix_multi_f64(icikt_multi* m, const double* X, int64_t n_feat, int64_t n_samp, int64_t ld,
                           const double* global_na, int n_global_na, const int32_t* pi, const int32_t* pj, int64_t n_pairs,
                           int perspective, int alternative, int continuity, uint32_t flags, int scale_max, int diag_good,
                           double* out5, uint8_t* keep, int64_t* reason_counts) {
  const MatrixArgs mx{global_na, n_global_na, scale_max, diag_good, out5, keep, reason_counts};
  return multi_impl(m, X, n_feat, n_samp, ld, pi, pj, n_pairs, perspective, alternative, continuity, flags, nullptr, nullptr,
                    nullptr, &mx);
}

}  // extern "C"
