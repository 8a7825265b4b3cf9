// icikt_capi.cpp -- host side of the C ABI declared in include/icikt.h.
//
// Owns the HIP device/stream/workspaces of a context and sequences the three kernels of
// icikt_kernels.hip.  There is deliberately NO CPU implementation of the arithmetic here: when no
// HIP device is usable every entry point fails with ICIKT_E_NO_DEVICE / ICIKT_E_HIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <functional>
#include <new>
#include <atomic>
#include <mutex>
#include <condition_variable>
#include <thread>
#include <string>
#include <vector>

#include "icikt.h"
#include "icikt_device.h"
#include "icikt_host.h"

using icikt::ColStats;
using icikt::PairRaw;
using icikt::PrepView;

static_assert(ICIKT_CNT_FIELDS == icikt::ICIKT_CNT_FIELDS_, "counts record layout");
static_assert(ICIKT_PERSPECTIVE_LOCAL == icikt::ICIKT_PERSPECTIVE_LOCAL_, "perspective code");

namespace icikt {
namespace host {

int fail(icikt_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  return code;
}

int use_device(icikt_ctx* c) {
  HIPCHK(c, hipSetDevice(c->device));
  return ICIKT_SUCCESS;
}

}  // namespace host
}  // namespace icikt

using icikt::host::fail;
using icikt::host::use_device;

namespace {

// fold the recorded event pairs of kernel k into the accumulated time (waits for them to complete)
int flush_timer(icikt_ctx* c, int k) {
  for (size_t i = 0; i < c->ev_used[k]; ++i) {
    HIPCHK(c, hipEventSynchronize(c->ev_pool[k][i].b));
    float t = 0.f;
    HIPCHK(c, hipEventElapsedTime(&t, c->ev_pool[k][i].a, c->ev_pool[k][i].b));
    c->ms[k] += (double)t;
    c->launches[k] += 1;
  }
  c->ev_used[k] = 0;
  return ICIKT_SUCCESS;
}

int timer_begin(icikt_ctx* c, int k, uint32_t flags) {
  if (!(flags & ICIKT_FLAG_TIMING)) return ICIKT_SUCCESS;
  if (c->ev_used[k] == c->ev_pool[k].size()) {
    if (c->ev_pool[k].size() >= 256) {  // pool full: the one place a timed launch waits for earlier ones
      int rc = flush_timer(c, k);
      if (rc) return rc;
    } else {
      icikt_ctx::EvPair p;
      HIPCHK(c, hipEventCreate(&p.a));
      HIPCHK(c, hipEventCreate(&p.b));
      c->ev_pool[k].push_back(p);
    }
  }
  HIPCHK(c, hipEventRecord(c->ev_pool[k][c->ev_used[k]].a, c->stream));
  c->ev_open[k] = true;
  return ICIKT_SUCCESS;
}

int timer_end(icikt_ctx* c, int k, uint32_t flags) {
  if (!(flags & ICIKT_FLAG_TIMING) || !c->ev_open[k]) return ICIKT_SUCCESS;
  HIPCHK(c, hipEventRecord(c->ev_pool[k][c->ev_used[k]].b, c->stream));
  c->ev_used[k] += 1;
  c->ev_open[k] = false;
  return ICIKT_SUCCESS;
}

// launch plan of the pair kernel for a given n
struct K1Plan {
  int np;            // pairs per wave: 2 (one per half, consecutive pairs with the same pi) or 1
  int wpb;           // waves per workgroup
  size_t lds_bytes;
  int perpair_bytes;
  int opts;          // bit 0: half-wave hot step (one pair per 32-lane half, 32-row sub-steps)
  int half_items;    // words per lane of a half-wave prefix rebuild (0: no half-wave step)
  int stride;        // 64-bit words between a pair's LDS / pend arrays (k1_lds_stride)
  int split;         // half-wave kernels: segments a task is cut in (1 | 2 | 4) when the task list leaves the chip half empty
};

K1Plan plan_k1(const PrepView& pv, int64_t n_pairs, int n_cu, const icikt_ctx::PlanOverride& ov, bool tied, int ntg_hint = -1, int group_hint = 0) {
  K1Plan pl{};
  const size_t lds_cap = 160 * 1024;
  // n <= 18 336 (a half wave rebuilds a prefix with <= 9 words per lane): two pairs per wave, one per 32-lane half.
  // Longer columns: pairs on the whole wave -- TWO per wave, one after the other, while eight waves of two pairs fit a
  // CU's LDS (n <~ 60 000), else one.  The two share the streamed column and the rec block, so one 8-byte gather per row
  // serves both: the pair kernel of long columns was bound by the L2's request rate (64 requests per wave and gather; the
  // block never sits in an L1), not by instruction issue (round 3: a quarter fewer vector instructions changed nothing
  // at n = 50 000).  Measured (tools/n_sweep.py, 512 columns, one pair -> two pairs per wave -> their singleton region in
  // the half layout): n = 20 000 1.24e7 -> 1.69e7 -> 1.98e7 pairs/s, 30 000 8.0e6 -> 1.13e7 -> 1.33e7, 36 000 6.6e6 -> 9.4e6
  // -> 1.11e7, 50 000 4.9e6 -> 5.8e6 -> 7.2e6, 60 000 4.1e6 -> 4.9e6 -> 6.0e6, 65 535 3.7e6 -> 3.9e6 -> 4.6e6.
  // No kernel keeps a second bitset for open tie groups any more (`pend`, in LDS or in per-wave global slots with a
  // persistent grid: rounds 1-2): the plan keys pend / gridmult / gridcap of icikt_debug_set_plan are accepted and ignored.
  // 18 337 .. 30 656 rows: BOTH families fit.  Continuous columns run 25-50 % faster in the long-column kernel (n = 20 000:
  // 1.98e7 vs 1.55e7 pairs/s; 30 000: 1.33e7 vs 0.88e7: the half-wave prefix rebuild costs 11 .. 15 words per lane there,
  // the long-column kernel takes singleton rows in the half layout over the two-level counts: bit 2 of `opts` below),
  // tied columns 1.4-2.1x faster in the half-wave kernels (their MIXED steps use the packed chains and the pre-pass's
  // masks; n = 30 000, ~15 000 / 3 000 / 600 distinct values: 1.48 / 1.95 / 1.74 ms vs 2.46 / 4.12 / 2.83): `tied` (the
  // prepared columns average more than eight tie groups: matrix_tied below) chooses.
  const bool half_fits = icikt::k1_half_items(pv.Wp) <= icikt::ICIKT_HALF_ITEMS_MAX;
  // (15 200 .. 18 336 rows, the upper part of nine words per lane: the same choice, for less -- continuous columns 3-5 % faster in
  //  the long-column kernel, n = 16 000: 2.48e7 vs 2.38e7 pairs/s, 18 336: 2.07e7 vs 1.97e7; level at 14 400)
  const int hi_words = icikt::k1_half_items(pv.Wp);
  const bool both_fit = hi_words > 9 || (hi_words == 9 && pv.n >= 15200 && n_pairs > (int64_t)4 * n_cu);   // (a short task list: the half-wave kernels, cut in segments)
  const bool half_ok = half_fits && (!both_fit || (ov.half < 0 ? tied : ov.half != 0));
  int np = half_ok ? 2 : 1;
  {
    const size_t two = 2 * ((size_t)icikt::k1_lds_stride(pv.Wp, 0) * 8 + icikt::K1_TL_BYTES);
    if (!half_ok && 7 * two <= lds_cap) np = 2;   // (65 535 rows: seven single-wave workgroups, 3.96e6 vs 3.74e6)
  }
  // few pairs: twice the waves hide the latency of a step better than two pairs per wave share their loads while
  // the chip is nearly empty (measured on the yeast matrix cut to 24 .. 96 columns: one pair per wave wins up to
  // 780 pairs, 0.209 vs 0.230 ms, two pairs per wave from 1 128 pairs on, 0.231 vs 0.242 ms)
  // (round 4, last change: that held for the tie steps of round 3.  With step records, count mode and SOLO steps the
  //  half-wave kernels win at every size on tied data -- yeast cut to 4 .. 32 columns: 0.13 against 0.22 ms -- and draw level
  //  on continuous data, 0.08 against 0.07 ms at 10 000 x 8 .. 32: only the whole-wave family still drops to one pair per wave)
  if (n_pairs <= (int64_t)4 * n_cu && !half_ok) np = 1;
  // columns too long for the half-wave kernels whose tie groups are many and short (matrix_tied): one pair per wave
  if (tied && !half_fits) np = 1;
  // overrides for experiments and tests (icikt_debug_set_plan; the product path reads no environment variable)
  if (ov.np == 1 || ov.np == 2) np = ov.np;
  int wpb = 4;
  if (!half_ok && np == 2) {   // fewer than two four-wave workgroups of two pairs fit: single-wave workgroups fill the LDS
    const size_t two = 2 * ((size_t)icikt::k1_lds_stride(pv.Wp, 0) * 8 + icikt::K1_TL_BYTES);
    if (8 * two > lds_cap) wpb = 1;
  }
  if (ov.wpb > 0) wpb = std::max(1, std::min(8, ov.wpb));
  pl.opts = 1;
  if (ov.half >= 0 && !both_fit) pl.opts = ov.half ? 1 : 0;
  int tg_max = 128;  // bits 8..: whole-wave kernels: joint ties of a closing group from the gathered column's tie-group list
                     // while it has at most this many groups (the kernels cap it at 128: two listed groups per lane), else row
                     // by row; half-wave kernels: the entries of a pair's counter table (count mode), sized below
  if (ov.has_tgmax) tg_max = std::max(-1, std::min(1 << 20, ov.tgmax));
  const bool row_only = tg_max < 0;
  if (row_only) { pl.opts |= 2; tg_max = 0; }  // bit 1: row mode only
  // bits 8..15: list mode (range counts per listed tie group at a group's close: a cost per CLOSE, right for few, long
  // groups -- and same-address atomics of count mode would serialise there) up to this many tie groups
  int tg_list = std::min(tg_max, 128);   // (raised to 256 below for the half-wave kernels of 11 .. 15 words per lane)
  // half-wave kernels: a half rebuilds a prefix with half_items words per lane, unpredicated, so the LDS arrays of such
  // a kernel are padded to 32 * half_items words; every other plan runs pairs on the whole wave
  pl.half_items = icikt::k1_half_items(pv.Wp);
  if (np != 2 || !half_ok || !(pl.opts & 1)) {
    pl.opts &= ~1;
    pl.half_items = 0;
  }
  // bit 2: two long-column pairs of a whole-wave kernel take the singleton region in the half layout (one pair per
  // 32-lane half, 32-row sub-steps, the half-wave in-step chain)
  if (np == 2 && pl.half_items == 0 && ov.hyb != 0) pl.opts |= 4;
  // final layout: the kernel derives the same stride from (Wp, half_items)
  pl.stride = icikt::k1_lds_stride(pv.Wp, pl.half_items);
  if (pl.half_items > 0) {
    // seen + prefix slots + the counters of count mode: one u16 per tie group of a gathered column (+ one for the rows
    // that are their own group), as many as the LDS holds WITHOUT costing the launch a wave it would otherwise run: six
    // waves per SIMD (five from 11 words per lane on) or, for a short task list, the waves the list fills
    const size_t base = (size_t)pl.stride * 8 + icikt::k1_half_pre(pl.half_items);
    auto pair_bytes = [&](int cap) { return (base + ICIKT_CNT_BYTES * ((size_t)cap + 2) + 15) & ~(size_t)15; };
    auto waves_fit = [&](int cap) { return (int)(lds_cap / ((size_t)wpb * np * pair_bytes(cap))) * wpb; };
    const int waves_max = 4 * (pl.half_items > 9 ? 5 : 6);
    const int64_t tasks = (n_pairs + 1) / 2;
    // a task list that leaves the chip half empty: every task is cut in 2 or 4 segments, a wave each (k1_pairs) -- such a
    // launch lasts as long as ONE task otherwise; the counter tables are then sized for the waves of the segments
    // (measured on yeast columns and 10 000-row synthetic ones, 4 .. 96 columns: four segments pay while the segments' waves
    //  number at most the chip's SIMDs, two up to 2.5 tasks per CU; beyond, the launch is bound by the SIMDs that hold three waves,
    //  not by a wave's latency, and the rows inserted twice only add work -- the full yeast matrix: 0.208 -> 0.237 ms)
    pl.split = 1;
    if (pv.n >= 2048) pl.split = (tasks <= (int64_t)n_cu) ? 4 : (2 * tasks <= (int64_t)5 * n_cu) ? 2 : 1;
    if (ov.split == 1 || ov.split == 2 || ov.split == 4) pl.split = ov.split;
    const int waves_needed = (int)std::min<int64_t>(waves_max, std::max<int64_t>(wpb, (tasks * pl.split + n_cu - 1) / std::max(1, n_cu)));
    int waves_target = std::min(waves_needed, std::max(wpb, waves_fit(0)));
    if (ov.waves > 0) waves_target = std::min(waves_target, std::max(wpb, ov.waves));
    const int tg_list_max = row_only ? 0 : (pl.half_items > 9 ? 256 : 128);   // (list mode's reach in this kernel, below)
    int cap = 0;
    if (!row_only) {
      static const int caps[] = {4094, 3072, 2048, 1536, 1024, 768, 512, 384, 256, 192, 128, 64};
      for (int cc : caps) {
        if (cc > pv.n / 2 + 64 && cc > 64) continue;              // (a column of n rows has at most n / 2 tie groups)
        if (waves_fit(cc) >= waves_target) { cap = cc; break; }
      }
      // Columns of 14 273 rows and more leave the tables next to nothing at that occupancy (15 words per lane: 64 counters
      // beside 4.9 KB of state per pair).  When the prepared columns' statistics (matrix_tied: read back once per matrix at
      // these lengths) say that their tie groups would fit a table at three waves per SIMD, the table is worth the waves:
      // count data -- negative binomial, ~700 tie groups per column, nine rows in ten in groups of more than 32 -- 20 000 x
      // 256: 4.2 -> 3.2 ms, 30 000 x 256: 6.8 -> 4.7 (row mode streams a long group three times).
      // (... when list mode cannot serve them and a tied pair of rows sits, on average, in a group of 256 rows or more -- from the
      //  columns' tie sums, fill group apart: 30 000-row columns of 600 tie groups of ~100 rows run 1.61 ms in row mode at four waves
      //  per SIMD, 1.98 in count mode at three)
      if (ntg_hint > std::max(cap, tg_list_max) && group_hint >= 256 && ov.waves <= 0 && pl.half_items >= 9) {
        for (int i = (int)(sizeof(caps) / sizeof(caps[0])) - 1; i >= 0; --i) {
          if (caps[i] < ntg_hint) continue;
          if (waves_fit(caps[i]) >= 12) cap = caps[i];
          break;
        }
      }
      if (ov.has_tgmax) cap = std::min(cap, tg_max);
    }
    tg_max = cap;
    pl.perpair_bytes = (int)pair_bytes(cap);
  } else {
    // seen + the two-level counts + the counters of count mode (round 4: the whole-wave kernels too): as many as the LDS holds
    // beside the waves the launch runs anyway -- 50 000 rows: 512 counters per pair at eight waves of two pairs (or sixteen of one)
    const size_t base = (size_t)pl.stride * 8 + icikt::K1_TL_BYTES;
    auto pair_bytes = [&](int cap) { return cap > 0 ? ((base + ICIKT_CNT_BYTES * ((size_t)cap + 2) + 15) & ~(size_t)15) : base; };
    auto waves_of = [&](size_t pb, int w0) {   // waves a CU holds with pb bytes per pair in workgroups of up to w0 waves
      const int w = std::max(1, std::min(w0, (int)(lds_cap / (pb * np))));
      return (int)(lds_cap / ((size_t)w * np * pb)) * w;
    };
    auto waves_any = [&](size_t pb) { return std::max(waves_of(pb, wpb), waves_of(pb, 1)); };   // (single-wave workgroups pack the LDS best)
    int cap = 0;
    if (!row_only && !ov.has_tgmax) {
      static const int caps[] = {4094, 3072, 2048, 1536, 1024, 768, 512, 384, 256};
      const int target = std::min(waves_of(base, wpb), np == 2 ? 12 : 24);   // (the registers' limit: 3 / 6 waves per SIMD)
      for (int cc : caps) {
        if (cc > pv.n / 2 + 64) continue;                     // (a column of n rows has at most n / 2 tie groups)
        if (waves_any(pair_bytes(cc)) >= target) { cap = cc; break; }
      }
      // the prepared columns hold more tie groups than that (matrix_tied's read-back): a table that covers them is worth one wave
      // of eight -- count-like data, 50 000 x 96, ~630 tie groups per column (at most 940): 2.76 ms with 512 counters at eight
      // waves per CU, 2.48 with 1 024 at seven
      // (... when a tied pair of rows sits, on average, in a group of 256 rows or more: groups of one or two steps close from
      //  registers in row mode, for less than the counters and the lost wave cost -- 50 000 x 128 columns of ~1 000 tie groups of
      //  ~120 rows: 5.24 ms in row mode, 5.77 with the table)
      if (ntg_hint > cap && group_hint >= 256 && target >= 6) {
        for (int i = (int)(sizeof(caps) / sizeof(caps[0])) - 1; i >= 0; --i) {
          if (caps[i] < ntg_hint) continue;
          if (waves_any(pair_bytes(caps[i])) >= target - 1) cap = caps[i];
          break;
        }
      }
      if (cap > 0 && ov.wpb <= 0 && waves_of(pair_bytes(cap), 1) > waves_of(pair_bytes(cap), wpb)) wpb = 1;
    } else if (!row_only && ov.tgmax > 128) {
      cap = std::min(ov.tgmax, 4094);   // (tests: a table whatever it costs in waves)
      while (cap > 0 && pair_bytes(cap) * np > lds_cap) cap /= 2;
    }
    pl.perpair_bytes = (int)pair_bytes(cap);
    pl.opts |= cap << 18;   // bits 18..: entries of a pair's counter table (count mode); 0: none
  }
  // (eight listed groups per lane in the kernels that have the registers: list mode up to 256 tie groups there -- count
  //  mode has 64 counters beside their 4.9 KB of LDS state per pair, and row mode streams a long group three times)
  if (pl.half_items > 9 && !row_only) tg_list = std::min(ov.has_tgmax ? std::max(0, ov.tgmax) : 256, 256);
  if (ov.list >= 0) tg_list = std::min(tg_list, ov.list);
  pl.opts |= tg_list << 8;          // bits 8..17
  if (ov.solo == 0) pl.opts |= 8;   // bit 3: no SOLO steps
  if (pl.half_items > 0) pl.opts |= tg_max << 18;   // bits 18..: entries of a pair's counter table (count mode)
  const int fit = std::max(1, (int)(lds_cap / ((size_t)pl.perpair_bytes * np)));
  pl.np = np;
  pl.wpb = std::min(wpb, fit);
  pl.lds_bytes = (size_t)pl.wpb * np * pl.perpair_bytes;
  return pl;
}

// Do the prepared columns hold tie groups beyond a fill group or so?  Only asked where the answer chooses the kernel:
// 15 200 .. 30 656 rows (the kernel family), and longer columns (pairs per wave: columns of many SHORT tie groups -- on
// average fewer than 32 rows per group -- run their tie steps one pair after the other whatever shares the wave, and one
// pair per wave then has twice the waves to hide a step's latency: 50 000 x 512 columns of ~10-row groups 123 -> 88 ms;
// columns of long groups share their gathers and keep two pairs per wave, 75 against 113 ms).  The statistics of up to 64
// columns are read back once per prepared matrix -- after `ready` (an event behind their pre-pass; nullptr: the context's
// stream) -- and the verdict is kept.
bool matrix_tied(icikt_ctx* c, int64_t ncols_ready, hipEvent_t ready) {
  const PrepView& pv = c->pv;
  const int hi = icikt::k1_half_items(pv.Wp);
  const bool long_cols = hi > icikt::ICIKT_HALF_ITEMS_MAX;
  // (9 words per lane: the family is the half-wave one anyway; the read-back serves the size of the counter tables, plan_k1)
  if (pv.wide || hi < 9 || pv.n <= 0 || (!long_cols && c->plan_ov.half >= 0) || (long_cols && c->plan_ov.np > 0)) return false;
  if (c->tied_state >= 0) return c->tied_state != 0;
  const int64_t m = std::min<int64_t>(std::min<int64_t>(ncols_ready, pv.n_samp), 64);
  if (m <= 0) return false;
  const hipError_t es = ready ? hipEventSynchronize(ready) : hipStreamSynchronize(c->stream);
  if (es != hipSuccess) { (void)hipGetLastError(); return false; }
  std::vector<icikt::ColStats> st((size_t)m);
  if (hipMemcpy2D(st.data(), sizeof(icikt::ColStats), pv.col_stats(0), (size_t)pv.mstride * sizeof(unsigned long long),
                  sizeof(icikt::ColStats), (size_t)m, hipMemcpyDeviceToHost) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  unsigned long long groups = 0, all_groups = 0, rows = 0;
  c->ntg_hint = 0;
  c->group_hint = 0;
  double g0 = 0.0, g1 = 0.0;
  for (const auto& t : st) {
    groups += t.ntg;
    c->ntg_hint = std::max(c->ntg_hint, (int)std::min<uint32_t>(t.ntg, 1u << 20));
    // the size of the tie group a random tied PAIR of rows sits in, fill group apart: sum t (t-1) (t-2) / sum t (t-1) + 2
    {
      const long long f = t.tfill;
      g0 += (double)(t.e0 - f * (f - 1));
      g1 += (double)(t.e1 - f * (f - 1) * (f - 2));
    }
    all_groups += (unsigned long long)std::max(t.ngroups, 1);
    rows += (unsigned long long)std::max(0, pv.n - t.nna);
  }
  c->group_hint = (g0 > 0.0) ? (int)std::min(1.0e6, g1 / g0 + 2.0) : 0;
  const bool tied = groups > 8ull * (unsigned long long)m;
  c->tied_state = (tied && (!long_cols || rows < 32ull * all_groups)) ? 1 : 0;
  if (c->plan_ov.verbose) fprintf(stderr, "[icikt] %lld columns read back: %.1f tie groups per column (at most %d), %.1f rows per group, a tied pair's group %d rows -> %s\n", (long long)m,
                                  (double)groups / (double)m, c->ntg_hint, (double)rows / (double)all_groups, c->group_hint,
                                  long_cols ? (c->tied_state ? "one pair per wave" : "two pairs per wave") : (c->tied_state ? "half-wave kernels" : "whole-wave kernels"));
  return c->tied_state != 0;
}

// Tasks (one wave each), as (pair index, pair index or -1).  np == 1: one pair per task.  np == 2: two pairs
// that share their streamed column pj and whose gathered columns pi are the two columns (2a, 2a+1) of one
// block of the interleaved rec table, so that ONE 8-byte gather per row serves both pairs; pairs without such
// a partner run alone.  Tasks are ordered by gathered block, then streamed column (cache locality of the
// block's rec table).
// the host copies of a combn range (the device arrays were filled by a kernel)
void ensure_host_pairs(icikt_ctx* c) {
  if (c->h_pairs_valid) return;
  const int64_t n_samp = c->combn_S, begin = c->combn_begin, end = c->combn_end;
  c->h_pi.resize((size_t)(end - begin));
  c->h_pj.resize((size_t)(end - begin));
  // walk combn order: row i holds pairs (i, i+1..S-1), starting at offset i*S - i(i+1)/2
  int64_t i = 0, row_start = 0;
  while (i < n_samp - 1 && row_start + (n_samp - 1 - i) <= begin) {
    row_start += n_samp - 1 - i;
    ++i;
  }
  int64_t j = i + 1 + (begin - row_start);
  for (int64_t p = begin; p < end; ++p) {
    c->h_pi[(size_t)(p - begin)] = (int32_t)i;
    c->h_pj[(size_t)(p - begin)] = (int32_t)j;
    if (++j >= n_samp) {
      ++i;
      j = i + 1;
    }
  }
  c->h_pairs_valid = true;
}

void build_units(icikt_ctx* c, int np) {
  ensure_host_pairs(c);
  auto& u = c->h_units;
  u.clear();
  const int64_t P = c->n_pairs;
  const int32_t* pi = c->h_pi.data();
  const int32_t* pj = c->h_pj.data();
  u.reserve((size_t)P * 2);
  auto push = [&u](int64_t a, int64_t b) { u.push_back((int32_t)a); u.push_back((int32_t)b); };
  if (np == 1) {
    for (int64_t p = 0; p < P; ++p) push(p, -1);
  } else {
    // combn ranges and setup_comparisons lists are sorted by (pi, pj): merge the rows 2a and 2a+1 on pj
    bool sorted = true;
    for (int64_t p = 1; p < P && sorted; ++p)
      sorted = (pi[p] > pi[p - 1]) || (pi[p] == pi[p - 1] && pj[p] > pj[p - 1]);
    if (sorted) {
      int64_t a = 0;
      while (a < P) {
        int64_t ae = a;
        while (ae < P && pi[ae] == pi[a]) ++ae;
        if ((pi[a] & 1) == 0 && ae < P && pi[ae] == pi[a] + 1) {
          int64_t be = ae;
          while (be < P && pi[be] == pi[ae]) ++be;
          int64_t x = a, y = ae;
          while (x < ae || y < be) {
            if (x < ae && y < be && pj[x] == pj[y]) { push(x, y); ++x; ++y; }
            else if (y >= be || (x < ae && pj[x] < pj[y])) { push(x, -1); ++x; }
            else { push(y, -1); ++y; }
          }
          a = be;
        } else {
          for (int64_t x = a; x < ae; ++x) push(x, -1);
          a = ae;
        }
      }
    } else {
      // any other list: order the pairs by (block of pi, pj, parity of pi) and pair up neighbours
      std::vector<int64_t> idx((size_t)P);
      for (int64_t p = 0; p < P; ++p) idx[(size_t)p] = p;
      auto key = [pi, pj](int64_t p) {
        return ((uint64_t)(uint32_t)(pi[p] >> 1) << 33) | ((uint64_t)(uint32_t)pj[p] << 1) | (uint64_t)(pi[p] & 1);
      };
      std::sort(idx.begin(), idx.end(), [&key](int64_t x, int64_t y) {
        const uint64_t kx = key(x), ky = key(y);
        return kx < ky || (kx == ky && x < y);
      });
      int64_t i = 0;
      while (i < P) {
        const int64_t x = idx[(size_t)i];
        if (i + 1 < P) {
          const int64_t y = idx[(size_t)i + 1];
          if ((pi[x] & 1) == 0 && pi[y] == pi[x] + 1 && pj[y] == pj[x]) { push(x, y); i += 2; continue; }
        }
        push(x, -1);
        ++i;
      }
    }
  }
  c->n_units = (int)(u.size() / 2);
  c->wpb = np;
}

int upload_pairs(icikt_ctx* c) {
  const int64_t P = c->n_pairs;
  HIPCHK(c, c->d_pi.reserve((size_t)std::max<int64_t>(P, 1)));
  HIPCHK(c, c->d_pj.reserve((size_t)std::max<int64_t>(P, 1)));
  HIPCHK(c, c->d_raw.reserve((size_t)std::max<int64_t>(P, 1)));
  if (P > 0) {
    int rc = icikt::host::upload_sync(c, c->d_pi.p, c->h_pi.data(), (size_t)P * sizeof(int32_t));
    if (!rc) rc = icikt::host::upload_sync(c, c->d_pj.p, c->h_pj.data(), (size_t)P * sizeof(int32_t));
    if (rc) return rc;
  }
  return ICIKT_SUCCESS;
}

int upload_units(icikt_ctx* c) {
  HIPCHK(c, c->d_unit_start.reserve(c->h_units.size()));
  return icikt::host::upload_sync(c, c->d_unit_start.p, c->h_units.data(), c->h_units.size() * sizeof(int32_t));
}

// One launch of the pair kernel over tasks [first, first + count) of the uploaded task list, on c->stream.
int launch_pair_tasks(icikt_ctx* c, const K1Plan& pl, int first, int count) {
  if (count <= 0) return ICIKT_SUCCESS;
  // Half-wave kernels: every wave takes one task (grid = all tasks; workgroups start in order, and the kernel's XCD-aware
  // block mapping keeps the tasks of one gathered block on one XCD's L2): measured 9 % faster on c4 than persistent
  // waves, which run in lockstep and end on a ragged last round.  Whole-wave kernels (long columns, few and long
  // tasks per CU): when the task list is longer than what the chip holds, the grid is what the chip holds and the waves
  // FETCH their tasks, in order, from one counter per XCD group (k1_pairs: why the order matters for the L2) -- a wave
  // that has finished goes on at once instead of waiting for the other waves of its workgroup to retire (c5: +2.4 %).
  // Half-wave kernels, a task list that leaves the chip half empty: every task is cut in 2 or 4 segments (k1_pairs), a wave
  // each -- such a launch lasts as long as ONE task otherwise.  The segments add their counts up: the pairs' records are
  // cleared first.
  int split = 1, opts = pl.opts;
  if (pl.half_items > 0 && pl.np == 2 && pl.split > 1) {
    split = pl.split;
    opts |= (split == 2 ? 1 : 2) << 4;
    HIPCHK(c, icikt::launch_zero_raw(c->d_unit_start.p + 2 * (size_t)first, count, c->d_raw.p, c->stream));
  }
  const int want = (int)(((int64_t)count * split + pl.wpb - 1) / pl.wpb);
  int blocks = std::max(1, want);
  int per_cu = 0;
  bool persistent = false;
  if (pl.half_items == 0) {
    HIPCHK(c, icikt::k1_blocks_per_cu(pl.np, pl.half_items, pl.wpb, pl.lds_bytes, &per_cu));
    if (per_cu < 1) per_cu = 1;
    const int64_t resident = (int64_t)per_cu * c->prop.multiProcessorCount;
    const int64_t mult = c->plan_ov.grid_mult > 0 ? c->plan_ov.grid_mult : 1;
    int64_t cap = mult * resident;
    // test hook: a grid far smaller than the task list makes every persistent wave run many tasks in a row (the per-XCD
    // task counters, the in-order fetch, the re-initialisation of a wave's LDS state) whatever the chip would hold
    if (c->plan_ov.grid_cap > 0) cap = std::min<int64_t>(cap, c->plan_ov.grid_cap);
    if (want > cap) {
      blocks = (int)std::max<int64_t>(cap, 1);
      persistent = true;
      HIPCHK(c, c->d_task_ctr.reserve(8));
      HIPCHK(c, hipMemsetAsync(c->d_task_ctr.p, 0, 8 * sizeof(int), c->stream));
    }
  }
  if (c->plan_ov.verbose)
    fprintf(stderr, "[icikt] K1 plan: np=%d half_items=%d wpb=%d lds=%zu B/block (%d B/pair, %d tie-group counters), %d blocks/CU x %d CUs, "
            "grid=%d%s, tasks=%d (from %d), %d segment(s) per task\n",
            pl.np, pl.half_items, pl.wpb, pl.lds_bytes, pl.perpair_bytes, pl.opts >> 18, per_cu,
            c->prop.multiProcessorCount, blocks, persistent ? " (persistent)" : "", count, first, split);
  HIPCHK(c, icikt::launch_k1(c->pv, c->d_unit_start.p + 2 * (size_t)first, count, c->d_pi.p, c->d_pj.p, c->d_raw.p, pl.np,
                             pl.half_items, pl.wpb, blocks, pl.lds_bytes, pl.perpair_bytes,
                             persistent ? c->d_task_ctr.p : nullptr, opts, c->stream));
  return ICIKT_SUCCESS;
}

}  // namespace

extern "C" {

int icikt_version(void) { return ICIKT_VERSION; }

int icikt_device_count(int* count) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) n = 0;
  if (count) *count = n;
  return n > 0 ? ICIKT_SUCCESS : ICIKT_E_NO_DEVICE;
}

int icikt_ctx_create(int device, icikt_ctx** out) {
  if (!out) return ICIKT_E_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return ICIKT_E_NO_DEVICE;
  if (device < 0 || device >= n) return ICIKT_E_INVALID;
  icikt_ctx* c = new (std::nothrow) icikt_ctx();
  if (!c) return ICIKT_E_NOMEM;
  c->device = device;
  if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&c->prop, device) != hipSuccess ||
      hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return ICIKT_E_HIP;
  }
  c->stream = c->own_stream;
  if (hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) {
    icikt_ctx_destroy(c);
    return ICIKT_E_HIP;
  }
  {
    // the pre-pass of a chunk must not queue behind the pair kernel of the chunks before it for longer than that
    // kernel's workgroups take to retire: the highest priority the device offers
    int lo_prio = 0, hi_prio = 0;
    if (hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio) != hipSuccess) { (void)hipGetLastError(); hi_prio = 0; }
    if (hipStreamCreateWithPriority(&c->prep_stream, hipStreamNonBlocking, hi_prio) != hipSuccess) {
      icikt_ctx_destroy(c);
      return ICIKT_E_HIP;
    }
  }
  for (auto& e : c->ev_copy)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      icikt_ctx_destroy(c);
      return ICIKT_E_HIP;
    }
  *out = c;
  return ICIKT_SUCCESS;
}

void icikt_ctx_destroy(icikt_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
  c->order.release(); c->hirow.release(); c->girow.release(); c->rec.release(); c->meta.release();
  c->tgroups.release(); c->tprog.release(); c->smask.release(); c->srow.release(); c->sort_keys.release(); c->sort_idx.release(); c->wide32.release(); c->order_w.release(); c->k0_bits.release();
  c->d_pi.release(); c->d_pj.release(); c->d_unit_start.release(); c->d_raw.release();
  c->d_task_ctr.release();
  c->d_X.release(); c->d_Xp.release(); c->d_out4.release(); c->d_counts.release(); c->d_reasons.release(); c->d_self.release();
  c->d_out5.release(); c->d_keep.release(); c->d_red.release(); c->d_pi_all.release(); c->d_pj_all.release();
  for (int k = 0; k < ICIKT_K_COUNT; ++k)
    for (auto& p : c->ev_pool[k]) {
      if (p.a) (void)hipEventDestroy(p.a);
      if (p.b) (void)hipEventDestroy(p.b);
    }
  for (auto& e : c->ev_copy)
    if (e) (void)hipEventDestroy(e);
  if (c->pinned) (void)hipHostFree(c->pinned);
  if (c->pinned_tasks) (void)hipHostFree(c->pinned_tasks);
  for (auto& ps : c->out_pinned)
    if (ps.p) (void)hipHostFree(ps.p);
  if (c->copy_pool) { icikt::host::destroy_copy_pool(c->copy_pool); c->copy_pool = nullptr; }
  for (auto& e : c->ev_chunk)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : c->ev_out)
    if (e) (void)hipEventDestroy(e);
  if (c->prep_stream) { (void)hipStreamSynchronize(c->prep_stream); (void)hipStreamDestroy(c->prep_stream); }
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

const char* icikt_last_error(const icikt_ctx* c) { return c ? c->err.c_str() : "null context"; }

int icikt_ctx_set_stream(icikt_ctx* c, void* hip_stream) {
  if (!c) return ICIKT_E_INVALID;
  int rc = use_device(c);
  if (rc) return rc;
  (void)hipStreamSynchronize(c->stream);
  c->stream = reinterpret_cast<hipStream_t>(hip_stream);  // NULL is HIP's default (null) stream
  return ICIKT_SUCCESS;
}

int icikt_ctx_use_own_stream(icikt_ctx* c) {
  if (!c) return ICIKT_E_INVALID;
  int rc = use_device(c);
  if (rc) return rc;
  (void)hipStreamSynchronize(c->stream);
  c->stream = c->own_stream;
  return ICIKT_SUCCESS;
}

int icikt_sync(icikt_ctx* c) {
  if (!c) return ICIKT_E_INVALID;
  int rc = use_device(c);
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ICIKT_SUCCESS;
}

}  // extern "C"

// Shared body of icikt_prepare_dev / icikt_prepare_cols_dev: allocate the prepared state for alloc_cols
// columns and run the pre-pass over columns [col_begin, col_end).
static const char* const kTooLong =
    "n_feat exceeds ICIKT_MAX_FEATURES_WIDE (262144 rows per column; the tuned kernels take up to 65535, the plain "
    "32-bit path up to 262144 -- its bitsets must fit the LDS of a CU)";
static const char* const kNoWide =
    "n_feat exceeds ICIKT_MAX_FEATURES (65535 rows per column): this entry has no path for wide columns";

static int check_shape(icikt_ctx* c, const char* who, int64_t n_feat, int64_t n_samp, int64_t ld, bool wide_ok = true) {
  if (n_feat < 0 || n_samp < 0 || ld < n_feat) return fail(c, ICIKT_E_INVALID, std::string(who) + ": bad matrix shape");
  if (n_feat > ICIKT_MAX_FEATURES_WIDE) return fail(c, ICIKT_E_TOO_LONG, std::string(who) + ": " + kTooLong);
  if (n_feat > ICIKT_MAX_FEATURES && !wide_ok) return fail(c, ICIKT_E_TOO_LONG, std::string(who) + ": " + kNoWide);
  return ICIKT_SUCCESS;
}

static int check_col_range(icikt_ctx* c, int64_t n_samp, int64_t col_begin, int64_t col_end, int64_t alloc_cols,
                           int64_t n_feat = 0) {
  if (n_feat > ICIKT_MAX_FEATURES && (col_begin != 0 || col_end != n_samp))   // the sharded pre-pass exchanges 16-bit arrays
    return fail(c, ICIKT_E_TOO_LONG, std::string("prepare (column range): ") + kNoWide);
  if (col_begin < 0 || col_end < col_begin || col_end > n_samp || alloc_cols < n_samp)
    return fail(c, ICIKT_E_INVALID, "prepare: bad column range");
  // rec is interleaved in blocks of two columns: a column range is contiguous memory only on even boundaries
  // (an EMPTY range is fine wherever it sits: a rank beyond the last column of an odd-width matrix has [n_samp, n_samp))
  if (col_begin < col_end && ((col_begin & 1) || ((col_end & 1) && col_end != n_samp)))
    return fail(c, ICIKT_E_INVALID, "prepare: a column range must start on an even column and end on one (or at n_samp)");
  return ICIKT_SUCCESS;
}

namespace icikt {
namespace host {

int prepare_alloc(icikt_ctx* c, int64_t n_feat, int64_t n_samp, int64_t alloc_cols, int64_t sort_cols) {
  c->prepared = false;
  c->raw_valid = false;
  PrepView pv{};
  pv.n = (int)n_feat;
  pv.n_pad = (int)((n_feat + 63) / 64 * 64);
  if (pv.n_pad == 0) pv.n_pad = 64;
  pv.n_ord = pv.n_pad + 256;  // K1 loads rows up to three steps ahead: indices up to n + 191, zero padded
  pv.rec_rows = pv.n_pad + 8;  // row n_pad: the guard row (PrepView::rec_rows)
  pv.W = (int)((n_feat + 63) / 64);
  pv.Wp = pv.W + 1;
  int np2 = 2;
  while (np2 < pv.n) np2 <<= 1;
  pv.npow2 = np2;
  pv.n_samp = (int)n_samp;
  const size_t S = (size_t)std::max<int64_t>(alloc_cols, 1);
  const size_t ncols = (size_t)std::max<int64_t>(sort_cols, 1);

  pv.wide = n_feat > ICIKT_MAX_FEATURES ? 1 : 0;
  pv.mstride = 3 * pv.Wp + (int)(sizeof(ColStats) / 8);
  HIPCHK(c, c->meta.reserve(S * (size_t)pv.mstride));
  pv.tg_stride = pv.n_pad / 2 + 1;
  // sort scratch: bounded to ~1 GiB
  size_t chunk = std::min<size_t>(ncols, std::max<size_t>(1, ((size_t)1 << 30) / ((size_t)np2 * 12)));
  if (pv.wide) {
    // 32-bit positions in four separate arrays; the 16-bit arrays of the tuned kernels are not needed
    HIPCHK(c, c->wide32.reserve(4 * S * (size_t)pv.n_pad));
    HIPCHK(c, c->k0_bits.reserve(chunk * 2 * (size_t)(pv.Wp + 1)));
    HIPCHK(c, c->order.reserve(1));
    HIPCHK(c, c->hirow.reserve(1));
    HIPCHK(c, c->girow.reserve(1));
    HIPCHK(c, c->rec.reserve(2));
    HIPCHK(c, c->tgroups.reserve(1));
    HIPCHK(c, c->tprog.reserve(1));
    HIPCHK(c, c->smask.reserve(1));
    HIPCHK(c, c->srow.reserve(1));
  } else {
    HIPCHK(c, c->order.reserve(S * pv.n_ord));
    HIPCHK(c, c->hirow.reserve(((S + 1) & ~(size_t)1) * pv.rec_rows));  // interleaved like rec: [S/2 blocks][rec_rows rows][2 columns]
    HIPCHK(c, c->girow.reserve(((S + 1) & ~(size_t)1) * pv.rec_rows));  // likewise
    HIPCHK(c, c->rec.reserve(((S + 1) & ~(size_t)1) * pv.rec_rows));  // [S/2 blocks][rec_rows rows][2 columns]
    HIPCHK(c, c->tgroups.reserve(S * (size_t)pv.tg_stride));
    // the tie program serves the half-wave kernels only (n <= 18 336); longer columns classify their steps in the pair kernel
    pv.tp_stride = (icikt::k1_half_items(pv.Wp) <= icikt::ICIKT_HALF_ITEMS_MAX) ? pv.n_pad + 2 : 0;
    HIPCHK(c, c->tprog.reserve(std::max<size_t>(1, S * (size_t)pv.tp_stride)));
    // a step's record (its rows in lane layout, a MIXED step's same-group masks); + 3 guard steps behind the last one
    pv.sr_steps = pv.tp_stride ? pv.n_pad / 17 + 8 : 0;
    HIPCHK(c, c->srow.reserve(std::max<size_t>(1, S * (size_t)pv.sr_steps * 64)));
    HIPCHK(c, c->smask.reserve(std::max<size_t>(1, S * (size_t)pv.sr_steps * 32)));
    if (pv.tp_stride == 0) HIPCHK(c, c->order_w.reserve(S * (size_t)pv.n_ord));   // (the whole-wave kernels' ring reload: PrepView::order_w)
  }
  HIPCHK(c, c->sort_keys.reserve(chunk * np2));
  HIPCHK(c, c->sort_idx.reserve(chunk * np2));
  c->sort_chunk = (int)chunk;

  pv.order = c->order.p; pv.hirow = c->hirow.p; pv.girow = c->girow.p; pv.rec = c->rec.p;
  pv.order_w = (!pv.wide && pv.tp_stride == 0) ? c->order_w.p : nullptr;
  pv.meta = c->meta.p;
  pv.sort_keys = c->sort_keys.p; pv.sort_idx = c->sort_idx.p;
  pv.tgroups = c->tgroups.p;
  pv.tprog = c->tprog.p;
  pv.srow = c->srow.p;
  pv.smask = c->smask.p;
  if (pv.wide) {
    pv.order32 = c->wide32.p;
    pv.q32 = pv.order32 + S * (size_t)pv.n_pad;
    pv.lo32 = pv.q32 + S * (size_t)pv.n_pad;
    pv.hi32 = pv.lo32 + S * (size_t)pv.n_pad;
    pv.k0_bits = c->k0_bits.p;
  }
  c->pv = pv;
  c->alloc_cols = (int64_t)S;
  return ICIKT_SUCCESS;
}

int mask_alloc(icikt_ctx* c, int64_t n_feat, int64_t n_samp) {
  c->prepared = false;
  c->raw_valid = false;
  PrepView pv{};
  pv.n = (int)n_feat;
  pv.n_pad = std::max(64, (int)((n_feat + 63) / 64 * 64));
  pv.W = (int)((n_feat + 63) / 64);
  pv.Wp = pv.W + 1;
  pv.n_samp = (int)n_samp;
  pv.mstride = pv.Wp;                      // the missing-row bitset is all a column's record holds here
  HIPCHK(c, c->meta.reserve((size_t)std::max<int64_t>(n_samp, 1) * (size_t)pv.mstride));
  pv.meta = c->meta.p;
  c->pv = pv;
  return ICIKT_SUCCESS;
}

int prepare_launch(icikt_ctx* c, const double* dX, int64_t ld, int64_t col_begin, int64_t col_end, hipStream_t stream) {
  const PrepView& pv = c->pv;
  if (!stream) stream = c->stream;
  c->raw_valid = false;
  c->tied_state = -1;
  c->ntg_hint = -1;
  c->group_hint = 0;
  if (col_end > col_begin)
    HIPCHK(c, hipMemsetAsync(c->meta.p + (size_t)col_begin * pv.mstride, 0,
                             (size_t)(col_end - col_begin) * pv.mstride * sizeof(unsigned long long), stream));
  if (pv.n > 0) {
    const int64_t chunk = std::max(1, c->sort_chunk);
    for (int64_t c0 = col_begin; c0 < col_end; c0 += chunk) {
      const int nc = (int)std::min<int64_t>(chunk, col_end - c0);
      const int small_shape = (c->k0_shape == 1 || (c->k0_shape < 0 && c->k0_small)) ? 1 : 0;
      HIPCHK(c, icikt::launch_k0(pv, dX, ld, (int)c0, nc, c->k0_mask, c->k0_keep, small_shape, stream));
    }
  }
  return ICIKT_SUCCESS;
}

}  // namespace host
}  // namespace icikt

using icikt::host::prepare_alloc;
using icikt::host::prepare_launch;

static int prepare_impl(icikt_ctx* c, const double* dX, int64_t n_feat, int64_t n_samp, int64_t ld,
                        int64_t col_begin, int64_t col_end, int64_t alloc_cols, uint32_t flags) {
  if (!c) return ICIKT_E_INVALID;
  int rc = check_shape(c, "prepare", n_feat, n_samp, ld);
  if (rc) return rc;
  if (n_samp > 0 && n_feat > 0 && !dX) return fail(c, ICIKT_E_INVALID, "prepare: null matrix");
  rc = check_col_range(c, n_samp, col_begin, col_end, alloc_cols, n_feat);
  if (rc) return rc;
  rc = use_device(c);
  if (rc) return rc;
  rc = prepare_alloc(c, n_feat, n_samp, alloc_cols, col_end - col_begin);
  if (rc) return rc;
  rc = timer_begin(c, ICIKT_K_PREPARE, flags);
  if (rc) return rc;
  rc = prepare_launch(c, dX, ld, col_begin, col_end);
  if (rc) return rc;
  rc = timer_end(c, ICIKT_K_PREPARE, flags);
  if (rc) return rc;
  c->prepared = true;
  return ICIKT_SUCCESS;
}

extern "C" {

int icikt_prepare_dev(icikt_ctx* c, const double* dX, int64_t n_feat, int64_t n_samp, int64_t ld, uint32_t flags) {
  return prepare_impl(c, dX, n_feat, n_samp, ld, 0, n_samp, n_samp, flags);
}

int icikt_prepare_cols_dev(icikt_ctx* c, const double* dX, int64_t n_feat, int64_t n_samp, int64_t ld,
                           int64_t col_begin, int64_t col_end, int64_t alloc_cols, uint32_t flags) {
  return prepare_impl(c, dX, n_feat, n_samp, ld, col_begin, col_end, alloc_cols, flags);
}

int icikt_prepare_cols_f64(icikt_ctx* c, const double* X, int64_t n_feat, int64_t n_samp, int64_t ld,
                           int64_t col_begin, int64_t col_end, int64_t alloc_cols, uint32_t flags) {
  if (!c) return ICIKT_E_INVALID;
  int rc = check_shape(c, "prepare", n_feat, n_samp, ld);
  if (rc) return rc;
  if (n_samp > 0 && n_feat > 0 && !X) return fail(c, ICIKT_E_INVALID, "prepare: null matrix");
  rc = check_col_range(c, n_samp, col_begin, col_end, alloc_cols, n_feat);
  if (rc) return rc;
  rc = use_device(c);
  if (rc) return rc;
  rc = prepare_alloc(c, n_feat, n_samp, alloc_cols, col_end - col_begin);
  if (rc) return rc;
  const icikt::host::PinnedScope pinned_scope(c, flags);
  rc = icikt::host::upload_and_prepare(c, X, n_feat, n_samp, ld, col_begin, col_end, flags);
  if (rc) return rc;
  c->prepared = true;
  return ICIKT_SUCCESS;
}

int icikt_expand_cols_dev(icikt_ctx* c, int64_t col_begin, int64_t col_end, uint32_t flags) {
  if (!c) return ICIKT_E_INVALID;
  if (!c->prepared) return fail(c, ICIKT_E_STATE, "expand_cols: nothing prepared");
  if (col_begin < 0 || col_end < col_begin || col_end > c->pv.n_samp)
    return fail(c, ICIKT_E_INVALID, "expand_cols: bad column range");
  if (c->pv.wide) return fail(c, ICIKT_E_TOO_LONG, std::string("expand_cols: ") + kNoWide);
  int rc = use_device(c);
  if (rc) return rc;
  c->raw_valid = false;
  rc = timer_begin(c, ICIKT_K_PREPARE, flags);
  if (rc) return rc;
  HIPCHK(c, icikt::launch_k0_expand(c->pv, (int)col_begin, (int)(col_end - col_begin), c->stream));
  return timer_end(c, ICIKT_K_PREPARE, flags);
}

int icikt_prep_arrays(icikt_ctx* c, void** ptrs, int64_t* bytes_per_col) {
  if (!c || !ptrs || !bytes_per_col) return ICIKT_E_INVALID;
  if (!c->prepared) return fail(c, ICIKT_E_STATE, "prep_arrays: nothing prepared");
  if (c->pv.wide) return fail(c, ICIKT_E_TOO_LONG, std::string("prep_arrays: ") + kNoWide);
  const PrepView& pv = c->pv;
  ptrs[0] = pv.order;    bytes_per_col[0] = (int64_t)pv.n_ord * 2;
  ptrs[1] = pv.rec;      bytes_per_col[1] = (int64_t)pv.rec_rows * 4;
  ptrs[2] = pv.hirow;    bytes_per_col[2] = (int64_t)pv.rec_rows * 2;
  ptrs[3] = pv.meta;     bytes_per_col[3] = (int64_t)pv.mstride * 8;
  ptrs[4] = pv.tgroups;  bytes_per_col[4] = (int64_t)pv.tg_stride * 4;
  return ICIKT_SUCCESS;
}

int icikt_set_pairs(icikt_ctx* c, const int32_t* pi, const int32_t* pj, int64_t n_pairs) {
  if (!c) return ICIKT_E_INVALID;
  if (n_pairs < 0 || (n_pairs > 0 && (!pi || !pj))) return fail(c, ICIKT_E_INVALID, "set_pairs: bad pair list");
  if (n_pairs >= ((int64_t)1 << 31) - 1) return fail(c, ICIKT_E_INVALID, "set_pairs: more than 2^31-2 pairs");
  int rc = use_device(c);
  if (rc) return rc;
  int32_t mx = -1;
  for (int64_t p = 0; p < n_pairs; ++p) {
    if (pi[p] < 0 || pj[p] < 0) return fail(c, ICIKT_E_INVALID, "set_pairs: negative column index");
    mx = std::max(mx, std::max(pi[p], pj[p]));
  }
  try {
    c->h_pi.assign(pi, pi + n_pairs);
    c->h_pj.assign(pj, pj + n_pairs);
  } catch (const std::bad_alloc&) {
    return fail(c, ICIKT_E_NOMEM, "set_pairs: host allocation failed");
  }
  c->n_pairs = n_pairs;
  c->pairs_nsamp = (int64_t)mx + 1;
  c->raw_valid = false;
  c->wpb = 0;  // units are (re)built at run time for the plan of the prepared matrix
  c->combn_S = -1;
  c->h_pairs_valid = true;
  return upload_pairs(c);
}

int icikt_set_pairs_combn(icikt_ctx* c, int64_t n_samp, int64_t begin, int64_t end) {
  if (!c) return ICIKT_E_INVALID;
  const int64_t total = n_samp * (n_samp - 1) / 2;
  if (n_samp < 0 || begin < 0 || end < begin || end > total)
    return fail(c, ICIKT_E_INVALID, "set_pairs_combn: range outside C(n_samp, 2)");
  if (end - begin >= ((int64_t)1 << 31) - 1) return fail(c, ICIKT_E_INVALID, "set_pairs_combn: too many pairs");
  int rc = use_device(c);
  if (rc) return rc;
  // the pair arrays of a combn range are arithmetic: a kernel fills the device copies (no host loop, no upload);
  // the host copies are made only if a host-built task list needs them (ensure_host_pairs)
  c->n_pairs = end - begin;
  c->pairs_nsamp = n_samp;
  c->raw_valid = false;
  c->wpb = 0;
  c->combn_S = n_samp; c->combn_begin = begin; c->combn_end = end;
  c->h_pairs_valid = false;
  const int64_t P = c->n_pairs;
  HIPCHK(c, c->d_pi.reserve((size_t)std::max<int64_t>(P, 1)));
  HIPCHK(c, c->d_pj.reserve((size_t)std::max<int64_t>(P, 1)));
  HIPCHK(c, c->d_raw.reserve((size_t)std::max<int64_t>(P, 1)));
  HIPCHK(c, icikt::launch_fill_combn(c->d_pi.p, c->d_pj.p, n_samp, begin, P, c->stream));
  return ICIKT_SUCCESS;
}

int64_t icikt_num_pairs(const icikt_ctx* c) { return c ? c->n_pairs : -1; }

int icikt_run_dev(icikt_ctx* c, int perspective, int alternative, int continuity, uint32_t flags,
                  double* d_out4, int64_t* d_counts, int32_t* d_reasons) {
  if (!c) return ICIKT_E_INVALID;
  if (!c->prepared) return fail(c, ICIKT_E_STATE, "run: icikt_prepare_dev has not been called");
  if (c->n_pairs < 0) return fail(c, ICIKT_E_STATE, "run: no pair list set");
  if (perspective != ICIKT_PERSPECTIVE_LOCAL && perspective != ICIKT_PERSPECTIVE_GLOBAL)
    return fail(c, ICIKT_E_INVALID, "run: perspective must be local (0) or global (1)");
  if (alternative < 0 || alternative > ICIKT_ALT_OTHER) return fail(c, ICIKT_E_INVALID, "run: bad alternative code");
  if (c->pairs_nsamp > c->pv.n_samp) return fail(c, ICIKT_E_INVALID, "run: pair list refers to a column >= n_samp");
  if (c->n_pairs > 0 && !d_out4) return fail(c, ICIKT_E_INVALID, "run: null output");
  int rc = use_device(c);
  if (rc) return rc;
  if (c->n_pairs == 0) return ICIKT_SUCCESS;

  if (c->pv.wide) {
    // wide columns: the plain 32-bit pair kernel (one wave per pair, persistent single-wave workgroups that fetch
    // pairs from a counter), exact integer arithmetic in the epilogue
    const bool reuse_w = (flags & ICIKT_FLAG_REUSE_COUNTS) && c->raw_valid;
    if (c->pv.n > 0 && !reuse_w) {
      rc = timer_begin(c, ICIKT_K_PAIRS, flags);
      if (rc) return rc;
      const size_t lds = icikt::k1_wide_lds_bytes(c->pv.Wp);
      int per_cu = 0;
      HIPCHK(c, icikt::k1_wide_blocks_per_cu(lds, &per_cu));
      if (per_cu < 1) return fail(c, ICIKT_E_HIP, "run: the wide pair kernel does not fit a CU");
      const int blocks = (int)std::min<int64_t>(c->n_pairs, (int64_t)per_cu * c->prop.multiProcessorCount);
      HIPCHK(c, c->d_task_ctr.reserve(8));
      HIPCHK(c, hipMemsetAsync(c->d_task_ctr.p, 0, 8 * sizeof(int), c->stream));
      if (c->plan_ov.verbose)
        fprintf(stderr, "[icikt] K1 wide: lds=%zu B/wave, %d waves/CU, grid=%d, pairs=%lld\n", lds, per_cu, blocks,
                (long long)c->n_pairs);
      HIPCHK(c, icikt::launch_k1_wide(c->pv, c->d_pi.p, c->d_pj.p, c->d_raw.p, c->n_pairs, blocks, lds, c->d_task_ctr.p,
                                      c->stream));
      rc = timer_end(c, ICIKT_K_PAIRS, flags);
      if (rc) return rc;
      c->raw_valid = true;
    }
    rc = timer_begin(c, ICIKT_K_EPILOGUE, flags);
    if (rc) return rc;
    HIPCHK(c, icikt::launch_k2(c->pv, c->d_pi.p, c->d_pj.p, c->d_raw.p, c->n_pairs, perspective, alternative,
                               continuity ? 1 : 0, /*exact64=*/1, d_out4, d_counts, d_reasons, c->stream));
    return timer_end(c, ICIKT_K_EPILOGUE, flags);
  }
  // The pair kernel's counts (dis, joint ties, both-missing rows) do not depend on perspective, alternative or
  // continuity: with ICIKT_FLAG_REUSE_COUNTS a second run over the same prepared matrix and pair list (the other
  // perspective of BASELINE config 5, another alternative) is the epilogue alone.
  const bool reuse = (flags & ICIKT_FLAG_REUSE_COUNTS) && c->raw_valid;
  const bool tied = reuse ? false : matrix_tied(c, c->pv.n_samp, nullptr);
  const K1Plan pl = plan_k1(c->pv, c->n_pairs, c->prop.multiProcessorCount, c->plan_ov, tied, c->ntg_hint, c->group_hint);
  if (c->pv.n > 0 && !reuse) {
    if (c->wpb != pl.np) {
      build_units(c, pl.np);
      c->units_dirty = true;
    }
    if (c->units_dirty) {
      rc = upload_units(c);
      if (rc) return rc;
      c->units_dirty = false;
    }
  }
  if (c->pv.n > 0 && !reuse) {
    rc = timer_begin(c, ICIKT_K_PAIRS, flags);
    if (rc) return rc;
    rc = launch_pair_tasks(c, pl, 0, c->n_units);
    if (rc) return rc;
    rc = timer_end(c, ICIKT_K_PAIRS, flags);
    if (rc) return rc;
    c->raw_valid = true;
  }
  rc = timer_begin(c, ICIKT_K_EPILOGUE, flags);
  if (rc) return rc;
  HIPCHK(c, icikt::launch_k2(c->pv, c->d_pi.p, c->d_pj.p, c->d_raw.p, c->n_pairs, perspective, alternative,
                             continuity ? 1 : 0, (flags & ICIKT_FLAG_EXACT_INT64) ? 1 : 0, d_out4, d_counts,
                             d_reasons, c->stream));
  return timer_end(c, ICIKT_K_EPILOGUE, flags);
}

int icikt_kernel_ms(icikt_ctx* c, int kernel, double* ms, int64_t* launches) {
  if (!c || kernel < 0 || kernel >= ICIKT_K_COUNT) return ICIKT_E_INVALID;
  int rc = use_device(c);
  if (rc) return rc;
  rc = flush_timer(c, kernel);
  if (rc) return rc;
  if (ms) *ms = c->ms[kernel];
  if (launches) *launches = c->launches[kernel];
  return ICIKT_SUCCESS;
}

int icikt_reset_timers(icikt_ctx* c) {
  if (!c) return ICIKT_E_INVALID;
  for (int k = 0; k < ICIKT_K_COUNT; ++k) {
    int rc = flush_timer(c, k);
    if (rc) return rc;
    c->ms[k] = 0.0;
    c->launches[k] = 0;
  }
  return ICIKT_SUCCESS;
}

}  // extern "C"

namespace icikt {
namespace host {

// Host-side copies into / out of the library's pinned buffers, on a few threads: one core moves ~10 GB/s, the c4
// matrix is 82 MB and PCIe takes it in 1.8 ms.  The threads belong to the context (started on first use, parked on a
// condition variable between copies: creating and joining seven threads per chunk cost ~0.2 ms a time, a millisecond of
// a c4 call); a copy is cut into parts that the workers and the calling thread take from a shared counter.
struct CopyPool {
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable cv_work, cv_done;
  const std::function<void(unsigned)>* job = nullptr;
  unsigned nparts = 0, done = 0;
  std::atomic<unsigned> next{0};
  unsigned long long gen = 0;
  bool stop = false;
  unsigned active = 0;   // workers that hold the current job (a copy is over when all parts are done AND nobody holds it)
  void worker() {
    unsigned long long seen = 0;
    for (;;) {
      const std::function<void(unsigned)>* f;
      unsigned n;
      {
        std::unique_lock<std::mutex> lk(m);
        cv_work.wait(lk, [&] { return stop || gen != seen; });
        if (stop) return;
        seen = gen; f = job; n = nparts;
        if (!f) continue;          // woke after the copy was over
        ++active;
      }
      unsigned mine = 0;
      for (unsigned i = next.fetch_add(1); i < n; i = next.fetch_add(1)) { (*f)(i); ++mine; }
      {
        std::lock_guard<std::mutex> lk(m);
        done += mine;
        --active;
        if (done == nparts && active == 0) cv_done.notify_one();
      }
    }
  }
  bool start(unsigned n) {
    try {
      while (th.size() < n) th.emplace_back(&CopyPool::worker, this);
    } catch (...) {}
    return !th.empty();
  }
  void run(unsigned n, const std::function<void(unsigned)>& f) {
    {
      std::lock_guard<std::mutex> lk(m);
      job = &f; nparts = n; done = 0; next.store(0); ++gen;
    }
    cv_work.notify_all();
    unsigned mine = 0;
    for (unsigned i = next.fetch_add(1); i < n; i = next.fetch_add(1)) { f(i); ++mine; }
    std::unique_lock<std::mutex> lk(m);
    done += mine;
    cv_done.wait(lk, [&] { return done == nparts && active == 0; });
    job = nullptr;
  }
  ~CopyPool() {
    { std::lock_guard<std::mutex> lk(m); stop = true; }
    cv_work.notify_all();
    for (auto& t : th) t.join();
  }
};
void destroy_copy_pool(void* p) { delete static_cast<CopyPool*>(p); }

// `rows` pieces of `row_bytes`, strides in bytes (a contiguous copy: one row)
static void par_copy2d(icikt_ctx* c, void* dst, size_t dst_stride, const void* src, size_t src_stride, size_t row_bytes, size_t rows) {
  const size_t total = row_bytes * rows;
  unsigned nt = (unsigned)std::min<size_t>(8, total / ((size_t)1 << 20));   // (12 threads measured slower than 8 on the pool's boxes)
  CopyPool* pool = nullptr;
  if (nt > 1) {
    if (!c->copy_pool) { try { c->copy_pool = new CopyPool(); } catch (...) { c->copy_pool = nullptr; } }
    pool = static_cast<CopyPool*>(c->copy_pool);
    if (!pool || !pool->start(7)) pool = nullptr;
  }
  if (!pool) {   // small, or no threads to be had: one core does it all
    for (size_t r = 0; r < rows; ++r)
      memcpy(static_cast<char*>(dst) + r * dst_stride, static_cast<const char*>(src) + r * src_stride, row_bytes);
    return;
  }
  if (rows == 1) {   // a contiguous copy: pieces of ~1 MB
    const size_t piece = (size_t)1 << 20;
    const unsigned parts = (unsigned)((total + piece - 1) / piece);
    const std::function<void(unsigned)> f = [=](unsigned i) {
      const size_t off = (size_t)i * piece;
      memcpy(static_cast<char*>(dst) + off, static_cast<const char*>(src) + off, std::min(piece, total - off));
    };
    pool->run(parts, f);
    return;
  }
  const size_t per = std::max<size_t>(1, ((size_t)1 << 20) / std::max<size_t>(row_bytes, 1));   // rows per part: ~1 MB
  const unsigned parts = (unsigned)((rows + per - 1) / per);
  const std::function<void(unsigned)> f = [=](unsigned i) {
    const size_t r0 = (size_t)i * per, r1 = std::min(rows, r0 + per);
    for (size_t r = r0; r < r1; ++r)
      memcpy(static_cast<char*>(dst) + r * dst_stride, static_cast<const char*>(src) + r * src_stride, row_bytes);
  };
  pool->run(parts, f);
}
static void par_memcpy(icikt_ctx* c, void* dst, const void* src, size_t bytes) { par_copy2d(c, dst, 0, src, 0, bytes, 1); }

int ensure_bounce(icikt_ctx* c, size_t need) {
  if (c->pinned_bytes >= need) return ICIKT_SUCCESS;
  if (c->pinned) (void)hipHostFree(c->pinned);
  c->pinned = nullptr;
  c->pinned_bytes = 0;
  // (write-combined memory was tried for this buffer, which the host only ever writes: no difference, round 4)
  HIPCHK(c, hipHostMalloc(&c->pinned, need, hipHostMallocDefault));
  c->pinned_bytes = need;
  return ICIKT_SUCCESS;
}

// H2D of columns [col_begin, col_end) + K0 over them.  The copies run on the context's copy stream in column
// chunks and K0 of a chunk waits only for its own chunk (an event per chunk), so the pre-pass of chunk i runs
// while chunk i + 1 crosses PCIe.  How the caller's matrix is read:
//   staged (default)  through the library's pinned double buffer (a threaded host memcpy per chunk): the GPU never
//                     touches the caller's pages, and the library never page-locks them
//   pinned            ICIKT_FLAG_HOST_PINNED on the call: the caller has page-locked the matrix itself; the copies are
//                     DMA straight out of it
//   small             matrices below kLockMin bytes take the runtime's own staging path
// prepass: kPrepassFull = K0; kPrepassMask = missing-row bitsets only (pairwise_completeness); kPrepassNone = the
// columns are only copied (icikt_pairs_complete_f64 sorts masked copies, not the columns).
int upload_and_prepare(icikt_ctx* c, const double* X, int64_t n_feat, int64_t n_samp, int64_t ld, int64_t col_begin,
                       int64_t col_end, uint32_t flags, bool pipelined, const std::function<int(size_t, int64_t)>* on_chunk,
                       int prepass) {
  // what runs over the columns of a chunk once they are on the device
  auto chunk_prepass = [&](int64_t c0, int64_t c1, hipStream_t st) -> int {
    if (prepass == kPrepassFull) return prepare_launch(c, c->d_X.p, n_feat, c0, c1, st);
    if (prepass == kPrepassMask) HIPCHK(c, icikt::launch_k0_mask(c->pv, c->d_X.p, n_feat, (int)c0, (int)(c1 - c0), st ? st : c->stream));
    return ICIKT_SUCCESS;
  };
  c->chunk_col_end.clear();
  const size_t nel = (size_t)std::max<int64_t>(n_feat * n_samp, 1);
  HIPCHK(c, c->d_X.reserve(nel));
  int rc = timer_begin(c, ICIKT_K_PREPARE, flags);
  if (rc) return rc;
  const int64_t ncols = col_end - col_begin;
  if (n_feat > 0 && ncols > 0) {
    const size_t col_bytes = (size_t)n_feat * sizeof(double);
    // ~8 MB per chunk (an even number of columns: K0 ranges need not be even, but keeps chunks aligned)
    int64_t chunk = std::max<int64_t>(2, (int64_t)(((size_t)8 << 20) / std::max<size_t>(col_bytes, 1)) & ~(int64_t)1);
    // pipelined: a chunk is also what one pre-pass launch and one pair-kernel launch cover.  A pre-pass launch takes
    // the time of ONE column's sort up to a workgroup per CU, so chunks of fewer columns only lengthen the chain of
    // pre-pass launches the last pair-kernel launch waits for (c4: ten chunks of 104 columns finished their pre-pass
    // 1.4 ms after the last copy, four chunks of 256 right behind it)
    if (pipelined) chunk = std::max<int64_t>(chunk, std::min<int64_t>(c->prop.multiProcessorCount, (n_samp / 4) & ~(int64_t)1));
    if (prepass == kPrepassFull) chunk = std::min<int64_t>(chunk, std::max(1, c->sort_chunk));
    const double* src0 = X + col_begin * ld;
    const size_t span = ((size_t)(ncols - 1) * (size_t)ld + (size_t)n_feat) * sizeof(double);
    const int mode = (span < kLockMin) ? 0 : (c->host_pinned ? 3 : 2);
    if (mode == 2) {
      rc = ensure_bounce(c, 2 * (size_t)chunk * col_bytes);
      if (rc) return rc;
    }
    // the copy stream must not overwrite d_X while earlier work on the compute stream still reads it
    hipError_t e = hipEventRecord(c->ev_copy[0], c->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->copy_stream, c->ev_copy[0], 0);
    if (e == hipSuccess && pipelined) e = hipStreamWaitEvent(c->prep_stream, c->ev_copy[0], 0);
    int k = 0;
    const int64_t first_chunk = chunk;   // (a short first chunk was measured twice: no gain -- what the last pair-kernel launch waits for is the last chunk's pre-pass, which finds no free CU until the launch before it drains)
    for (int64_t c0 = col_begin, step = first_chunk; c0 < col_end && e == hipSuccess && rc == 0; c0 += step, step = chunk, ++k) {
      const int64_t nc = std::min<int64_t>(step, col_end - c0);
      double* dst = c->d_X.p + (size_t)c0 * (size_t)n_feat;
      hipEvent_t ev = c->ev_copy[1 + (k % 3)];
      if (mode == 2) {
        char* stage = static_cast<char*>(c->pinned) + (size_t)(k & 1) * (size_t)chunk * col_bytes;
        if (k >= 2) e = hipEventSynchronize(c->ev_copy[1 + ((k - 2) % 3)]);  // the copy that last used this half
        if (e != hipSuccess) break;
        par_copy2d(c, stage, col_bytes, X + c0 * ld, (size_t)ld * sizeof(double), col_bytes, (size_t)nc);
        e = hipMemcpyAsync(dst, stage, (size_t)nc * col_bytes, hipMemcpyHostToDevice, c->copy_stream);
      } else {
        e = hipMemcpy2DAsync(dst, col_bytes, X + c0 * ld, (size_t)ld * sizeof(double), col_bytes, (size_t)nc,
                             hipMemcpyHostToDevice, c->copy_stream);
      }
      if (e == hipSuccess) e = hipEventRecord(ev, c->copy_stream);
      if (pipelined) {
        // the chunk's pre-pass on the pre-pass stream; an event of its own tells the pair kernel's stream when
        if (e == hipSuccess) e = hipStreamWaitEvent(c->prep_stream, ev, 0);
        // (Every chunk's pre-pass runs the 1 024-thread shape.  A workgroup of it needs an EMPTY CU, so a later chunk's
        //  pre-pass starts only when the pair-kernel launch in front of it drains -- round 3 put 0.5 ms of a c4 call down to
        //  that.  Round 4 built the 256-thread shape (k0_prepare_small: fits beside a running pair kernel, plan key k0=1) and
        //  measured it here: the pre-pass then does finish earlier, 3.9 against 6.4 ms into the call, but the call gets
        //  LONGER, 11.76 against 11.61 ms staged, 11.24 against 11.13 pinned (profiles/r04_ab_pipe_k0.log) -- both kernels
        //  are bound by vector issue, so overlapping them gains nothing, and the small shape takes 0.80 ms for the 1 024
        //  columns where the large one takes 0.65.)
        if (e == hipSuccess) rc = chunk_prepass(c0, c0 + nc, c->prep_stream);
        if ((size_t)k >= c->ev_chunk.size()) {
          hipEvent_t ne = nullptr;
          if (e == hipSuccess) e = hipEventCreateWithFlags(&ne, hipEventDisableTiming);
          if (e == hipSuccess) c->ev_chunk.push_back(ne);
        }
        if (e == hipSuccess && rc == 0) e = hipEventRecord(c->ev_chunk[(size_t)k], c->prep_stream);
        if (e == hipSuccess && rc == 0) c->chunk_col_end.push_back(c0 + nc);
        // the caller's work for this chunk (its pair-kernel launch) is enqueued NOW, before the host stages the next chunk
        if (e == hipSuccess && rc == 0 && on_chunk) rc = (*on_chunk)((size_t)k, c0 + nc);
      } else {
        if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, ev, 0);
        if (e == hipSuccess) rc = chunk_prepass(c0, c0 + nc, nullptr);
      }
    }
    if (pipelined && (e != hipSuccess || rc)) (void)hipStreamSynchronize(c->prep_stream);
    // The staging buffer serves the later transfers of the call as their bounce buffer, so its last copy is waited for
    // here.  A matrix the caller has page-locked is read in place: the pipelined callers go on with host work (the task
    // list) and wait for the copy stream themselves before they return (finish_upload) -- no entry point returns while
    // a copy still reads the caller's memory.
    if (!(pipelined && mode == 3) || e != hipSuccess || rc) (void)hipStreamSynchronize(c->copy_stream);
    if (e != hipSuccess) return fail(c, ICIKT_E_HIP, std::string("H2D of the matrix: ") + hipGetErrorString(e));
    if (rc) return rc;
  } else if (ncols > 0 && prepass == kPrepassFull) {
    rc = prepare_launch(c, c->d_X.p, n_feat, col_begin, col_end);  // n_feat == 0: statistics of empty columns
    if (rc) return rc;
  }
  return timer_end(c, ICIKT_K_PREPARE, flags);
}

// The pair kernel's task list for the prepared shape and the current pair list, built on the host now (while
// copies and the pre-pass run) instead of inside icikt_run_dev, which then only uploads it.
void prebuild_units(icikt_ctx* c) {
  if (c->n_pairs <= 0 || c->pv.wide) return;
  const K1Plan pl = plan_k1(c->pv, c->n_pairs, c->prop.multiProcessorCount, c->plan_ov, false);   // (icikt_run_dev rebuilds the list if the columns' tie structure asks for another np)
  if (c->wpb != pl.np) {
    build_units(c, pl.np);
    c->units_dirty = true;
  }
}

// The host entries' H2D + pre-pass + pair kernel.  A matrix of several 8 MB chunks is PIPELINED: all copies and all
// pre-pass launches are enqueued at once (copy stream -> pre-pass stream, an event per chunk); meanwhile the host
// builds the task list, orders it by the chunk in which a task's LAST column arrives, and enqueues one pair-kernel
// launch per chunk behind that chunk's pre-pass event.  The pairs among the columns that have arrived are counted
// while the rest of the matrix still crosses PCIe: the available work grows with the square of the arrived columns,
// so after the first millisecond of a c4-sized call the GPU never waits for the link again.
int upload_prepare_pairs(icikt_ctx* c, const double* X, int64_t n_feat, int64_t n_samp, int64_t ld, uint32_t flags) {
  const size_t col_bytes = (size_t)n_feat * sizeof(double);
  const size_t span = (n_samp > 0 && n_feat > 0) ? ((size_t)(n_samp - 1) * (size_t)ld + (size_t)n_feat) * sizeof(double) : 0;
  const bool can = !c->pv.wide && n_feat > 0 && c->n_pairs > 0;
  const bool want = c->pipe_mode < 0 ? (span >= ((size_t)24 << 20)) : (c->pipe_mode == 1 && (size_t)n_samp * col_bytes > ((size_t)8 << 20));
  // A pair list that does not fill the chip several times: copies and pre-pass stay pipelined by chunks, but the pairs run in
  // ONE launch behind the last chunk's pre-pass -- a launch of a few thousand tasks lasts as long as one task whatever its
  // size, and a launch per chunk puts those latencies one behind the other (50 000 x 96 tied columns: 15.5 -> 9.5 ms)
  const bool merge_launches = c->plan_ov.merge >= 0 ? c->plan_ov.merge != 0 : c->n_pairs < (int64_t)4 * 48 * c->prop.multiProcessorCount;
  if (!(can && want)) {
    int rc = upload_and_prepare(c, X, n_feat, n_samp, ld, 0, n_samp, flags);
    if (rc) return rc;
    c->prepared = true;
    prebuild_units(c);   // host work under the copies; icikt_run_dev uploads the list
    return ICIKT_SUCCESS;   // (the caller's icikt_run_dev runs the pair kernel)
  }
  const auto t0 = std::chrono::steady_clock::now();
  auto ms_since = [&t0]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  K1Plan pl = plan_k1(c->pv, c->n_pairs, c->prop.multiProcessorCount, c->plan_ov, false);
  const bool all_combn = c->combn_S == n_samp && c->combn_begin == 0 && c->combn_end == n_samp * (n_samp - 1) / 2;
  double t_enq = 0, t_built = 0, t_up = 0;
  int rc = ICIKT_SUCCESS;
  size_t nchunks = 0;
  if (all_combn) {
    // All pairs of the upper triangle: a chunk's tasks are arithmetic.  They are written chunk by chunk straight into a
    // pinned buffer, copied asynchronously and launched AS SOON AS the chunk's copy and pre-pass are enqueued (the
    // callback below runs inside upload_and_prepare's chunk loop: with staged transfers the host is busy copying the
    // next chunk into its pinned buffer meanwhile) -- the first launch is enqueued a fraction of a millisecond into the
    // call, no host copy of the pair list is ever made.  Order inside a chunk: gathered block, then streamed
    // column (the order build_units gives: a block's rec table stays in cache); a task is the two pairs (2a, j),
    // (2a + 1, j) that share their streamed column j, the pair (2a, 2a + 1) runs alone.
    const int64_t S = n_samp;
    auto pidx = [S](int64_t i, int64_t j) { return (int32_t)(i * (2 * S - i - 1) / 2 + (j - i - 1)); };
    const size_t cap_tasks = (size_t)c->n_pairs + (size_t)S + 8;
    HIPCHK(c, c->d_unit_start.reserve(cap_tasks * 2));
    if (c->pinned_tasks_bytes < cap_tasks * 2 * sizeof(int32_t)) {
      if (c->pinned_tasks) (void)hipHostFree(c->pinned_tasks);
      c->pinned_tasks = nullptr; c->pinned_tasks_bytes = 0;
      HIPCHK(c, hipHostMalloc(&c->pinned_tasks, cap_tasks * 2 * sizeof(int32_t), hipHostMallocDefault));
      c->pinned_tasks_bytes = cap_tasks * 2 * sizeof(int32_t);
    }
    int32_t* u = static_cast<int32_t*>(c->pinned_tasks);
    size_t nt = 0;
    int64_t cb = 0;
    const std::function<int(size_t, int64_t)> on_chunk = [&](size_t q, int64_t ce) -> int {
      // (18 337 .. 30 656 rows: the first chunk's columns say which kernel family runs; both take two pairs per task)
      if (q == 0) {
        const bool tied = matrix_tied(c, ce, c->ev_chunk[0]);
        if (tied || c->ntg_hint >= 0) pl = plan_k1(c->pv, c->n_pairs, c->prop.multiProcessorCount, c->plan_ov, tied, c->ntg_hint, c->group_hint);
      }
      const size_t first_q = nt;
      if (pl.np == 2) {
        for (int64_t a2 = 0; a2 < ce; a2 += 2) {          // gathered block: columns a2, a2 + 1
          const int64_t c1 = a2 + 1;
          for (int64_t j = std::max(cb, a2 + 1); j < ce; ++j) {
            if (j == c1) { u[2 * nt] = pidx(a2, c1); u[2 * nt + 1] = -1; }
            else { u[2 * nt] = pidx(a2, j); u[2 * nt + 1] = pidx(c1, j); }
            ++nt;
          }
        }
      } else {
        for (int64_t i = 0; i < ce; ++i)
          for (int64_t j = std::max(cb, i + 1); j < ce; ++j) { u[2 * nt] = pidx(i, j); u[2 * nt + 1] = -1; ++nt; }
      }
      cb = ce;
      if (merge_launches && ce < n_samp) return ICIKT_SUCCESS;   // (the whole list in one launch, behind the last chunk)
      const size_t first = merge_launches ? 0 : first_q;
      const size_t cnt = nt - first;
      if (cnt == 0) return ICIKT_SUCCESS;
      // (the launches stay on ONE stream: alternating consecutive chunks' launches between two streams, so that one starts
      //  while the other drains, was measured slower -- 12.3 against 11.6 ms on c4, round 4: two launches in flight share
      //  the SIMDs and the cache with the small-shape pre-pass and each other)
      hipError_t e = hipMemcpyAsync(c->d_unit_start.p + 2 * first, u + 2 * first, cnt * 2 * sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
      if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_chunk[q], 0);
      if (e != hipSuccess) return fail(c, ICIKT_E_HIP, std::string("pipelined pairs: ") + hipGetErrorString(e));
      return launch_pair_tasks(c, pl, (int)first, (int)cnt);
    };
    rc = timer_begin(c, ICIKT_K_PAIRS, flags);
    if (rc == 0) rc = upload_and_prepare(c, X, n_feat, n_samp, ld, 0, n_samp, flags & ~ICIKT_FLAG_TIMING, true, &on_chunk);
    if (rc) { (void)hipStreamSynchronize(c->prep_stream); return rc; }
    c->prepared = true;
    nchunks = c->chunk_col_end.size();
    c->n_units = (int)nt;
    c->wpb = 0;               // (h_units does not hold this list: a later device-resident run rebuilds it)
    c->units_dirty = true;
    t_enq = t_built = t_up = ms_since();
  } else {
  rc = upload_and_prepare(c, X, n_feat, n_samp, ld, 0, n_samp, flags & ~ICIKT_FLAG_TIMING, true, nullptr);
  if (rc) return rc;
  c->prepared = true;
  t_enq = ms_since();
  nchunks = c->chunk_col_end.size();
  if (nchunks > 0) {
    const bool tied = matrix_tied(c, c->chunk_col_end[0], c->ev_chunk[0]);
    if (tied || c->ntg_hint >= 0) pl = plan_k1(c->pv, c->n_pairs, c->prop.multiProcessorCount, c->plan_ov, tied, c->ntg_hint, c->group_hint);
  }
  build_units(c, pl.np);
  // tasks by the chunk of their last column (a stable counting sort: inside a chunk the cache-friendly order stays)
  const int T = c->n_units;
  std::vector<int> first(nchunks + 1, 0);
  {
    std::vector<int32_t> col_chunk((size_t)n_samp, 0);
    size_t k = 0;
    for (int64_t col = 0; col < n_samp; ++col) {
      while (k + 1 < nchunks && col >= c->chunk_col_end[k]) ++k;
      col_chunk[(size_t)col] = (int32_t)k;
    }
    std::vector<int32_t> tchunk((size_t)T);
    ensure_host_pairs(c);
    const int32_t* pi = c->h_pi.data();
    const int32_t* pj = c->h_pj.data();
    for (int t = 0; t < T; ++t) {
      const int32_t p0 = c->h_units[2 * (size_t)t], p1 = c->h_units[2 * (size_t)t + 1];
      int32_t col = std::max(pi[p0], pj[p0]);
      if (p1 >= 0) col = std::max(col, std::max(pi[p1], pj[p1]));
      tchunk[(size_t)t] = col_chunk[(size_t)col];
      first[(size_t)tchunk[(size_t)t] + 1] += 1;
    }
    for (size_t q = 0; q < nchunks; ++q) first[q + 1] += first[q];
    std::vector<int> fill(first.begin(), first.end() - 1);
    std::vector<int32_t> sorted((size_t)T * 2);
    for (int t = 0; t < T; ++t) {
      const int d = fill[(size_t)tchunk[(size_t)t]]++;
      sorted[2 * (size_t)d] = c->h_units[2 * (size_t)t];
      sorted[2 * (size_t)d + 1] = c->h_units[2 * (size_t)t + 1];
    }
    c->h_units.swap(sorted);
  }
  t_built = ms_since();
  rc = upload_units(c);
  if (rc) { (void)hipStreamSynchronize(c->prep_stream); return rc; }
  c->units_dirty = false;
  t_up = ms_since();
  rc = timer_begin(c, ICIKT_K_PAIRS, flags);
  for (size_t q = 0; q < nchunks && rc == 0; ++q) {
    if (merge_launches && q + 1 < nchunks) continue;   // (one launch behind the last chunk: see above)
    const hipError_t e = hipStreamWaitEvent(c->stream, c->ev_chunk[q], 0);
    if (e != hipSuccess) { rc = fail(c, ICIKT_E_HIP, std::string("pipelined pairs: ") + hipGetErrorString(e)); break; }
    if (merge_launches) rc = launch_pair_tasks(c, pl, first[0], first[nchunks] - first[0]);
    else rc = launch_pair_tasks(c, pl, first[q], first[q + 1] - first[q]);
  }
  }
  if (rc) { (void)hipStreamSynchronize(c->prep_stream); return rc; }
  rc = timer_end(c, ICIKT_K_PAIRS, flags);
  if (rc) return rc;
  c->raw_valid = true;
  if (c->plan_ov.verbose) {
    const double t_launched = ms_since();
    (void)hipStreamSynchronize(c->copy_stream);
    const double t_copied = ms_since();
    (void)hipStreamSynchronize(c->prep_stream);
    const double t_prepped = ms_since();
    (void)hipStreamSynchronize(c->stream);
    fprintf(stderr, "[icikt] pipelined: %zu chunks; enqueued copies + pre-pass %.2f ms, tasks built %.2f, uploaded %.2f, pair launches "
                    "enqueued %.2f, copies done %.2f, pre-pass done %.2f, pairs done %.2f ms\n", nchunks, t_enq, t_built, t_up,
            t_launched, t_copied, t_prepped, ms_since());
  }
  return ICIKT_SUCCESS;
}

int upload_sync(icikt_ctx* c, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return ICIKT_SUCCESS;
  if (bytes >= kLockMin) {
    // through the library's pinned bounce buffer, a chunk at a time (pair and task lists: the library's own vectors
    // or the caller's index arrays -- ICIKT_FLAG_HOST_PINNED speaks of the matrix and the result arrays only)
    const size_t cap = (size_t)8 << 20;
    int rc = ensure_bounce(c, std::min(bytes, cap));
    if (rc) return rc;
    for (size_t off = 0; off < bytes; off += cap) {
      const size_t m = std::min(cap, bytes - off);
      par_memcpy(c, c->pinned, static_cast<const char*>(src) + off, m);
      hipError_t e = hipMemcpyAsync(static_cast<char*>(dst) + off, c->pinned, m, hipMemcpyHostToDevice, c->stream);
      const hipError_t es = hipStreamSynchronize(c->stream);
      if (e == hipSuccess) e = es;
      if (e != hipSuccess) return fail(c, ICIKT_E_HIP, std::string("H2D copy (staged): ") + hipGetErrorString(e));
    }
    return ICIKT_SUCCESS;
  }
  hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream);
  const hipError_t es = hipStreamSynchronize(c->stream);       // the host range must outlive the copy
  if (e == hipSuccess) e = es;
  if (e != hipSuccess) return fail(c, ICIKT_E_HIP, std::string("H2D copy: ") + hipGetErrorString(e));
  return ICIKT_SUCCESS;
}

int download(icikt_ctx* c, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return ICIKT_SUCCESS;
  if (bytes >= kLockMin && !c->host_pinned) {
    // into a pinned buffer of the library's (kept from call to call: slot = position among the call's downloads);
    // finish_downloads() moves it to the caller's array
    const size_t slot = c->bounced_out.size();
    if (slot >= c->out_pinned.size()) c->out_pinned.resize(slot + 1);
    auto& ps = c->out_pinned[slot];
    if (ps.bytes < bytes) {
      if (ps.p) (void)hipHostFree(ps.p);
      ps.p = nullptr; ps.bytes = 0;
      HIPCHK(c, hipHostMalloc(&ps.p, bytes, hipHostMallocDefault));
      ps.bytes = bytes;
    }
    // in pieces, an event behind each: finish_stream() moves a piece to the caller's array while the next ones are
    // still on their way (c4: 19 MB of results, 0.4 ms of PCIe and 0.25 ms of host copy that used to run one after the other)
    const size_t piece = std::max<size_t>((size_t)4 << 20, (bytes + 3) / 4);
    for (size_t off = 0; off < bytes; off += piece) {
      const size_t m = std::min(piece, bytes - off);
      HIPCHK(c, hipMemcpyAsync(static_cast<char*>(ps.p) + off, static_cast<const char*>(src) + off, m, hipMemcpyDeviceToHost, c->stream));
      hipEvent_t ev = nullptr;
      if (c->ev_out_used < c->ev_out.size()) ev = c->ev_out[c->ev_out_used];
      else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) c->ev_out.push_back(ev);
      else { (void)hipGetLastError(); ev = nullptr; }
      if (ev) { c->ev_out_used += 1; HIPCHK(c, hipEventRecord(ev, c->stream)); }
      c->bounced_out.push_back(icikt_ctx::Bounce{static_cast<char*>(ps.p) + off, static_cast<char*>(dst) + off, m, ev});
    }
    return ICIKT_SUCCESS;
  }
  HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  return ICIKT_SUCCESS;
}

// (the stream has been synchronised; `ok` = it ended without an error, i.e. the bounced bytes are the results)
void finish_downloads(icikt_ctx* c, bool ok) {
  for (auto& b : c->bounced_out)
    if (ok) par_memcpy(c, b.dst, b.pinned, b.bytes);
  c->bounced_out.clear();
  c->ev_out_used = 0;
}

hipError_t finish_stream(icikt_ctx* c, bool ok) {
  for (auto& b : c->bounced_out) {
    if (!ok) break;
    // (a piece without an event -- none could be created -- waits for everything enqueued so far)
    const hipError_t e = b.ev ? hipEventSynchronize(b.ev) : hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { (void)hipGetLastError(); ok = false; break; }
    par_memcpy(c, b.dst, b.pinned, b.bytes);
  }
  c->bounced_out.clear();
  c->ev_out_used = 0;
  return hipStreamSynchronize(c->stream);
}

}  // namespace host
}  // namespace icikt

// every index of a host pair list inside [0, n_samp)
static int check_pair_list(icikt_ctx* c, const char* who, const int32_t* pi, const int32_t* pj, int64_t n_pairs,
                           int64_t n_samp) {
  if (n_pairs < 0 || (n_pairs > 0 && (!pi || !pj))) return fail(c, ICIKT_E_INVALID, std::string(who) + ": bad pair list");
  for (int64_t p = 0; p < n_pairs; ++p)
    if (pi[p] < 0 || pi[p] >= n_samp || pj[p] < 0 || pj[p] >= n_samp)
      return fail(c, ICIKT_E_INVALID, std::string(who) + ": column index out of range");
  return ICIKT_SUCCESS;
}

extern "C" {

int icikt_pairs_f64(icikt_ctx* c, const double* X, int64_t n_feat, int64_t n_samp, int64_t ld,
                    const int32_t* pi, const int32_t* pj, int64_t n_pairs, int perspective, int alternative,
                    int continuity, uint32_t flags, double* out4, int64_t* counts, int32_t* reasons) {
  if (!c) return ICIKT_E_INVALID;
  // every argument is validated before the first asynchronous copy reads the caller's memory
  int rc = check_shape(c, "pairs", n_feat, n_samp, ld);
  if (rc) return rc;
  if (n_feat > 0 && n_samp > 0 && !X) return fail(c, ICIKT_E_INVALID, "pairs: null matrix");
  if (pi == nullptr) {
    if (pj != nullptr) return fail(c, ICIKT_E_INVALID, "pairs: pi is null but pj is not");
    n_pairs = n_samp * (n_samp - 1) / 2;
  } else {
    rc = check_pair_list(c, "pairs", pi, pj, n_pairs, n_samp);
    if (rc) return rc;
  }
  if (n_pairs > 0 && !out4) return fail(c, ICIKT_E_INVALID, "pairs: null output");
  if (perspective != ICIKT_PERSPECTIVE_LOCAL && perspective != ICIKT_PERSPECTIVE_GLOBAL)
    return fail(c, ICIKT_E_INVALID, "pairs: perspective must be local (0) or global (1)");
  if (alternative < 0 || alternative > ICIKT_ALT_OTHER) return fail(c, ICIKT_E_INVALID, "pairs: bad alternative code");
  rc = use_device(c);
  if (rc) return rc;
  rc = pi ? icikt_set_pairs(c, pi, pj, n_pairs) : icikt_set_pairs_combn(c, n_samp, 0, n_pairs);
  if (rc) return rc;
  rc = prepare_alloc(c, n_feat, n_samp, n_samp, n_samp);
  if (rc) return rc;
  // The copies and the pre-pass are only ENQUEUED here; the host builds the pair kernel's task list while the matrix
  // crosses PCIe (it used to wait for the copies first and build the list afterwards, with the GPU idle: 1.2 ms of
  // 14.3 on c4).  Nothing may still read the caller's matrix when the call returns: finish_upload.
  const icikt::host::PinnedScope pinned_scope(c, flags);
  rc = icikt::host::upload_prepare_pairs(c, X, n_feat, n_samp, ld, flags);
  auto finish_upload = [&]() {
    (void)hipStreamSynchronize(c->prep_stream);
    (void)hipStreamSynchronize(c->copy_stream);
  };
  if (rc) { finish_upload(); return rc; }
  const int64_t P = c->n_pairs;
  if (P == 0) {
    const hipError_t e0 = hipStreamSynchronize(c->stream);
    finish_upload();
    if (e0 != hipSuccess) return fail(c, ICIKT_E_HIP, std::string("pairs: ") + hipGetErrorString(e0));
    return ICIKT_SUCCESS;
  }
  auto body = [&]() -> int {
    HIPCHK(c, c->d_out4.reserve((size_t)P * 4));
    if (counts) HIPCHK(c, c->d_counts.reserve((size_t)P * ICIKT_CNT_FIELDS));
    if (reasons) HIPCHK(c, c->d_reasons.reserve((size_t)P));
    // (pipelined: the pair kernel has been enqueued chunk by chunk already -- raw_valid -- and this is the epilogue alone)
    int r = icikt_run_dev(c, perspective, alternative, continuity, flags | (c->raw_valid ? ICIKT_FLAG_REUSE_COUNTS : 0u),
                          c->d_out4.p, counts ? c->d_counts.p : nullptr, reasons ? c->d_reasons.p : nullptr);
    if (r) return r;
    r = icikt::host::download(c, out4, c->d_out4.p, (size_t)P * 4 * sizeof(double));
    if (!r && counts) r = icikt::host::download(c, counts, c->d_counts.p, (size_t)P * ICIKT_CNT_FIELDS * sizeof(int64_t));
    if (!r && reasons) r = icikt::host::download(c, reasons, c->d_reasons.p, (size_t)P * sizeof(int32_t));
    return r;
  };
  rc = body();
  // success or not: nothing may still be reading or writing the caller's buffers when this returns
  const hipError_t es = icikt::host::finish_stream(c, rc == 0);
  finish_upload();
  if (rc) return rc;
  if (es != hipSuccess) return fail(c, ICIKT_E_HIP, std::string("pairs: ") + hipGetErrorString(es));
  return ICIKT_SUCCESS;
}

}  // extern "C"

namespace icikt {
namespace host {

// global_na (R/utils.R:1-23) -> the pre-pass's exclusion rule
int make_mask_spec(icikt_ctx* c, const double* global_na, int n_global_na, icikt::MaskSpec* ms) {
  *ms = icikt::MaskSpec{};
  if (n_global_na < 0 || (n_global_na > 0 && !global_na)) return fail(c, ICIKT_E_INVALID, "matrix: bad global_na");
  for (int k = 0; k < n_global_na; ++k) {
    const double v = global_na[k];
    if (v != v) ms->mask_nan = 1;
    else if (v - v != 0.0) ms->mask_inf = 1;   // +-Inf
    else {
      bool dup = false;
      for (int q = 0; q < ms->n_vals; ++q) dup = dup || ms->vals[q] == v;
      if (dup) continue;
      if (ms->n_vals == icikt::ICIKT_MASK_VALS)
        return fail(c, ICIKT_E_INVALID, "matrix: more than 32 distinct finite values in global_na (mask the matrix on the host and pass NaN)");
      ms->vals[ms->n_vals++] = v;
    }
  }
  return ICIKT_SUCCESS;
}

}  // namespace host
}  // namespace icikt

extern "C" {

// ici_kendalltau() below the argument checks, in one call: exclusion rule + pre-pass + pair kernel + epilogue +
// scale_and_reshape on the device, one D2H of the five matrices.
int icikt_matrix_f64(icikt_ctx* c, const double* X, int64_t n_feat, int64_t n_samp, int64_t ld, const double* global_na,
                     int n_global_na, const int32_t* pi, const int32_t* pj, int64_t n_pairs, int perspective,
                     int alternative, int continuity, uint32_t flags, int scale_max, int diag_good, double* out5,
                     uint8_t* keep, int64_t* reason_counts) {
  if (!c) return ICIKT_E_INVALID;
  int rc = check_shape(c, "matrix", n_feat, n_samp, ld);
  if (rc) return rc;
  if (n_feat > 0 && n_samp > 0 && !X) return fail(c, ICIKT_E_INVALID, "matrix: null matrix");
  if (pi == nullptr) {
    if (pj != nullptr) return fail(c, ICIKT_E_INVALID, "matrix: pi is null but pj is not");
    n_pairs = n_samp * (n_samp - 1) / 2;
  } else {
    rc = check_pair_list(c, "matrix", pi, pj, n_pairs, n_samp);
    if (rc) return rc;
  }
  if (n_samp > 0 && !out5) return fail(c, ICIKT_E_INVALID, "matrix: null output");
  if (perspective != ICIKT_PERSPECTIVE_LOCAL && perspective != ICIKT_PERSPECTIVE_GLOBAL)
    return fail(c, ICIKT_E_INVALID, "matrix: perspective must be local (0) or global (1)");
  if (alternative < 0 || alternative > ICIKT_ALT_OTHER) return fail(c, ICIKT_E_INVALID, "matrix: bad alternative code");
  icikt::MaskSpec ms;
  rc = icikt::host::make_mask_spec(c, global_na, n_global_na, &ms);
  if (rc) return rc;
  if (reason_counts) for (int k = 0; k < 5; ++k) reason_counts[k] = 0;
  if (n_samp == 0) return ICIKT_SUCCESS;
  rc = use_device(c);
  if (rc) return rc;
  rc = pi ? icikt_set_pairs(c, pi, pj, n_pairs) : icikt_set_pairs_combn(c, n_samp, 0, n_pairs);
  if (rc) return rc;
  rc = prepare_alloc(c, n_feat, n_samp, n_samp, n_samp);
  if (rc) return rc;
  const size_t S = (size_t)n_samp, P = (size_t)c->n_pairs;
  const size_t keep_bytes = keep ? S * (size_t)n_feat : 0;
  if (keep_bytes) HIPCHK(c, c->d_keep.reserve(keep_bytes));
  HIPCHK(c, c->d_out5.reserve(5 * S * S));
  HIPCHK(c, c->d_red.reserve(8));
  HIPCHK(c, c->d_out4.reserve(std::max<size_t>(P, 1) * 4));
  HIPCHK(c, c->d_reasons.reserve(std::max<size_t>(P, 1)));
  const icikt::host::PinnedScope pinned_scope(c, flags);
  c->k0_mask = &ms;
  c->k0_keep = keep_bytes ? c->d_keep.p : nullptr;
  rc = icikt::host::upload_prepare_pairs(c, X, n_feat, n_samp, ld, flags);
  c->k0_mask = nullptr;
  c->k0_keep = nullptr;
  auto finish_upload = [&]() {
    (void)hipStreamSynchronize(c->prep_stream);
    (void)hipStreamSynchronize(c->copy_stream);
  };
  if (rc) { finish_upload(); return rc; }
  unsigned long long red[8] = {};
  auto body = [&]() -> int {
    int r = icikt_run_dev(c, perspective, alternative, continuity, flags | (c->raw_valid ? ICIKT_FLAG_REUSE_COUNTS : 0u),
                          c->d_out4.p, nullptr, c->d_reasons.p);
    if (r) return r;
    HIPCHK(c, icikt::launch_out_stats(c->pv, c->d_out4.p, c->d_reasons.p, (int64_t)P, nullptr, c->d_red.p, c->stream));
    HIPCHK(c, icikt::launch_assemble(c->pv, c->d_out4.p, c->d_pi.p, c->d_pj.p, (int64_t)P, nullptr, c->d_red.p,
                                     scale_max ? 1 : 0, diag_good ? 1 : 0, c->d_out5.p, c->stream));
    r = icikt::host::download(c, out5, c->d_out5.p, 5 * S * S * sizeof(double));
    if (!r && keep_bytes) r = icikt::host::download(c, keep, c->d_keep.p, keep_bytes);
    if (!r) r = icikt::host::download(c, red, c->d_red.p, sizeof(red));
    return r;
  };
  rc = body();
  const hipError_t es = icikt::host::finish_stream(c, rc == 0);
  finish_upload();
  if (rc) return rc;
  if (es != hipSuccess) return fail(c, ICIKT_E_HIP, std::string("matrix: ") + hipGetErrorString(es));
  if (reason_counts) for (int k = 0; k < 5; ++k) reason_counts[k] = (int64_t)red[1 + k];
  return ICIKT_SUCCESS;
}

}  // extern "C"

extern "C" {

// kt_fast(use = "pairwise.complete.obs"): every pair gets its own two columns with the rows that miss either value
// masked in both (k_mask_pairs), sorted (K0) and counted (K1, one pair per wave) on the device; the perspective is
// "local" by construction.  Pairs go through in chunks that keep the masked columns within ~1.5 GiB.
int icikt_pairs_complete_f64(icikt_ctx* c, const double* X, int64_t n_feat, int64_t n_samp, int64_t ld,
                             const int32_t* pi, const int32_t* pj, int64_t n_pairs, int alternative, int continuity,
                             uint32_t flags, double* out4, int64_t* counts, int32_t* reasons) {
  if (!c) return ICIKT_E_INVALID;
  int rc = check_shape(c, "pairs_complete", n_feat, n_samp, ld, /*wide_ok=*/false);
  if (rc) return rc;
  if (n_feat > 0 && n_samp > 0 && !X) return fail(c, ICIKT_E_INVALID, "pairs_complete: null matrix");
  rc = check_pair_list(c, "pairs_complete", pi, pj, n_pairs, n_samp);
  if (rc) return rc;
  if (n_pairs == 0) return ICIKT_SUCCESS;
  if (!out4) return fail(c, ICIKT_E_INVALID, "pairs_complete: null output");
  if (alternative < 0 || alternative > ICIKT_ALT_OTHER) return fail(c, ICIKT_E_INVALID, "pairs_complete: bad alternative code");
  rc = use_device(c);
  if (rc) return rc;
  // the matrix and the pair list, once
  rc = icikt_set_pairs(c, pi, pj, n_pairs);
  if (rc) return rc;
  const icikt::host::PinnedScope pinned_scope(c, flags);
  DevBuf<int32_t> all_pi, all_pj;  // the caller's list stays on the device while d_pi / d_pj hold a chunk's (2k, 2k+1)
  HIPCHK(c, all_pi.reserve((size_t)n_pairs));
  HIPCHK(c, all_pj.reserve((size_t)n_pairs));
  auto body = [&]() -> int {
    HIPCHK(c, hipMemcpyAsync(all_pi.p, c->d_pi.p, (size_t)n_pairs * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(all_pj.p, c->d_pj.p, (size_t)n_pairs * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
    int r = ICIKT_SUCCESS;
    if (n_feat > 0) {  // H2D of the matrix (staged like every other matrix upload); none of its own columns is sorted, only the masked pair columns are
      r = icikt::host::upload_and_prepare(c, X, n_feat, n_samp, ld, 0, n_samp, 0u, false, nullptr, icikt::host::kPrepassNone);
      if (r) return r;
    }
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(n_pairs, ((int64_t)3 << 29) / std::max<int64_t>(16 * n_feat, 16)));
    HIPCHK(c, c->d_Xp.reserve((size_t)std::max<int64_t>(2 * chunk * n_feat, 1)));
    HIPCHK(c, c->d_out4.reserve((size_t)chunk * 4));
    if (counts) HIPCHK(c, c->d_counts.reserve((size_t)chunk * ICIKT_CNT_FIELDS));
    if (reasons) HIPCHK(c, c->d_reasons.reserve((size_t)chunk));
    std::vector<int32_t> qi, qj;
    for (int64_t first = 0; first < n_pairs; first += chunk) {
      const int64_t m = std::min(chunk, n_pairs - first);
      HIPCHK(c, icikt::launch_mask_pairs(c->d_X.p, n_feat, (int)n_feat, all_pi.p, all_pj.p, first, m, c->d_Xp.p, c->stream));
      r = icikt_prepare_dev(c, c->d_Xp.p, n_feat, 2 * m, n_feat, flags);
      if (r) return r;
      if ((int64_t)qi.size() != m) {
        qi.resize((size_t)m); qj.resize((size_t)m);
        for (int64_t k = 0; k < m; ++k) { qi[(size_t)k] = (int32_t)(2 * k); qj[(size_t)k] = (int32_t)(2 * k + 1); }
        r = icikt_set_pairs(c, qi.data(), qj.data(), m);
        if (r) return r;
      }
      r = icikt_run_dev(c, ICIKT_PERSPECTIVE_LOCAL, alternative, continuity, flags, c->d_out4.p,
                        counts ? c->d_counts.p : nullptr, reasons ? c->d_reasons.p : nullptr);
      if (r) return r;
      r = icikt::host::download(c, out4 + 4 * first, c->d_out4.p, (size_t)m * 4 * sizeof(double));
      if (!r && counts) r = icikt::host::download(c, counts + ICIKT_CNT_FIELDS * first, c->d_counts.p, (size_t)m * ICIKT_CNT_FIELDS * sizeof(int64_t));
      if (!r && reasons) r = icikt::host::download(c, reasons + first, c->d_reasons.p, (size_t)m * sizeof(int32_t));
      if (r) return r;
      {  // the chunk's buffers are reused by the next one
        const hipError_t ec = hipStreamSynchronize(c->stream);
        icikt::host::finish_downloads(c, ec == hipSuccess);
        HIPCHK(c, ec);
      }
    }
    return ICIKT_SUCCESS;
  };
  rc = body();
  const hipError_t es = hipStreamSynchronize(c->stream);
  icikt::host::finish_downloads(c, rc == 0 && es == hipSuccess);
  all_pi.release();
  all_pj.release();
  if (rc) return rc;
  if (es != hipSuccess) return fail(c, ICIKT_E_HIP, std::string("pairs_complete: ") + hipGetErrorString(es));
  return ICIKT_SUCCESS;
}

int icikt_pair_f64(icikt_ctx* c, const double* x, const double* y, int64_t n, int perspective, int alternative,
                   int continuity, uint32_t flags, double* out4, int64_t* counts, int32_t* reason) {
  if (!c) return ICIKT_E_INVALID;
  if (n < 0 || (n > 0 && (!x || !y))) return fail(c, ICIKT_E_INVALID, "pair: bad vectors");
  if (n > ICIKT_MAX_FEATURES_WIDE) return fail(c, ICIKT_E_TOO_LONG, std::string("pair: ") + kTooLong);
  std::vector<double> xy;
  try {
    xy.resize((size_t)std::max<int64_t>(2 * n, 1));
  } catch (const std::bad_alloc&) {
    return fail(c, ICIKT_E_NOMEM, "pair: host allocation failed");
  }
  if (n > 0) {
    memcpy(xy.data(), x, (size_t)n * sizeof(double));
    memcpy(xy.data() + n, y, (size_t)n * sizeof(double));
  }
  const int32_t pi = 0, pj = 1;
  // (the two vectors travel in a vector of this function's: whatever the caller says about ITS memory does not hold for it)
  return icikt_pairs_f64(c, xy.data(), n, 2, n, &pi, &pj, 1, perspective, alternative, continuity,
                         flags & ~ICIKT_FLAG_HOST_PINNED, out4, counts, reason);
}

int icikt_missingness_f64(icikt_ctx* c, const double* X, int64_t n_feat, int64_t n_samp, int64_t ld,
                          const int32_t* pi, const int32_t* pj, int64_t n_pairs, int64_t* missingness) {
  if (!c) return ICIKT_E_INVALID;
  int rc = check_shape(c, "missingness", n_feat, n_samp, ld);
  if (rc) return rc;
  if (n_pairs < 0) return fail(c, ICIKT_E_INVALID, "missingness: bad shape");
  if (n_pairs == 0) return ICIKT_SUCCESS;
  if (!pi || !pj || !missingness) return fail(c, ICIKT_E_INVALID, "missingness: null argument");
  if (n_feat > 0 && n_samp > 0 && !X) return fail(c, ICIKT_E_INVALID, "missingness: null matrix");
  rc = check_pair_list(c, "missingness", pi, pj, n_pairs, n_samp);
  if (rc) return rc;
  rc = use_device(c);
  if (rc) return rc;
  rc = icikt_set_pairs(c, pi, pj, n_pairs);
  if (rc) return rc;
  // The mask-only pre-pass (k0_mask): one streaming pass over each column chunk as it arrives, no sort -- the
  // reference's missing_either (R/kendalltau.R:626-629) is sum(in_x | in_y) and nothing more.
  rc = icikt::host::mask_alloc(c, n_feat, n_samp);
  if (rc) return rc;
  if (n_feat > 0) {
    rc = icikt::host::upload_and_prepare(c, X, n_feat, n_samp, ld, 0, n_samp, 0u, false, nullptr, icikt::host::kPrepassMask);
    if (rc) return rc;
  }
  auto body = [&]() -> int {
    HIPCHK(c, c->d_counts.reserve((size_t)n_pairs));
    HIPCHK(c, icikt::launch_missingness(c->pv, c->d_pi.p, c->d_pj.p, n_pairs, c->d_counts.p, c->stream));
    return icikt::host::download(c, missingness, c->d_counts.p, (size_t)n_pairs * sizeof(int64_t));
  };
  rc = body();
  const hipError_t es = hipStreamSynchronize(c->stream);
  icikt::host::finish_downloads(c, rc == 0 && es == hipSuccess);
  if (rc) return rc;
  if (es != hipSuccess) return fail(c, ICIKT_E_HIP, std::string("missingness: ") + hipGetErrorString(es));
  return ICIKT_SUCCESS;
}

// Development / test hook: "key=value,key=value" overrides of the pair kernel's launch plan and of the host
// path's H2D mode; NULL or "" restores the library's choices.  Keys: np (pairs per wave: 1 | 2), pend (l | g),
// wpb (waves per workgroup), half (0 | 1), tgmax (list / count mode up to this many tie groups of the gathered column; -1 =
// row mode), list (list mode up to this many, <= 128), solo (0: no SOLO steps), waves (waves per CU the counter tables may
// cost the launch down to), split (1 | 2 | 4 segments per half-wave task), gridmult (persistent grid as a multiple of the resident waves, always), gridcap (persistent
// grid: at most this many workgroups),
// pipe (0 | 1: the host entries' chunk pipeline off / on whenever possible), k0 (0 | 1: the pre-pass always in its
// 1 024-thread / 256-thread shape), verbose (0 | 1: print the plan to stderr).
// (The keys h2d and regfail of rounds 2-3 are gone with the mode they steered: the library no longer page-locks
// caller memory, icikt_host.h.)
int icikt_debug_set_plan(icikt_ctx* c, const char* spec) {
  if (!c) return ICIKT_E_INVALID;
  icikt_ctx::PlanOverride ov;
  int pipe = -1, k0 = -1;
  std::string sp = spec ? spec : "";
  size_t pos = 0;
  while (pos < sp.size()) {
    size_t end = sp.find(',', pos);
    if (end == std::string::npos) end = sp.size();
    const std::string item = sp.substr(pos, end - pos);
    pos = end + 1;
    if (item.empty()) continue;
    const size_t eq = item.find('=');
    if (eq == std::string::npos) return fail(c, ICIKT_E_INVALID, "debug_set_plan: expected key=value, got '" + item + "'");
    const std::string key = item.substr(0, eq), val = item.substr(eq + 1);
    if (val.empty()) continue;  // "key=" keeps the default
    if (key == "np") ov.np = atoi(val.c_str());
    else if (key == "pend") ov.pend = (val[0] == 'g') ? 1 : 0;
    else if (key == "wpb") ov.wpb = atoi(val.c_str());
    else if (key == "half") ov.half = (val[0] == '1') ? 1 : 0;
    else if (key == "hyb") ov.hyb = (val[0] == '1') ? 1 : 0;
    else if (key == "tgmax") { ov.has_tgmax = true; ov.tgmax = atoi(val.c_str()); }
    else if (key == "list") ov.list = atoi(val.c_str());
    else if (key == "solo") ov.solo = atoi(val.c_str());
    else if (key == "split") ov.split = atoi(val.c_str());
    else if (key == "merge") ov.merge = atoi(val.c_str());
    else if (key == "waves") ov.waves = atoi(val.c_str());
    else if (key == "verbose") ov.verbose = (val[0] == '1');
    else if (key == "gridmult") ov.grid_mult = atoi(val.c_str());
    else if (key == "gridcap") ov.grid_cap = atoi(val.c_str());
    else if (key == "pipe") pipe = (val[0] == '1') ? 1 : 0;
    else if (key == "k0") k0 = (val[0] == '1') ? 1 : 0;
    else return fail(c, ICIKT_E_INVALID, "debug_set_plan: unknown key '" + key + "'");
  }
  c->plan_ov = ov;
  c->pipe_mode = pipe;
  c->k0_shape = k0;

  c->raw_valid = false;
  c->wpb = 0;  // tasks are rebuilt for the new plan
  return ICIKT_SUCCESS;
}

// Development hook: per step kind of the pair kernel [steps x 8 | rows x 8 | wave cycles x 8] since the last reset.
// Only a diagnostic build (-DICIKT_STEP_STATS) counts; the product build answers ICIKT_E_STATE.
int icikt_debug_step_stats(icikt_ctx* c, uint64_t* out24, int reset) {
  if (!c || !out24) return ICIKT_E_INVALID;
  int rc = use_device(c);
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  unsigned long long tmp[24];
  const hipError_t e = icikt::read_step_stats(tmp, reset);
  if (e == hipErrorNotSupported) { (void)hipGetLastError(); return fail(c, ICIKT_E_STATE, "debug_step_stats: not a -DICIKT_STEP_STATS build"); }
  HIPCHK(c, e);
  for (int i = 0; i < 24; ++i) out24[i] = tmp[i];
  return ICIKT_SUCCESS;
}

int icikt_selftest(icikt_ctx* c) {
  if (!c) return ICIKT_E_INVALID;
  int rc = use_device(c);
  if (rc) return rc;
  HIPCHK(c, c->d_self.reserve(640));
  HIPCHK(c, icikt::launch_selftest(c->d_self.p, c->stream));
  uint32_t h[640];
  HIPCHK(c, hipMemcpyAsync(h, c->d_self.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (uint32_t l = 0; l < 64; ++l) {
    if (h[l] != (l + 1) * (l + 2) / 2) return fail(c, ICIKT_E_HIP, "selftest: wave_incl_scan mismatch");
    if (h[64 + l] != (l == 0 ? 0xABCDu : 3u * (l - 1))) return fail(c, ICIKT_E_HIP, "selftest: wave_shr1 mismatch");
    if (h[128 + l] != (l < 41 ? 0xFFFFFFFFu : l - 41)) return fail(c, ICIKT_E_HIP, "selftest: repeated wave_shr1 mismatch");
    if (h[256 + l] != (l < 32 ? l : 100u + (l - 32))) return fail(c, ICIKT_E_HIP, "selftest: permlane32_swap[0] mismatch");
    if (h[320 + l] != (l < 32 ? 32u + l : 100u + l)) return fail(c, ICIKT_E_HIP, "selftest: permlane32_swap[1] mismatch");
    const uint32_t k = (l & 31u) + 1, b = (l & 32u) + 1;  // sum of (lane+1) over my half up to me
    if (h[384 + l] != k * (2 * b + k - 1) / 2) return fail(c, ICIKT_E_HIP, "selftest: half_incl_scan mismatch");
    if (h[448 + l] != 63u) {
      if (c->plan_ov.verbose) fprintf(stderr, "[icikt] selftest lane %u: lane_xor pass mask %u\n", l, h[448 + l]);
      return fail(c, ICIKT_E_HIP, "selftest: lane_xor mismatch");
    }
  }
  // half-wave all-pairs: per 32-lane half, #{(a, j): a before j, q_a < lo_j}; the lanes' shares are summed
  for (uint32_t half = 0; half < 2; ++half) {
    uint32_t want = 0, got = 0;
    for (uint32_t l = half * 32; l < half * 32 + 32; ++l) {
      const uint32_t lo = ((l * 40503u + 977u) >> 3) & 0xFFFu;
      for (uint32_t j = half * 32; j < l; ++j) want += (((j * 2654435761u >> 20) & 0xFFFu) < lo) ? 1u : 0u;
      got += h[192 + l];
    }
    if (got != want) {
      if (c->plan_ov.verbose) fprintf(stderr, "[icikt] selftest half %u: got %u want %u\n", half, got, want);
      return fail(c, ICIKT_E_HIP, "selftest: half_allpairs mismatch");
    }
  }
  // ... and the packed form: two sub-steps' rows at once, the counts of both summed
  for (uint32_t half = 0; half < 2; ++half) {
    uint32_t want = 0, got = 0;
    for (uint32_t l = half * 32; l < half * 32 + 32; ++l) {
      const uint32_t lo = ((l * 40503u + 977u) >> 3) & 0xFFFu, lo2 = ((l * 69069u + 12345u) >> 5) & 0x27BFu;
      for (uint32_t j = half * 32; j < l; ++j) {
        want += (((j * 2654435761u >> 20) & 0xFFFu) < lo) ? 1u : 0u;
        want += (((j * 1103515245u >> 17) & 0x27BFu) < lo2) ? 1u : 0u;
      }
      got += h[576 + l];
    }
    if (got != want) {
      if (c->plan_ov.verbose) fprintf(stderr, "[icikt] selftest half %u (packed): got %u want %u\n", half, got, want);
      return fail(c, ICIKT_E_HIP, "selftest: half_step_count mismatch");
    }
  }
  return ICIKT_SUCCESS;
}

}  // extern "C"
