// icikt_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the ICI-Kendall-tau pair engine.
//
// What is computed is what the reference's ici_kt() computes (src/kendallc.cpp:166-366); how it is
// computed is re-designed for a 64-wide wavefront with LDS-resident bitsets -- see DESIGN.md.
//
//   K0  k0_prepare   one workgroup per column: NA bitset, fill = min - 0.1, stable sort, tie groups,
//                    tie sums.  Replaces the two std::stable_sort calls per PAIR (kendallc.cpp:247,254)
//                    by one sort per COLUMN.
//   K1  k1_pairs     one column pair per wavefront: strict-discordance count and joint-tie count.
//                    Replaces kendall_discordant's Fenwick tree (:69-100) and compare_both (:33-51).
//   K2  k2_epilogue  one pair per lane: tau, tau_max, completeness, variance, z, p (:280-335), with
//                    perspective = "local" DERIVED from the global counts.
//
// No MFMA: the work is integer compare / popcount / prefix-sum.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "icikt_device.h"

namespace icikt {

// ------------------------------------------------------------------------------------------------
// wavefront primitives (wave64, DPP)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// whole-wave shift right by one lane; lane 0 keeps `old`
__device__ __forceinline__ uint32_t dpp_wave_shr1(uint32_t old, uint32_t src) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
}

// inclusive prefix sum over the 64 lanes: row_shr 1/2/4/8 inside each row of 16, then row_bcast 15 / 31
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142 /*row_bcast:15*/, 0xa, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143 /*row_bcast:31*/, 0xc, 0xf, false);
  return v;
}

// Whole-wave reductions as a butterfly over ds_swizzle (lane ^ 1 .. 16: the pattern is an immediate) and
// v_permlane32_swap (lane ^ 32).  (__shfl_xor goes through ds_bpermute with one address register per distance;
// hipcc keeps those five registers alive from the first reduction of a task to the last -- across the hot loop --
// and spills them.  DPP scans need no addresses either, but cost three times the vector instructions.)
template <int X>
__device__ __forceinline__ uint32_t swz_xor(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (X << 10) | 0x1F);
}
__device__ __forceinline__ uint32_t xor32(uint32_t v) {
  const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);   // [0]: lanes 0..31 twice, [1]: lanes 32..63 twice
  return r[0] ^ r[1] ^ v;                                                // the other half's value
}
template <int X>
__device__ __forceinline__ unsigned long long swz_xor64(unsigned long long v) {
  return (unsigned long long)swz_xor<X>((uint32_t)v) | ((unsigned long long)swz_xor<X>((uint32_t)(v >> 32)) << 32);
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
  v += swz_xor64<1>(v);
  v += swz_xor64<2>(v);
  v += swz_xor64<4>(v);
  v += swz_xor64<8>(v);
  v += swz_xor64<16>(v);
  v += (unsigned long long)xor32((uint32_t)v) | ((unsigned long long)xor32((uint32_t)(v >> 32)) << 32);
  return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
  v = max(v, (int)swz_xor<1>((uint32_t)v));
  v = max(v, (int)swz_xor<2>((uint32_t)v));
  v = max(v, (int)swz_xor<4>((uint32_t)v));
  v = max(v, (int)swz_xor<8>((uint32_t)v));
  v = max(v, (int)swz_xor<16>((uint32_t)v));
  v = max(v, (int)xor32((uint32_t)v));
  return v;
}

// Orders this wave's LDS traffic for the compiler: lanes of one wave exchange data through LDS
// (atomic OR by one lane, read by another).  The hardware keeps one wave's DS operations in order;
// this keeps the compiler from moving accesses across the hand-off.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// a wave-uniform 64-bit value moved to scalar registers (readfirstlane returns a SIGNED int:
// widen through uint32_t, or bit 31 smears into the upper word)
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return (unsigned long long)lo | ((unsigned long long)hi << 32);
}

// global loads addressed as uniform base + 32-bit per-lane offset (saddr form: no 64-bit VALU address math)
__device__ __forceinline__ uint32_t gload_u32(const uint32_t* base, uint32_t idx) {
  return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(base) + (size_t)(idx << 2));
}
__device__ __forceinline__ uint32_t gload_u16(const uint16_t* base, uint32_t idx) {
  return *reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(base) + (size_t)(idx << 1));
}

// both columns of a rec block for the lane's row: [row][2] u32, one 8-byte gather
__device__ __forceinline__ uint2 gload_rec2(const uint32_t* blk, uint32_t row) {
  return *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(blk) + (size_t)(row << 3));
}

// set bit `pos` of an LDS bitset: a 32-bit LDS atomic on the half of the 64-bit word that holds the bit
// (little endian: word w = dwords 2w, 2w+1), half the data of a 64-bit one
__device__ __forceinline__ void seen_insert(unsigned long long* bits, uint32_t pos) {
  atomicOr(reinterpret_cast<uint32_t*>(bits) + (pos >> 5), 1u << (pos & 31u));
}

// popcount(x) + acc in the one instruction that does both (hipcc splits chains of these into popcounts and adds)
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {
  uint32_t r;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
  return r;
}
// (hi << 16) | lo in one instruction (hipcc emits a shift and an OR when it can prove the operands disjoint)
__device__ __forceinline__ uint32_t pack16(uint32_t hi, uint32_t lo) {
  uint32_t r;
  asm("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(r) : "v"(hi), "v"(lo));
  return r;
}
__device__ __forceinline__ uint32_t bcnt64_acc(unsigned long long x, uint32_t acc) {
  return bcnt_acc((uint32_t)(x >> 32), bcnt_acc((uint32_t)x, acc));
}
__device__ __forceinline__ unsigned long long low_mask64(uint32_t bits /*0..63*/) {
  return (1ull << bits) - 1ull;
}

// ------------------------------------------------------------------------------------------------
// K0: per-column pre-pass
// ------------------------------------------------------------------------------------------------
// Two shapes of the pre-pass kernel, the same code: 1 024 threads x 4 elements each (16 waves: the fastest way through
// ONE column, and what the library runs everywhere), and 256 threads x 16 elements (4 waves, one per SIMD, 96 VGPRs: a
// workgroup that fits a CU as soon as ONE workgroup of a running pair kernel retires, where the 1 024-thread shape needs
// all four SIMDs' registers, i.e. an EMPTY CU).  The small shape was built in round 4 to let the later chunks of the
// pipelined host path sort beside the pair kernel; measured, that overlap makes the call longer (both kernels are bound
// by vector issue; icikt_capi.cpp, upload_and_prepare), so it is kept as an option of the debug plan (k0=1) and a test.
constexpr int K0_THREADS = 1024;   // the large shape (and what the shared scratch arrays are sized for)
constexpr int K0_THREADS_SMALL = 256;
#ifndef ICIKT_K0_MIN_WAVES
#define ICIKT_K0_MIN_WAVES 4   // waves per SIMD the pre-pass is compiled for: 4 = one 1 024-thread workgroup per CU (<= 128 VGPRs)
#endif
#ifndef ICIKT_K0_WAVES_SMALL
#define ICIKT_K0_WAVES_SMALL 5    // the small shape is compiled for five waves per SIMD (<= 96 registers): a wave of it fits a SIMD on which ONE pair-kernel wave (80 of 512 registers, six resident) has retired
#endif
constexpr int K0_UB = 8;       // iterations of a column pass whose loads a thread issues together
constexpr int K0_TILE = 4096;  // elements of the LDS-resident sort tile (48 KB) = threads x elements per thread of the standard shapes


__device__ __forceinline__ unsigned long long sortable_key(double v) {
  if (v == 0.0) v = 0.0;  // -0.0 and +0.0 tie (x[i] < x[j] is false both ways, kendallc.cpp:9,23)
  long long b = __double_as_longlong(v);
  unsigned long long u = (unsigned long long)b;
  return (b < 0) ? ~u : (u | 0x8000000000000000ull);
}

// Block-wide reductions: inside a wave by shuffles, across the waves through one small LDS table -- one or two workgroup
// barriers per BATCH of values.  (Rounds 1-3 ran a 1 024-entry LDS tree per value: 12 barriers each, 14 values per
// column: 170 of a column's ~450 barriers.  Removing them changed nothing: the pre-pass is bound by the vector
// instructions of its sort network, see kv_gt.)
template <typename T, typename Op>
__device__ __forceinline__ T wave_reduce(T v, Op op) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = op(v, __shfl_xor(v, o, 64));
  return v;
}
// value of lane ^ J for J = 1, 2, 4, 8, 16, 32 without LDS: quad_perm for 1 and 2, row_shl / row_shr 4 with a
// select, row_ror:8 (inside a 16-lane row rotating by 8 IS xor 8), v_permlane16_swap / v_permlane32_swap of two
// copies for 16 and 32.  Every DPP runs with all lanes active; the selects come afterwards.
template <int J>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v, uint32_t lane) {
  if (J == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1 /*quad_perm:[1,0,3,2]*/, 0xf, 0xf, false);
  if (J == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E /*quad_perm:[2,3,0,1]*/, 0xf, 0xf, false);
  if (J == 4) {
    const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x104 /*row_shl:4*/, 0xf, 0xf, false);
    const uint32_t dn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114 /*row_shr:4*/, 0xf, 0xf, false);
    return (lane & 4u) ? dn : up;
  }
  if (J == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128 /*row_ror:8*/, 0xf, 0xf, false);
  if (J == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);  // r[0] = rows [0,0,2,2], r[1] = rows [1,1,3,3]
    return (lane & 16u) ? r[0] : r[1];
  }
  const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);    // r[0] = halves [lo, lo], r[1] = [hi, hi]
  return (lane & 32u) ? r[0] : r[1];
}

// ---- bitonic stages held in registers ---------------------------------------------------------------
// Thread t owns the E consecutive elements E t .. E t + E - 1 of the tile (E = 4 with 1 024 threads, 16 with 256):
// compare-exchange distances below E stay inside the thread, distances E .. 32 E pair it with lane t ^ (j / E) of its own
// wave (DPP and permlane exchanges: no LDS, no barrier); only distances >= 64 E need the LDS tile and a workgroup barrier.
// FAST: the element is ONE word, top 48 bits of the sortable key | row (16 bits) -- unique, so a single 64-bit compare
// orders it and no index travels beside it: five vector instructions per element of a lane stage instead of nine, two
// registers instead of three, 8 bytes per element through the LDS tile and the scratch instead of 12.  Exact whenever no
// two values of the column differ ONLY in the low 16 bits of their keys (relative difference below 2^-36): k0_prepare
// checks the result against the full keys and repeats the column with the three-word elements if it finds an inversion.
template <bool FAST>
__device__ __forceinline__ bool kv_gt(unsigned long long ka, uint32_t ia, unsigned long long kb, uint32_t ib) {
  if constexpr (FAST) return ka > kb;
  return (ka > kb) || (ka == kb && ia > ib);
}
// stages j = jmax .. 1 (jmax <= 32 E) of merge step k; gi = global index of the thread's first element
template <bool FAST, int E>
__device__ __forceinline__ void k0_reg_stages(unsigned long long (&ek)[E], uint32_t (&ei)[E], int k, int jmax,
                                              int gi, int tid) {
  const uint32_t lane = (uint32_t)tid & 63u;
  const bool upw = (gi & k) == 0;  // lane stages run for k >= 2 E: the direction is the same for the thread's elements
  // one stage: partner lane = lane ^ LX (element distance E * LX), exchanged in registers (lane_xor: DPP / permlane)
#define ICIKT_K0_XSTAGE(LX)                                                                              \
  if (jmax >= E * (LX)) {                                                                                \
    const bool want_gt = (((tid & (LX)) == 0) == upw); /* take the partner's when (mine > theirs) == want_gt */ \
    _Pragma("unroll") for (int r = 0; r < E; ++r) {                                                      \
      const uint32_t klo = lane_xor<(LX)>((uint32_t)ek[r], lane), khi = lane_xor<(LX)>((uint32_t)(ek[r] >> 32), lane); \
      const uint32_t oi = FAST ? 0u : lane_xor<(LX)>(ei[r], lane);                                       \
      const unsigned long long ok = (unsigned long long)klo | ((unsigned long long)khi << 32);           \
      if (kv_gt<FAST>(ek[r], ei[r], ok, oi) == want_gt) { ek[r] = ok; ei[r] = oi; }                      \
    }                                                                                                    \
  }
  ICIKT_K0_XSTAGE(32) ICIKT_K0_XSTAGE(16) ICIKT_K0_XSTAGE(8) ICIKT_K0_XSTAGE(4) ICIKT_K0_XSTAGE(2) ICIKT_K0_XSTAGE(1)
#undef ICIKT_K0_XSTAGE
  // inside the thread: distances E / 2 .. 1; the direction of a pair follows its lower element (bit k of gi + a: for
  // k >= E that of gi, below it the bit of a itself)
#pragma unroll
  for (int j = E / 2; j >= 1; j >>= 1) {
    if (jmax >= j) {
#pragma unroll
      for (int a = 0; a < E; ++a) {
        if ((a & j) == 0) {
          const int b = a | j;
          const bool up = ((gi + a) & k) == 0;
          if (kv_gt<FAST>(ek[a], ei[a], ek[b], ei[b]) == up) {
            const unsigned long long tk = ek[a]; ek[a] = ek[b]; ek[b] = tk;
            const uint32_t ti = ei[a]; ei[a] = ei[b]; ei[b] = ti;
          }
        }
      }
    }
  }
}

// Compare-exchange stages on an array (the LDS tile, or the column's global scratch), TWO stages per pass and barrier: a
// thread loads the four elements i0, i0 + j/2, i0 + j, i0 + 3j/2, runs stage j (pairs at distance j) and stage j/2 in
// registers and stores them back -- half the barriers and half the traffic of one stage per pass.  Stages jmax .. jmin
// (powers of two) of merge step k over `count` elements; dbase = what is added to an index to find its direction bit.
template <bool FAST, int NT, typename KP, typename IP>
__device__ __forceinline__ void k0_mem_stages(KP tk, IP ti, int count, int dbase, int k, int jmax, int jmin, int tid) {
  int j = jmax;
  while (j >= 2 * jmin) {
    const int jh = j >> 1;
    for (int t = tid; t < (count >> 2); t += NT) {
      const int i0 = ((t & ~(jh - 1)) << 2) | (t & (jh - 1));
      const bool up = ((dbase + i0) & k) == 0;      // the same for the four: bits j/2 and j of i0 are clear, j + j/2 < k
      unsigned long long ek[4];
      uint32_t ei[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { ek[r] = tk[i0 + r * jh]; ei[r] = FAST ? 0u : ti[i0 + r * jh]; }
#define ICIKT_CE4(a, b)                                                              \
      if (kv_gt<FAST>(ek[a], ei[a], ek[b], ei[b]) == up) {                           \
        const unsigned long long xk = ek[a]; ek[a] = ek[b]; ek[b] = xk;              \
        const uint32_t xi = ei[a]; ei[a] = ei[b]; ei[b] = xi;                        \
      }
      ICIKT_CE4(0, 2) ICIKT_CE4(1, 3) ICIKT_CE4(0, 1) ICIKT_CE4(2, 3)
#undef ICIKT_CE4
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        tk[i0 + r * jh] = ek[r];
        if (!FAST) ti[i0 + r * jh] = ei[r];
      }
    }
    __syncthreads();
    j >>= 2;
  }
  if (j >= jmin) {   // an odd number of stages: the last one alone
    for (int t = tid; t < (count >> 1); t += NT) {
      const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
      const int l = i | j;
      const bool up = ((dbase + i) & k) == 0;
      const unsigned long long ka = tk[i], kb = tk[l];
      const uint32_t ia = FAST ? 0u : ti[i], ib = FAST ? 0u : ti[l];
      if (kv_gt<FAST>(ka, ia, kb, ib) == up) {
        tk[i] = kb; tk[l] = ka;
        if (!FAST) { ti[i] = ib; ti[l] = ia; }
      }
    }
    __syncthreads();
  }
}

// ---- the tie program of a column (PrepView::tprog, srow, smask) ---------------------------------------------------
// From where a column's tie groups begin, the half-wave pair kernels walk it in steps cut at group boundaries:
//   HOT    64 singleton rows;
//   MIXED  two SUB-STEPS of up to 32 rows each, every one made of COMPLETE groups of at most 32 rows: no group
//          straddles the two, so rows of one group only ever meet inside a sub-step, in registers;
//   SOLO   a MIXED step whose sub-steps hold ONE group each (k0_step_at);
//   GROUP  up to 64 rows of ONE longer group (`closes`: the step holds the group's last row).
// The cut depends on the streamed column alone, so it is made here, once per column, by one wave (the step sequence is
// scalar code: every lane computes the same values, lane 0 stores) instead of by every pair that streams the column --
// S - 1 times, on the scalar unit of the pair kernel, with the flag words fetched from memory inside its step loop.
// Every step gets its RECORD (PrepView::srow): its rows in the lane layout the pair kernel runs it in, the guard row in
// the empty lanes.  For a MIXED / SOLO step the record also says, per lane, WHICH FLAGS of the pair kernel's in-step compare
// vectors (half_step_flags: bit layout below) belong to pairs inside a tie group (PrepView::smask: the masks of the lane's
// two rows, one per sub-step, combined): the pair kernel masks them out of the discordance count and counts the joint
// ties among them, whatever the sizes of the groups.
// gf: the column's group-start flags in PROCESSING order, W words + a zero guard word (in LDS).  The program starts at
// pos0 = 64 floor(hot_until / 64), where the pair kernel's singleton loop ends, and runs to n.
//
// Flag layout of half_step_flags (lane = (pair h, l = 0..31), DPP row = l >> 4, p = l & 15; sub-step s of the lane's two
// rows in byte s ("b" slots) and byte 2 + s ("a" slots) of both vectors):
//   vector 1   a bit i = in-row distance 2i + 1, b bit i = in-row distance 2i + 2 (i <= 6); b bit 7 = rows crossed at
//              rotation 0.  In-row distance d: the lane is the LATER row, its partner sits d lanes in front.
//   vector 2   a bit 4 + i = rotation 2i + 1, b bit 4 + i = rotation 2i + 2 (i <= 2); rotation 7 = bit 15 (s = 0) / 31.
//   rotation r of an UPPER-row lane: the lane is the later row, its partner the lower row's lane (p - r) mod 16, i.e.
//   D = (p >= r ? 16 + r : r) lanes in front; of a LOWER-row lane: the lane is the EARLIER row, its partner the upper
//   row's lane (p - r - 1) mod 16, D = (p - r - 1 >= 0 ? 15 - r : 31 - r) lanes behind.
// Two rows D lanes apart belong to one group iff D <= idx of the later one (rows of its group in front of it) iff
// D <= fwd of the earlier one (rows of its group behind it).
constexpr int TPROG_KS = 32;   // == k1_ks(true): MIXED steps take groups of up to this many rows
// The cross-row part of a row's masks depends on its lane and on ONE number, lim = idx of an upper-row lane / fwd of a
// lower-row lane (<= 31): a table [32][64] in LDS, filled once per workgroup, replaces eight compares per row and step.
//   bit 7: vector 1, rotation 0; bits 4..6: vector 2, rotations 2 4 6 (b slots); bits 12..14: rotations 1 3 5 (a slots);
//   bit 16: rotation 7
__device__ inline void k0_cross_table(uint32_t* T, int tid, int nthreads) {
  for (int i = tid; i < 32 * 64; i += nthreads) {
    const uint32_t lim = (uint32_t)i >> 6, l = (uint32_t)i & 31u, p = l & 15u;
    const bool upper = (l & 16u) != 0u;
    uint32_t t = 0u;
    for (uint32_t r = 0; r < 8; ++r) {
      const uint32_t D = upper ? ((p >= r) ? 16u + r : r) : ((p >= r + 1u) ? 15u - r : 31u - r);
      if (D > lim) continue;
      if (r == 0u) t |= 0x80u;
      else if (r == 7u) t |= 0x10000u;
      else if (r & 1u) t |= 0x1000u << ((r - 1u) >> 1);
      else t |= 0x10u << ((r - 2u) >> 1);
    }
    T[i] = t;
  }
}

// One step's entry, as if a step STARTED at position pos (every position is evaluated, in parallel; the steps the walk
// really takes are then picked out by following the chain).  16 bits: the layout of PrepView::tprog entries.
__device__ __forceinline__ uint32_t k0_step_at(const unsigned long long* gf, int n, int W, int pos) {
  const int wc = pos >> 6, fb = pos & 63;
  const unsigned long long w0 = gf[min(wc, W)], w1 = gf[min(wc + 1, W)];
  unsigned long long F = fb ? ((w0 >> fb) | (w1 << (64 - fb))) : w0;
  const bool fnbit = ((w1 >> fb) & 1ull) != 0ull;          // position pos + 64 starts a group
  const int remaining = n - pos;
  if (remaining < 64) F &= (1ull << remaining) - 1ull;
  const int avail = (remaining <= 64) ? remaining : 64;
  const bool endbit = fnbit || remaining == 64;            // offset 64 starts a group / is the end of the data
  if (avail == 64 && F == ~0ull && endbit) return 64u | (TPROG_KIND_HOT << 7) | (1u << 9);
  // group boundaries of the window: the starts, and the end of the data (one marker: no boundary beyond it)
  const unsigned long long Fe = (avail < 64) ? (F | (1ull << avail)) : F;
  const unsigned long long Fr = Fe & ~1ull;
  const int next = (Fr != 0ull) ? (int)__builtin_ctzll(Fr) : (endbit ? 64 : 65);   // end of the group at pos
  if ((F & 1ull) == 0ull || next > TPROG_KS) {
    const int glim = min(remaining, 64);
    return (uint32_t)min(next, glim) | (TPROG_KIND_GROUP << 7) | ((next <= glim) ? (1u << 9) : 0u);
  }
  // MIXED.  Sub-step 0: up to the last boundary within 32 rows (there is one: the first group has at most 32 rows);
  // sub-step 1: from b0 up to the last boundary within the next 32 rows -- none: the next group is longer (or the
  // data end at b0) and the step ends at b0.  The boundary at offset 64 (b0 == 32 only) is `endbit`.
  const int b0 = 63 - (int)__builtin_clzll(Fe & 0x1FFFFFFFEull);
  unsigned long long rest = (Fe >> b0) & 0x1FFFFFFFEull;
  if (b0 == 32 && endbit) rest |= 1ull << 32;
  const int b1 = (rest != 0ull) ? (63 - (int)__builtin_clzll(rest)) : 0;
  // SOLO: a MIXED step whose sub-steps hold ONE group each (sub-step 0 = the group at pos, sub-step 1 = the next group or
  // nothing): no pair of its rows is discordant inside a sub-step, so a pair kernel whose gathered columns have counter
  // tables runs it without the in-step chains (k1_pairs: hot_step<2>); otherwise it is a MIXED step like any other.
  const bool one0 = next == b0;
  const unsigned long long in1 = (Fe >> b0) & ((b1 > 1) ? ((1ull << b1) - 2ull) : 0ull);   // starts strictly inside sub-step 1
  const bool solo = one0 && in1 == 0ull;
  return (uint32_t)(b0 + b1) | ((solo ? TPROG_KIND_SOLO : TPROG_KIND_MIXED) << 7) | (1u << 9) | ((uint32_t)b0 << 10);
}

// the same-group flag masks of the rows of the MIXED step that starts at pos (entry e): one lane per row
// (returns the class of the step's largest group, wave-uniform: half_step_flags_near)
__device__ __forceinline__ uint32_t k0_step_masks(const unsigned long long* gf, const uint32_t* crossT, int n, int W, int pos,
                                                  uint32_t e, uint2& m_out, uint32_t lane) {
  const uint32_t b0 = tprog_n0(e), b1 = tprog_rows(e) - b0;
  const uint32_t sb = lane >> 5, l = lane & 31u;
  const bool vrow = l < (sb ? b1 : b0);          // (every lane of the wave stays: the step's largest group is a wave maximum)
  const int wc = pos >> 6, fb = pos & 63;
  const unsigned long long w0 = gf[min(wc, W)], w1 = gf[min(wc + 1, W)];
  unsigned long long Fe = fb ? ((w0 >> fb) | (w1 << (64 - fb))) : w0;
  const int remaining = n - pos;
  if (remaining < 64) Fe = (Fe & ((1ull << remaining) - 1ull)) | (1ull << remaining);
  const uint32_t o = min(sb ? b0 + l : l, 63u);                                   // offset of the row in the window (< 64)
  const unsigned long long upto = Fe & ((o < 63u) ? ((2ull << o) - 1ull) : ~0ull);
  const uint32_t idx = o - (63u - (uint32_t)__builtin_clzll(upto | 1ull));        // bit 0 is set: the step starts a group
  const unsigned long long above = (o < 63u) ? (Fe >> (o + 1u)) : 0ull;
  const uint32_t nxt = (above != 0ull) ? (o + 1u + (uint32_t)__builtin_ctzll(above)) : 64u;   // (offset 64: the step ends there)
  const uint32_t fwd = nxt - 1u - o;
  // class of the step's largest group: which distances of the pair kernel's second flag chain can hold a pair of one group
  const int gmax = __builtin_amdgcn_readfirstlane(wave_max_i32(vrow ? (int)(idx + fwd + 1u) : 0));
  const uint32_t cls = (gmax <= 3) ? 0u : (gmax <= 5) ? 1u : (gmax <= 9) ? 2u : 3u;
  m_out = make_uint2(0u, 0u);
  if (!vrow) return cls;
  const uint32_t p = l & 15u;
  const uint32_t c = min(min(idx, p), 15u);                                       // in-row partners inside my group
  const uint32_t a1 = (1u << ((c + 1u) >> 1)) - 1u, b1m = (1u << (c >> 1)) - 1u;   // vector 1: a / b slots
  const uint32_t t = crossT[min((l & 16u) ? idx : fwd, 31u) * 64u + lane];
  const uint32_t sh = 8u * sb;
  uint2 m;
  m.x = ((b1m | (t & 0x80u)) << sh) | (a1 << (16u + sh));
  m.y = ((t & 0x70u) << sh) | (((t >> 8) & 0x70u) << (16u + sh)) | ((t & 0x10000u) ? (0x8000u << (16u * sb)) : 0u);
  m_out = m;
  return cls;
}

// The whole workgroup builds the program, a WINDOW of TPROG_WIN positions at a time: (A) every position's would-be step,
// in parallel, into E (u16 per position of the window, LDS); (B) one wave follows the chain from where it stands -- one
// LDS read per step instead of a hundred dependent scalar operations -- while the step STARTS inside the window, writes
// the entries and lists the steps; (C) the waves share out the window's steps and write their RECORDS (PrepView::srow,
// smask): the step's rows in the lane layout the pair kernel runs it in -- an empty lane names the guard row -- and, for
// a MIXED step, per lane l = 0..31 the same-group masks of its two rows (sub-step 0: lane l, sub-step 1: lane l + 32).
// Steps per window: a MIXED step that does not reach 33 rows is followed by a group of more than 32 rows, and a closing
// GROUP step that does not reach 33 rows follows a 64-row piece of its group: any two consecutive steps hold >= 34 rows
// -- at most TPROG_WIN / 17 + 1 steps start in a window, n / 17 + 2 in a column (PrepView::sr_steps has room for them and
// for the three guard steps behind the last one: what the pair kernel loads ahead).
// scratch: E (TPROG_WIN u16) | list of the window's steps (TPROG_LIST pairs of u16: window offset, entry index) | crossT
// (32 x 64 u32); `cnt`: four shared ints (steps of the window; the chain's position and entry count between windows; the
// column's streaming cost, see below).
constexpr int TPROG_WIN = 16384, TPROG_LIST = 1024;
// SEGMENT MARKS (round 4, last change): four positions of the walk at which a pair kernel task may be CUT when a launch has
// fewer tasks than the chip has waves -- the wave that takes the part behind a mark first inserts the rows in front of it
// without counting them (k1_pairs).  A mark is the start of a step that starts a tie group (or a multiple of 64 inside the
// singleton region); marks 1 cuts the walk in two parts of equal cost, marks 0, 2, 3 in four (inserting a row costs about
// an eighth of counting it; the closed-form tail costs next to nothing).  Written behind the program: prog_tail[0..3] =
// positions (0xFFFFFFFF: none), prog_tail[4..7] = their program steps (0xFFFFFFFF: in the singleton region).
constexpr int TPROG_MARKS = 4;
__device__ inline void k0_tie_program(const unsigned long long* gf, uint16_t* E, uint16_t* mlist, uint32_t* crossT, int* cnt,
                                      int n, int W, uint32_t* prog, uint32_t* prog_tail, const uint16_t* ord, uint16_t* srow,
                                      uint2* smask, int sr_steps, uint32_t guard_row, int tid, int nthreads) {
  const uint32_t lane = (uint32_t)tid & 63u;
  const int wave = tid >> 6, nwaves = nthreads >> 6;
  k0_cross_table(crossT, tid, nthreads);
  if (wave == 0) {
    int first_cont = n, best = 0;   // first position that continues a group; highest group start
    for (int w = (int)lane; w < W; w += 64) {
      unsigned long long f = gf[w], z = ~gf[w];
      if (w == W - 1 && (n & 63)) { f &= (1ull << (n & 63)) - 1ull; z &= (1ull << (n & 63)) - 1ull; }
      if (z != 0ull) first_cont = min(first_cont, w * 64 + (int)__builtin_ctzll(z));
      if (f != 0ull) best = max(best, w * 64 + 63 - (int)__builtin_clzll(f));
    }
    first_cont = -__builtin_amdgcn_readfirstlane(wave_max_i32(-first_cont));
    const int last_start = __builtin_amdgcn_readfirstlane(wave_max_i32(best));
    const int hot_until = (first_cont < n) ? first_cont - 1 : n;
    const int pos0 = (hot_until >> 6) << 6;
    // cnt[3]: what streaming this column costs a pair, in half hot steps (a hot step of 64 rows = 2, a MIXED step = 3, a
    // GROUP step = 3: their vector instructions): the singleton region now, the program's steps below
    if (lane == 0u) {
      cnt[1] = pos0; cnt[2] = 0; cnt[3] = 2 * (hot_until >> 6);
      // the segment marks: targets as shares of the rows in front of the last tie group (it runs in closed form when it is
      // longer than a step); those inside the singleton region are known now, the others are met by the walk below
      const int n_eff = (n - last_start > 64) ? last_start : n;
      const int share[TPROG_MARKS] = {307, 545, 578, 815};   // (in 1 / 1024: 0.300, 0.532, 0.564, 0.796 of the rows)
      for (int j = 0; j < TPROG_MARKS; ++j) {
        const int T = (int)(((uint32_t)n_eff * (uint32_t)share[j]) >> 10);   // (n <= 65 535: fits 32 bits)
        cnt[12 + j] = T;
        if (T < pos0) { cnt[4 + j] = (T >> 6) << 6; cnt[8 + j] = -1; }   // (step -1: in the singleton region)
        else { cnt[4 + j] = -1; cnt[8 + j] = -1; }                         // (position -1: not met yet)
      }
      cnt[16] = 0;   // the walk stands inside a group of several GROUP steps
    }
  }
  __syncthreads();
  for (int win = (cnt[1] / TPROG_WIN) * TPROG_WIN; win < n; win += TPROG_WIN) {
    const int wend = min(n, win + TPROG_WIN);
    for (int p = win + tid; p < wend; p += nthreads) E[p - win] = (uint16_t)k0_step_at(gf, n, W, p);
    __syncthreads();
    if (wave == 0) {
      int pos = cnt[1], ne = cnt[2], ns = 0, cost = cnt[3];
      int mpos[TPROG_MARKS], mstep[TPROG_MARKS], mT[TPROG_MARKS];
      for (int j = 0; j < TPROG_MARKS; ++j) { mpos[j] = cnt[4 + j]; mstep[j] = cnt[8 + j]; mT[j] = cnt[12 + j]; }
      bool open = cnt[16] != 0;
      int mj = 0;   // marks met so far (those inside the singleton region were set before the walk)
      while (mj < TPROG_MARKS && mpos[mj] >= 0) ++mj;
      int nextT = (mj < TPROG_MARKS) ? mT[mj] : 0x7FFFFFFF;   // the next mark's target: one compare per step of the walk
      while (pos < wend) {
        const uint32_t e = (uint32_t)E[pos - win];
        // (the targets ascend: the marks are met in turn, one compare per step)
        if (!open && pos >= nextT) {   // (rare: four times per column)
          while (mj < TPROG_MARKS && pos >= mT[mj]) { mpos[mj] = pos; mstep[mj] = ne; ++mj; }
          nextT = (mj < TPROG_MARKS) ? mT[mj] : 0x7FFFFFFF;
        }
        open = tprog_kind(e) == TPROG_KIND_GROUP && !tprog_closes(e);
        cost += (tprog_kind(e) == TPROG_KIND_HOT) ? 2 : 3;
        if (lane == 0u) {
          prog[ne] = e;
          if (ns < TPROG_LIST) { mlist[2 * ns] = (uint16_t)(pos - win); mlist[2 * ns + 1] = (uint16_t)ne; }
        }
        ++ns;
        ++ne;
        pos += (int)tprog_rows(e);
      }
      if (lane == 0u) {
        if (wend == n) prog[ne] = 0u;
        cnt[0] = min(ns, TPROG_LIST); cnt[1] = pos; cnt[2] = ne; cnt[3] = cost;
        for (int j = 0; j < TPROG_MARKS; ++j) { cnt[4 + j] = mpos[j]; cnt[8 + j] = mstep[j]; }
        cnt[16] = open ? 1 : 0;
      }
    }
    __syncthreads();
    const int ns = cnt[0];
    for (int i = wave; i < ns; i += nwaves) {
      const int pos = win + (int)mlist[2 * i];
      const int st = (int)mlist[2 * i + 1];
      if (st + 3 >= sr_steps) continue;   // (cannot happen: the bound above)
      const uint32_t e = (uint32_t)E[pos - win];
      const uint32_t rows = tprog_rows(e);
      if (tprog_kind(e) == TPROG_KIND_MIXED || tprog_kind(e) == TPROG_KIND_SOLO) {
        const uint32_t b0 = tprog_n0(e), sb = lane >> 5, l = lane & 31u;
        const bool vrow = l < (sb ? rows - b0 : b0);
        srow[st * 64 + (int)lane] = (uint16_t)(vrow ? (uint32_t)ord[(uint32_t)pos + (sb ? b0 + l : l)] : guard_row);
        uint2 m;
        const uint32_t cls = k0_step_masks(gf, crossT, n, W, pos, e, m, lane);
        m.x |= (uint32_t)__shfl_xor((int)m.x, 32, 64);
        m.y |= (uint32_t)__shfl_xor((int)m.y, 32, 64);
        if (lane < 32u) smask[st * 32 + (int)lane] = m;
        // the class of the step's largest group rides in the step's entry (bits 16..17), which the pair kernel holds two
        // steps ahead; the entry was written by wave 0 before the barrier above
        if (lane == 0u && cls != 0u) prog[st] |= cls << 16;
      } else {
        srow[st * 64 + (int)lane] = (uint16_t)((lane < rows) ? (uint32_t)ord[(uint32_t)pos + lane] : guard_row);
      }
    }
    __syncthreads();   // (E and the list are rewritten by the next window)
  }
  // three guard steps behind the last one (the pair kernel reads its rows three steps ahead)
  {
    const int ne = cnt[2];
    if (ne + 3 <= sr_steps)
      for (int i = tid; i < 192; i += nthreads) srow[ne * 64 + i] = (uint16_t)guard_row;
  }
  if (tid < 2 * TPROG_MARKS) prog_tail[tid] = (uint32_t)cnt[4 + tid];   // (-1 -> 0xFFFFFFFF)
}

// WIDE (65 535 < n): 32-bit positions in separate arrays (order32, q32, lo32, hi32), the phase-3 bitsets in global
// memory (they outgrow the static LDS), no tie-group list and no rec staging.
template <bool WIDE, int NT, int E>
__device__ __forceinline__ void k0_prepare_body(const PrepView& pv, const double* __restrict__ X, int64_t ld, int col_begin,
                                                const MaskSpec& ms, uint8_t* __restrict__ keep) {
  static_assert((NT == K0_THREADS || NT == K0_THREADS_SMALL) && (E == 4 || E == 8 || E == 16), "pre-pass shapes");
  constexpr int TILE = NT * E;                      // elements of the LDS-resident sort tile
  constexpr int NW = NT / 64;                       // waves of the workgroup
  __shared__ long long sh_ll[K0_THREADS];           // (scratch sized for either shape: a 1 024-word bitset lives here in phase 3)
  __shared__ int sh_i[K0_THREADS];
  __shared__ unsigned long long sh_bits_lds[1024];  // fill-group bitset, W <= 1024 words
  __shared__ unsigned long long sh_st_lds[1028];    // phase 1: the min reduction; phase 3: group starts, n + 1 <= 65 536 bits
  unsigned long long* const sh_bits = WIDE ? pv.k0_bits + (size_t)blockIdx.x * 2 * (size_t)(pv.Wp + 1) : sh_bits_lds;
  unsigned long long* const sh_st = WIDE ? sh_bits + (pv.Wp + 1) : sh_st_lds;
  __shared__ unsigned long long sh_sort[TILE + TILE / 2];  // 48 KB: the sort tile, later the rec staging area
  __shared__ uint16_t sh_bigpre[1024];              // phase 3: tie groups of >= 2 rows that start in the words before w
  unsigned long long* sh_tk = sh_sort;                                      // sort tile: keys
  uint32_t* sh_ti = reinterpret_cast<uint32_t*>(sh_sort + TILE);         // sort tile: row indices
  uint32_t* rec_s = reinterpret_cast<uint32_t*>(sh_sort);                   // after the sort: rec by row
  // The per-row arrays are written by ROW (scattered): through the free sort tile, then out in order, as far as it holds
  // them -- rec and (hi | tie-group index << 16) for columns of up to 3 TILE / 2 rows (12 288 with the 8 192-element tile),
  // the latter alone (two scattered 2-byte stores per row otherwise, against one 4-byte store for rec) up to 3 TILE rows
  const bool stage_hg = !WIDE && pv.n_pad <= 3 * TILE;
  const bool stage_rec = stage_hg && 2 * pv.n_pad <= 3 * TILE;
  uint32_t* hg_s = stage_rec ? rec_s + pv.n_pad : rec_s;
  // girow: every column the half-wave pair kernels can run on; longer columns when they hold more than K1_CNT_MIN_GROUPS tie
  // groups (the whole-wave kernels' count mode asks for nothing less: decided per column below, once its groups are counted)
  bool want_gi = pv.tp_stride > 0;

  const int c = col_begin + blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int n = pv.n;
  const int W = pv.W;
  const int npow2 = pv.npow2;
  const double* col = X + (int64_t)c * ld;
  unsigned long long* keys = pv.sort_keys + (int64_t)blockIdx.x * npow2;
  uint32_t* idx = pv.sort_idx + (int64_t)blockIdx.x * npow2;
  unsigned long long* mask = pv.col_mask(c);
  unsigned long long* fmask = pv.col_fillmask(c);
  unsigned long long* gflag = pv.col_gflag(c);
  uint16_t* order = WIDE ? nullptr : pv.order + (int64_t)c * pv.n_ord;
  uint32_t* rec = WIDE ? nullptr : pv.rec + ((int64_t)(c >> 1) * pv.rec_rows) * 2 + (c & 1);  // [block][row][2]: stride 2
  uint16_t* hirow = WIDE ? nullptr : pv.hirow + ((int64_t)(c >> 1) * pv.rec_rows) * 2 + (c & 1);            // [block][row][2]: stride 2
  uint16_t* girow = WIDE ? nullptr : pv.girow + ((int64_t)(c >> 1) * pv.rec_rows) * 2 + (c & 1);            // likewise
  if (!WIDE && tid == 0) {   // the guard row (PrepView::rec_rows)
    rec[2 * pv.n_pad] = (uint32_t)pv.n_pad; hirow[2 * pv.n_pad] = 0; girow[2 * pv.n_pad] = GIROW_NONE;
  }
  uint32_t* tgl = WIDE ? nullptr : pv.tgroups + (int64_t)c * pv.tg_stride;
  uint32_t* order32 = WIDE ? pv.order32 + (int64_t)c * pv.n_pad : nullptr;
  uint32_t* q32 = WIDE ? pv.q32 + (int64_t)c * pv.n_pad : nullptr;
  uint32_t* lo32 = WIDE ? pv.lo32 + (int64_t)c * pv.n_pad : nullptr;
  uint32_t* hi32 = WIDE ? pv.hi32 + (int64_t)c * pv.n_pad : nullptr;

  // ---- phase 1: NA bitset, NA count, min of the non-missing values (kendallc.cpp:187-218) --------
  // The caller's global_na rule (setup_missing_matrix, R/utils.R:1-23) is applied HERE, while the column is read:
  // an excluded cell is missing (R/kendalltau.R:119-121), and so is every NaN (Rcpp is_na, kendallc.cpp:181).
  double tmin = __longlong_as_double(0x7FF0000000000000ll);  // +Inf
  int nna = 0, nexcl = 0;
  uint8_t* keep_c = keep ? keep + (int64_t)c * n : nullptr;
  // (The passes over the column are BLOCKED: a thread issues the loads of K0_UB iterations, then consumes them.  One
  //  workgroup per CU hides no latency by itself, and with a ballot in the loop body the compiler keeps one load in
  //  flight per thread: a pass then cost one memory round trip per iteration -- ten for 10 000 rows -- and the passes
  //  outside the sort were half of the kernel's time.)
  for (int base0 = 0; base0 < pv.n_pad; base0 += NT * K0_UB) {
    double vv[K0_UB];
#pragma unroll
    for (int u = 0; u < K0_UB; ++u) {
      const int i = base0 + u * NT + tid;
      vv[u] = (i < n) ? col[i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < K0_UB; ++u) {
      if (base0 + u * NT < pv.n_pad) {   // (uniform over the workgroup: the ballot runs with every lane)
        const int i = base0 + u * NT + tid;
        const double v = vv[u];
        const bool excl = (i < n) && mask_excluded(ms, v);
        const bool isna = (i < n) && (excl || v != v);
        nexcl += excl ? 1 : 0;
        if (keep_c && i < n) keep_c[i] = excl ? 0 : 1;
        const unsigned long long b = __ballot(isna);
        if (lane == 0 && (i >> 6) < W) mask[i >> 6] = b;
        if (i < n && !isna) tmin = (v < tmin) ? v : tmin;
        nna += isna ? 1 : 0;
      }
    }
  }
  if (tid == 0) { mask[W] = 0ull; }
  // plain double min (no NaN among candidates), the missing and the excluded rows: one batch, two barriers
  {
    tmin = wave_reduce(tmin, [](double a, double b) { return (b < a) ? b : a; });
    nna = wave_reduce(nna, [](int a, int b) { return a + b; });
    nexcl = wave_reduce(nexcl, [](int a, int b) { return a + b; });
    double* sh_d = reinterpret_cast<double*>(sh_st_lds);   // (the start-flag bitset of phase 3 lives here later)
    if (lane == 0) { sh_d[tid >> 6] = tmin; sh_i[tid >> 6] = nna; sh_i[NW + (tid >> 6)] = nexcl; }
    __syncthreads();
    tmin = sh_d[0]; nna = sh_i[0]; nexcl = sh_i[NW];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      const double b = sh_d[w];
      tmin = (b < tmin) ? b : tmin;
      nna += sh_i[w];
      nexcl += sh_i[NW + w];
    }
    __syncthreads();
  }
  const double fill = tmin - 0.1;  // kendallc.cpp:214-215, double arithmetic

  // ---- phases 1b + 2: sortable keys, sort.  FAST: one word per element (top 48 key bits | row), see kv_gt ------
  auto sort_pass = [&](auto fast_tag) {
  constexpr bool FAST = decltype(fast_tag)::value;
  for (int base0 = 0; base0 < npow2; base0 += NT * K0_UB) {
    double vv[K0_UB];
#pragma unroll
    for (int u = 0; u < K0_UB; ++u) {
      const int i = base0 + u * NT + tid;
      vv[u] = (i < n) ? col[i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < K0_UB; ++u) {
      const int i = base0 + u * NT + tid;
      if (i < npow2) {
        unsigned long long k = ~0ull;
        uint32_t id = 0xFFFFFFFFu;
        if (i < n) {
          double v = vv[u];
          if (v != v || mask_excluded(ms, v)) v = fill;
          k = sortable_key(v);
          if (FAST) k = (k & ~0xFFFFull) | (unsigned long long)i;    // (n <= 65 535: the row fits the low 16 bits)
          id = (uint32_t)i;
        }
        keys[i] = k;
        if (!FAST) idx[i] = id;
      }
    }
  }
  __syncthreads();

  // ---- phase 2: bitonic sort of (key, row): the row index breaks ties, which makes the result the
  //      stable order std::stable_sort gives in sortedIndex (kendallc.cpp:5-12).  Compare-exchange
  //      distances below the tile size run on an LDS-resident tile; only the long distances of the
  //      last merges touch global memory. -------------------------------------------------------------
  {
    const int T = (npow2 < TILE) ? npow2 : TILE;
    const int ntiles = npow2 / T;
    if (T == TILE) {
      // full tiles: register / shuffle stages (k0_reg_stages) around the LDS stages with distance >= 256
      // (a wave holds 64 E consecutive elements: distances up to 32 E in registers, 64 E and more on the LDS tile)
      unsigned long long ek[E];
      uint32_t ei[E];
      for (int tile = 0; tile < ntiles; ++tile) {
        const int tb = tile * T;
        if (tb >= n) continue;  // a tile of padding only (equal keys) is sorted in either direction already
        const int gi = tb + E * tid;
#pragma unroll
        for (int r = 0; r < E; ++r) { ek[r] = keys[gi + r]; ei[r] = FAST ? 0u : idx[gi + r]; }
        for (int k = 2; k <= 64 * E; k <<= 1) k0_reg_stages<FAST, E>(ek, ei, k, k >> 1, gi, tid);
        for (int k = 128 * E; k <= T; k <<= 1) {
#pragma unroll
          for (int r = 0; r < E; ++r) { sh_tk[E * tid + r] = ek[r]; if (!FAST) sh_ti[E * tid + r] = ei[r]; }
          __syncthreads();
          k0_mem_stages<FAST, NT>(sh_tk, sh_ti, T, tb, k, k >> 1, 64 * E, tid);
#pragma unroll
          for (int r = 0; r < E; ++r) { ek[r] = sh_tk[E * tid + r]; ei[r] = FAST ? 0u : sh_ti[E * tid + r]; }
          __syncthreads();  // the tile is rewritten by the next step's stores
          k0_reg_stages<FAST, E>(ek, ei, k, 32 * E, gi, tid);
        }
#pragma unroll
        for (int r = 0; r < E; ++r) { keys[gi + r] = ek[r]; if (!FAST) idx[gi + r] = ei[r]; }
      }
      __syncthreads();
      // merges across tiles: distances >= T in global memory, 2048..256 on the LDS tile, the rest in registers
      for (int k = 2 * T; k <= npow2; k <<= 1) {
        k0_mem_stages<FAST, NT>(keys, idx, npow2, 0, k, k >> 1, T, tid);   // (global scratch: __syncthreads orders a workgroup's global accesses)
        for (int tile = 0; tile < ntiles; ++tile) {
          const int tb = tile * T;
          const int gi = tb + E * tid;
#pragma unroll
          for (int r = 0; r < E; ++r) { sh_tk[E * tid + r] = keys[gi + r]; if (!FAST) sh_ti[E * tid + r] = idx[gi + r]; }
          __syncthreads();
          k0_mem_stages<FAST, NT>(sh_tk, sh_ti, T, tb, k, T >> 1, 64 * E, tid);   // (the direction is constant inside a tile: k > T)
#pragma unroll
          for (int r = 0; r < E; ++r) { ek[r] = sh_tk[E * tid + r]; ei[r] = FAST ? 0u : sh_ti[E * tid + r]; }
          __syncthreads();
          k0_reg_stages<FAST, E>(ek, ei, k, 32 * E, gi, tid);
#pragma unroll
          for (int r = 0; r < E; ++r) { keys[gi + r] = ek[r]; if (!FAST) idx[gi + r] = ei[r]; }
        }
        __syncthreads();
      }
    } else {
      // short columns (npow2 < 4096): one partial tile, every stage on LDS
      for (int i = tid; i < T; i += NT) { sh_tk[i] = keys[i]; if (!FAST) sh_ti[i] = idx[i]; }
      __syncthreads();
      for (int k = 2; k <= T; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
          for (int t = tid; t < (T >> 1); t += NT) {
            const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
            const int l = i | j;
            const bool up = ((i & k) == 0);
            const unsigned long long ka = sh_tk[i], kb = sh_tk[l];
            const uint32_t ia = FAST ? 0u : sh_ti[i], ib = FAST ? 0u : sh_ti[l];
            if (kv_gt<FAST>(ka, ia, kb, ib) == up) {
              sh_tk[i] = kb; sh_tk[l] = ka;
              if (!FAST) { sh_ti[i] = ib; sh_ti[l] = ia; }
            }
          }
          __syncthreads();
        }
      }
      for (int i = tid; i < T; i += NT) { keys[i] = sh_tk[i]; if (!FAST) idx[i] = sh_ti[i]; }
      __syncthreads();
    }
  }
  };   // sort_pass
  bool starts_done = false;   // the group-start bitset has been made (by the pass behind the one-word sort)
  if constexpr (WIDE) {
    sort_pass(std::false_type{});
  } else {
    // One-word elements first.  ONE pass then turns the sorted words into what phase 3 needs -- the row of every position
    // (idx) and the group-start bitset -- from the FULL keys (one gather of the column per position; a position's
    // predecessor is the lane below, a wave's first lane gathers its predecessor itself), and checks the order against the
    // full keys on the way: an inversion means two values of the column share the top 48 bits of their keys and differ
    // below them; the column is then sorted again with three-word elements (full key, row) and the bitset is made from
    // those.  Equal full keys are in row order either way: the row is part of the word.
    sort_pass(std::true_type{});
    int inv = 0;
    for (int base0 = 0; base0 <= ((n >> 6) << 6); base0 += NT * K0_UB) {
      // two dependent loads per position (the sorted word, then the value of its row): issued K0_UB positions at a time
      uint32_t rowv[K0_UB], rowp[K0_UB];
      double vv[K0_UB], vp[K0_UB];
#pragma unroll
      for (int u = 0; u < K0_UB; ++u) {
        const int k = base0 + u * NT + tid;
        rowv[u] = (k < n) ? ((uint32_t)keys[k] & 0xFFFFu) : 0u;
        rowp[u] = (lane == 0 && k > 0 && k < n) ? ((uint32_t)keys[k - 1] & 0xFFFFu) : 0u;   // a wave's first lane: its predecessor too
      }
#pragma unroll
      for (int u = 0; u < K0_UB; ++u) {
        const int k = base0 + u * NT + tid;
        vv[u] = (k < n) ? col[rowv[u]] : 0.0;
        vp[u] = (lane == 0 && k > 0 && k < n) ? col[rowp[u]] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < K0_UB; ++u) {
        if (base0 + u * NT <= ((n >> 6) << 6)) {   // (uniform over the workgroup)
          const int k = base0 + u * NT + tid;
          unsigned long long fk = 0ull;
          if (k < n) {
            double v = vv[u];
            if (v != v || mask_excluded(ms, v)) v = fill;
            fk = sortable_key(v);
            idx[k] = rowv[u];
          }
          unsigned long long prev = __shfl_up(fk, 1, 64);
          if (lane == 0 && k > 0 && k < n) {
            double v = vp[u];
            if (v != v || mask_excluded(ms, v)) v = fill;
            prev = sortable_key(v);
          }
          const bool st = (k <= n) && (k == 0 || k == n || prev != fk);
          inv |= (k > 0 && k < n && prev > fk) ? 1 : 0;
          const unsigned long long b = __ballot(st);
          if (lane == 0 && (k >> 6) <= (n >> 6)) sh_st[k >> 6] = b;
        }
      }
    }
    if (__syncthreads_or(inv)) sort_pass(std::false_type{});
    else starts_done = true;
  }

  // ---- phase 3: tie groups in ascending order -----------------------------------------------------
  // Group starts as a bitset in LDS, computed once with coalesced loads: bit k = "position k starts a tie group";
  // bit n is set (the position after the last one starts a group), so "position k ends a group" is bit k + 1.
  // Every position then finds its group by two bit scans of that set -- first position `lo` = the last start at or
  // before it, last position `hi` = the next start - 1 -- and thread t takes positions t, t + 1024, ...: every
  // global access of the phase is coalesced except the writes by ROW (rec through the LDS tile, hirow).  Round 2a
  // gave each thread a run of consecutive positions: the lanes of a wave then read keys and row indices at a
  // stride of 8 n / 1024 bytes, three key loads per position and pass, and carried the open group from thread to
  // thread with two 1 024-wide LDS scans (K0 without its sort: 0.44 of 1.03 ms on c4).
  if (!starts_done) {
    for (int base = 0; base <= ((n >> 6) << 6); base += NT) {
      const int k = base + tid;
      const bool st = (k <= n) && (k == 0 || k == n || keys[k - 1] != keys[k]);
      const unsigned long long b = __ballot(st);
      if (lane == 0 && (k >> 6) <= (n >> 6)) sh_st[k >> 6] = b;
    }
  }
  unsigned long long* sh_big = reinterpret_cast<unsigned long long*>(sh_ll);  // bit k: a group of >= 2 rows starts at k
  for (int w = tid; w < (WIDE ? pv.Wp : 1024); w += NT) sh_bits[w] = 0ull;
  if (WIDE) {
    for (int k = n + tid; k < pv.n_pad; k += NT) order32[k] = 0u;
  } else {
    for (int k = n + tid; k < pv.n_ord; k += NT) order[k] = 0;  // zero padding: K1 prefetches one step ahead
    if (pv.order_w) { for (int k = n + tid; k < pv.n_ord; k += NT) pv.order_w[(int64_t)c * pv.n_ord + k] = 0u; }
  }
  __syncthreads();
  auto is_start = [&](int k) -> bool { return (sh_st[k >> 6] >> (k & 63)) & 1ull; };   // 0 <= k <= n
  auto prev_start = [&](int k) -> int {   // last start at or before k (bit 0 is set)
    int w = k >> 6;
    unsigned long long m = sh_st[w] & (~0ull >> (63 - (k & 63)));
    while (m == 0ull) m = sh_st[--w];
    return (w << 6) + 63 - (int)__builtin_clzll(m);
  };
  auto next_start = [&](int k) -> int {   // first start after k (bit n is set)
    const int q = k + 1;
    int w = q >> 6;
    unsigned long long m = sh_st[w] & (~0ull << (q & 63));
    while (m == 0ull) m = sh_st[++w];
    return (w << 6) + (int)__builtin_ctzll(m);
  };

  // bit k of sh_big: a tie group of >= 2 rows starts at ascending position k (a start whose successor is not one);
  // sh_bigpre[w]: such groups in the words before w.  A group's rank among them is its place in the column's tie-group
  // list (tgroups, below) and the index every one of its rows carries in girow: the pair kernel counts a streamed
  // group's rows per tie group of the gathered column in a table of counters indexed by it.
  if (!WIDE) {
    const int nw = (n + 63) >> 6;                  // <= 1024 words
    int gbase = 0;                                 // groups in the blocks before this one (the same in every thread)
    for (int wb = 0; wb < nw; wb += NT) {
      const int wi = wb + tid;
      unsigned long long bg = 0ull;
      if (wi < nw) {
        const unsigned long long st = sh_st[wi];
        bg = st & ~((st >> 1) | (sh_st[wi + 1] << 63));
        const int last = n - 1 - wi * 64;          // positions up to n - 1 (bit n of sh_st is the end marker)
        if (last < 63) bg &= (2ull << last) - 1ull;
        sh_big[wi] = bg;
      }
      const int cnt = (int)__popcll(bg);
      const int incl = (int)wave_incl_scan((uint32_t)cnt);
      if (lane == 63) sh_i[tid >> 6] = incl;
      __syncthreads();
      int wbase = gbase, btot = 0;
      for (int w = 0; w < NW; ++w) {
        const int t = sh_i[w];
        if (w < (tid >> 6)) wbase += t;
        btot += t;
      }
      if (wi < nw) sh_bigpre[wi] = (uint16_t)(wbase + incl - cnt);
      gbase += btot;
      __syncthreads();   // (sh_i is rewritten by the next block; the tables are complete for the loop below)
    }
    want_gi = want_gi || gbase > K1_CNT_MIN_GROUPS;
  }

  // per-thread tie statistics over the groups that START at my positions
  int ngroups = 0, maxgroup = 0, tfill = 0, ntg_local = 0, oddtie = 0;
  uint32_t s0 = 0, s1 = 0, s2 = 0;      // int32 arithmetic of Rcpp sugar, as wrapping uint32
  long long e0 = 0, e1 = 0, e2 = 0;     // exact
  for (int base0 = 0; base0 < n; base0 += NT * K0_UB) {
    uint32_t rows[K0_UB];
#pragma unroll
    for (int u = 0; u < K0_UB; ++u) {
      const int k = base0 + u * NT + tid;
      rows[u] = (k < n) ? idx[k] : 0u;
    }
#pragma unroll
    for (int u = 0; u < K0_UB; ++u) {
    if (base0 + u * NT >= n) break;   // (uniform over the workgroup)
    const int k = base0 + u * NT + tid;
    if (k < n) {
      const int lo = prev_start(k), hi = next_start(k) - 1;
      const uint32_t row = rows[u];
      if (WIDE) {
        hi32[row] = (uint32_t)hi; q32[row] = (uint32_t)k; lo32[row] = (uint32_t)lo;
        order32[n - 1 - k] = row;
      } else {
      const uint32_t gi = (want_gi && hi > lo) ? ((uint32_t)sh_bigpre[lo >> 6] + (uint32_t)__popcll(sh_big[lo >> 6] & ((1ull << (lo & 63)) - 1ull)))
                                               : (uint32_t)GIROW_NONE;
      if (stage_hg) hg_s[row] = (uint32_t)hi | (gi << 16);
      else { hirow[2 * row] = (uint16_t)hi; if (want_gi) girow[2 * row] = (uint16_t)gi; }
      // rec is written by row (scattered): through the free sort tile when the column fits, then out in order
      if (stage_rec) rec_s[row] = (uint32_t)k | ((uint32_t)lo << 16);
      else rec[2 * row] = (uint32_t)k | ((uint32_t)lo << 16);
      order[n - 1 - k] = (uint16_t)row;  // processing order of K1: descending value
      if (pv.order_w) pv.order_w[(int64_t)c * pv.n_ord + (n - 1 - k)] = row;
      }
      if (lo == 0 && nna > 0) atomicOr(&sh_bits[row >> 6], 1ull << (row & 63));
      if (lo == k) {
        const int t = hi - lo + 1;
        ++ngroups;
        if (!WIDE) maxgroup = (int)max((uint32_t)maxgroup, ((uint32_t)t << 16) | (uint32_t)lo);  // size << 16 | first position
        if (lo == 0) tfill = t;
        if (t >= 2) {
          ++ntg_local;
          oddtie |= lo & 1;
          const uint32_t ut = (uint32_t)t;
          const uint32_t tt1 = ut * (ut - 1u);
          s0 += tt1;
          s1 += tt1 * (ut - 2u);
          s2 += tt1 * (2u * ut + 5u);
          const long long lt = t;
          e0 += lt * (lt - 1);
          e1 += lt * (lt - 1) * (lt - 2);
          e2 += lt * (lt - 1) * (2 * lt + 5);
        }
      }
    }
    }
  }
  __syncthreads();
  if (stage_rec) {
    for (int r = tid; r < n; r += NT) rec[2 * r] = rec_s[r];
  }
  if (stage_hg) {
    for (int r = tid; r < n; r += NT) {
      const uint32_t hg = hg_s[r];
      hirow[2 * r] = (uint16_t)hg;
      if (want_gi) girow[2 * r] = (uint16_t)(hg >> 16);
    }
  }

  // group-start flags in PROCESSING order k' = n-1-k: a group starts at k' where it ends at k
  for (int base = 0; base < pv.n_pad; base += NT) {
    const int kp = base + tid;
    bool flag = false;
    if (kp < n) {
      const int k = n - 1 - kp;
      flag = is_start(k + 1);
    }
    const unsigned long long b = __ballot(flag);
    if (lane == 0 && (kp >> 6) < W) gflag[kp >> 6] = b;
  }
  if (tid == 0) { gflag[W] = 0ull; }
  for (int w = tid; w <= W; w += NT) fmask[w] = (w < W) ? sh_bits[w] : 0ull;
  int stream_cost = 2 * ((n + 63) >> 6);   // columns without a tie program (long, wide): the steps of their walk
  if (!WIDE && pv.tp_stride > 0) {
    // the tie program of the column: its flag words, still in registers of the lanes that wrote them, go to LDS (the
    // fill-group bitset there has just been copied out) and one wave cuts the steps from that copy
    __syncthreads();
    for (int base = 0; base < pv.n_pad; base += NT) {
      const int kp = base + tid;
      const bool flag = (kp < n) && is_start(n - kp);          // the same bit as above: position n-1-kp ends a group
      const unsigned long long b = __ballot(flag);
      if (lane == 0 && (kp >> 6) < W) sh_bits_lds[kp >> 6] = b;
    }
    if (tid == 0) sh_bits_lds[W] = 0ull;
    __syncthreads();
    // scratch in the sort tile / rec staging area (48 KB, copied out above): E TPROG_WIN u16 | crossT 2 048 u32 | list
    uint16_t* Ewin = reinterpret_cast<uint16_t*>(sh_sort);
    uint32_t* crossT = reinterpret_cast<uint32_t*>(Ewin + TPROG_WIN);
    uint16_t* mlist = reinterpret_cast<uint16_t*>(crossT + 2048);
    k0_tie_program(sh_bits_lds, Ewin, mlist, crossT, &sh_i[0], n, W, pv.tprog + (int64_t)c * pv.tp_stride,
                   pv.tprog + (int64_t)c * pv.tp_stride + (pv.tp_stride - 2 * TPROG_MARKS), order,
                   pv.srow + (int64_t)c * pv.sr_steps * 64, pv.smask + (int64_t)c * pv.sr_steps * 32, pv.sr_steps,
                   (uint32_t)pv.n_pad, tid, NT);
    stream_cost = sh_i[3];   // (every thread reads it; thread 0 stores it with the statistics)
    __syncthreads();   // (sh_i is used by the reductions below)
  }

  // list of the tie groups (size >= 2) in ascending order, lo | hi << 16: K1 counts the joint ties of a
  // tie group of the OTHER column that spans several steps once, when that group closes.  A group's place in the
  // list = the groups of >= 2 rows that start before it: a prefix over the words of sh_big.
  if (!WIDE) {
    const int nw = (n + 63) >> 6;
    for (int wi = tid; wi < nw; wi += NT) {
      int off = (int)sh_bigpre[wi];
      unsigned long long m = sh_big[wi];
      while (m != 0ull) {
        const int k = (wi << 6) + (int)__builtin_ctzll(m);
        m &= m - 1ull;
        tgl[off++] = (uint32_t)k | ((uint32_t)(next_start(k) - 1) << 16);
      }
    }
  }
  // the column's statistics: every value reduced inside its wave, the waves' results combined by thread 0 -- one barrier
  const auto add_i = [](int a, int b) { return a + b; };
  const auto add_u = [](uint32_t a, uint32_t b) { return a + b; };
  const auto add_l = [](long long a, long long b) { return a + b; };
  const int ntg_w = wave_reduce(ntg_local, add_i);
  ngroups = wave_reduce(ngroups, add_i);
  maxgroup = (int)wave_reduce((uint32_t)maxgroup, [](uint32_t a, uint32_t b) { return a > b ? a : b; });
  tfill = wave_reduce(tfill, [](int a, int b) { return a > b ? a : b; });
  oddtie = wave_reduce(oddtie, [](int a, int b) { return a | b; });
  s0 = wave_reduce(s0, add_u); s1 = wave_reduce(s1, add_u); s2 = wave_reduce(s2, add_u);
  e0 = wave_reduce(e0, add_l); e1 = wave_reduce(e1, add_l); e2 = wave_reduce(e2, add_l);
  __syncthreads();   // (sh_i / sh_ll: the list offsets above and the big-group bitset are done with)
  if (lane == 0) {
    int* wi = sh_i + (tid >> 6) * 8;
    wi[0] = ntg_w; wi[1] = ngroups; wi[2] = maxgroup; wi[3] = tfill; wi[4] = oddtie; wi[5] = (int)s0; wi[6] = (int)s1; wi[7] = (int)s2;
    long long* wl = sh_ll + (tid >> 6) * 3;
    wl[0] = e0; wl[1] = e1; wl[2] = e2;
  }
  __syncthreads();
  int ntg = 0;
  if (tid == 0) {
    ngroups = 0; maxgroup = 0; tfill = 0; oddtie = 0; s0 = s1 = s2 = 0u; e0 = e1 = e2 = 0;
    for (int w = 0; w < NW; ++w) {
      const int* wi = sh_i + w * 8;
      ntg += wi[0]; ngroups += wi[1];
      maxgroup = ((uint32_t)wi[2] > (uint32_t)maxgroup) ? wi[2] : maxgroup;
      tfill = max(tfill, wi[3]); oddtie |= wi[4];
      s0 += (uint32_t)wi[5]; s1 += (uint32_t)wi[6]; s2 += (uint32_t)wi[7];
      const long long* wl = sh_ll + w * 3;
      e0 += wl[0]; e1 += wl[1]; e2 += wl[2];
    }
  }

  if (tid == 0) {
    ColStats st;
    st.nna = nna;
    st.ngroups = ngroups;
    st.tfill = (nna > 0) ? tfill : 0;
    st.maxgroup = maxgroup;
    st.s0 = s0; st.s1 = s1; st.s2 = s2; st.ntg = (uint32_t)ntg;
    st.e0 = e0; st.e1 = e1; st.e2 = e2;
    st.fill = fill;
    st.nexcl = nexcl;
    st.flags = (oddtie ? COL_ODD_TIE : 0) | (int32_t)((uint32_t)min(stream_cost, 0xFFFFFF) << 8);
    *pv.col_stats(c) = st;
  }
}

// the kernels of the pre-pass: one body, four shapes (a register budget is a per-kernel attribute)
__global__ void __launch_bounds__(K0_THREADS, ICIKT_K0_MIN_WAVES)
k0_prepare_large(PrepView pv, const double* __restrict__ X, int64_t ld, int col_begin, const MaskSpec ms, uint8_t* __restrict__ keep) {
  k0_prepare_body<false, K0_THREADS, 4>(pv, X, ld, col_begin, ms, keep);
}
// columns of more than 4 096 rows: 8 elements per thread, an 8 192-element tile (96 KB of LDS, 124 KB with the bitsets) --
// half the tiles, one merge level less through the global scratch: 20 000 rows 0.90 -> 0.70 ms per 512 columns, the yeast
// shape 0.090 -> 0.079, 50 000 rows 2.80 -> 2.62 (round 4).  Shorter columns would take the all-LDS path of a partial
// tile there (3 000 rows: 0.19 -> 0.26 ms per 1 024 columns) and keep the 4 096-element tile.
__global__ void __launch_bounds__(K0_THREADS, ICIKT_K0_MIN_WAVES)
k0_prepare_large8(PrepView pv, const double* __restrict__ X, int64_t ld, int col_begin, const MaskSpec ms, uint8_t* __restrict__ keep) {
  k0_prepare_body<false, K0_THREADS, 8>(pv, X, ld, col_begin, ms, keep);
}
__global__ void __launch_bounds__(K0_THREADS, ICIKT_K0_MIN_WAVES)
k0_prepare_wide(PrepView pv, const double* __restrict__ X, int64_t ld, int col_begin, const MaskSpec ms, uint8_t* __restrict__ keep) {
  k0_prepare_body<true, K0_THREADS, K0_TILE / K0_THREADS>(pv, X, ld, col_begin, ms, keep);
}
__global__ void __launch_bounds__(K0_THREADS, ICIKT_K0_WAVES_SMALL)   // (bounds of the LARGE shape: they make the register budget binding; launched with K0_THREADS_SMALL threads)
k0_prepare_small(PrepView pv, const double* __restrict__ X, int64_t ld, int col_begin, const MaskSpec ms, uint8_t* __restrict__ keep) {
  k0_prepare_body<false, K0_THREADS_SMALL, K0_TILE / K0_THREADS_SMALL>(pv, X, ld, col_begin, ms, keep);
}

// ------------------------------------------------------------------------------------------------
// K0x: rebuild rec / hirow / tgroups of a column from its order and gflag
// ------------------------------------------------------------------------------------------------
// order (the descending permutation) and gflag (its tie-group starts) determine the other per-row arrays:
// for the row at descending position k, in the group [s, e] of descending positions,
//   q = n-1-k,  lo = n-1-e,  hi = n-1-s   (ascending position, first and last position of its tie group).
// Ranks therefore exchange only order, the three bitsets and stats (24 KB per column of length 10 000 instead
// of 104 KB) and rebuild the rest locally.  One workgroup per column: wave 0 scans the flag words, then the
// waves take the 64-position steps in turn.
constexpr int KX_WAVES = 4;
__global__ void __launch_bounds__(64 * KX_WAVES) k0_expand(PrepView pv, int col_begin, int ncols, int staged) {
  // staged: the scattered per-row writes go to an LDS copy of the column's rec / hirow first and leave as
  // sequential stores (dynamic LDS: n_pad * 8 bytes; the host stages columns of up to 12 288 rows)
  extern __shared__ __attribute__((aligned(16))) unsigned char kx_stage[];
  uint32_t* rec_s = reinterpret_cast<uint32_t*>(kx_stage);
  uint16_t* hi_s = reinterpret_cast<uint16_t*>(kx_stage + (size_t)pv.n_pad * 4);
  uint16_t* gi_s = reinterpret_cast<uint16_t*>(kx_stage + (size_t)pv.n_pad * 6);
  __shared__ __attribute__((aligned(8))) int prevs[1032];   // highest group start in the words before w (-1: none)
  __shared__ int nexts[1032];   // lowest group start in the words after w (n: none)
  __shared__ int msuf[1032];    // groups of size >= 2 that start in the words after w
  __shared__ uint32_t kx_cross[32 * 64];   // k0_tie_program's scratch: cross table, would-be steps, MIXED step list
  __shared__ uint16_t kx_E[TPROG_WIN];
  __shared__ uint16_t kx_list[2 * TPROG_LIST];
  __shared__ int kx_cnt[24];
  const int wave = (int)(threadIdx.x >> 6);
  const int lane = (int)(threadIdx.x & 63);
  const int c = col_begin + (int)blockIdx.x;
  const int n = pv.n, W = pv.W, Wp = pv.Wp;
  const unsigned long long* gf = pv.col_gflag(c);
  const uint16_t* ord = pv.order + (int64_t)c * pv.n_ord;
  uint32_t* rec = pv.rec + ((int64_t)(c >> 1) * pv.rec_rows) * 2 + (c & 1);
  uint16_t* hirow = pv.hirow + ((int64_t)(c >> 1) * pv.rec_rows) * 2 + (c & 1);
  uint16_t* girow = pv.girow + ((int64_t)(c >> 1) * pv.rec_rows) * 2 + (c & 1);
  uint32_t* tgl = pv.tgroups + (int64_t)c * pv.tg_stride;
  if (threadIdx.x == 0) {   // the guard row (PrepView::rec_rows)
    rec[2 * pv.n_pad] = (uint32_t)pv.n_pad; hirow[2 * pv.n_pad] = 0; girow[2 * pv.n_pad] = GIROW_NONE;
  }
  if (pv.order_w) {   // long columns: the received order (with its zero padding) as 32-bit words
    uint32_t* ow = pv.order_w + (int64_t)c * pv.n_ord;
    for (int k = (int)threadIdx.x; k < pv.n_ord; k += (int)blockDim.x) ow[k] = (uint32_t)ord[k];
  }

  // starts of groups of size >= 2: a start whose successor position exists and is not a start
  auto multi = [&](int w) -> unsigned long long {
    const unsigned long long f = gf[w];
    const unsigned long long fn = (f >> 1) | (gf[w + 1] << 63);  // gflag has a zero guard word at W
    unsigned long long m = f & ~fn;
    const int last = n - 2 - w * 64;  // positions k <= n-2 have a successor
    if (last < 63) m &= (last < 0) ? 0ull : ((2ull << last) - 1ull);
    return m;
  };
  if (wave == 0) {
  const int items = (W + 63) >> 6;
  const int w0 = min(W, lane * items), w1 = min(W, w0 + items);
  int hb = -1, lb = n, mc = 0;
  for (int w = w0; w < w1; ++w) {
    const unsigned long long f = gf[w];
    if (f != 0ull) {
      hb = w * 64 + 63 - (int)__builtin_clzll(f);
      if (lb == n) lb = w * 64 + (int)__builtin_ctzll(f);
    }
    mc += (int)__popcll(multi(w));
  }
  // exclusive prefix max of hb, exclusive suffix min of lb, exclusive suffix sum of mc over the lanes
  int pmax = hb, smin = lb, ssum = mc;
  for (int o = 1; o < 64; o <<= 1) {
    const int a = __shfl_up(pmax, o, 64), b = __shfl_down(smin, o, 64), d = __shfl_down(ssum, o, 64);
    if (lane >= o) pmax = max(pmax, a);
    if (lane + o < 64) { smin = min(smin, b); ssum += d; }
  }
  int run_prev = __shfl_up(pmax, 1, 64);
  if (lane == 0) run_prev = -1;
  int run_next = __shfl_down(smin, 1, 64), run_ms = __shfl_down(ssum, 1, 64);
  if (lane == 63) { run_next = n; run_ms = 0; }
  for (int w = w0; w < w1; ++w) {
    prevs[w] = run_prev;
    const unsigned long long f = gf[w];
    if (f != 0ull) run_prev = w * 64 + 63 - (int)__builtin_clzll(f);
  }
  for (int w = w1 - 1; w >= w0; --w) {
    nexts[w] = run_next;
    msuf[w] = run_ms;
    const unsigned long long f = gf[w];
    if (f != 0ull) run_next = w * 64 + (int)__builtin_ctzll(f);
    run_ms += (int)__popcll(multi(w));
  }
  }
  __syncthreads();

  for (int w = wave; w < W; w += KX_WAVES) {
    const int k = w * 64 + lane;
    if (k >= n) break;
    const unsigned long long f = gf[w];
    const unsigned long long le = (lane < 63) ? ((2ull << lane) - 1ull) : ~0ull;  // bits <= lane
    const unsigned long long below = f & le, above = f & ~le;
    const int s = (below != 0ull) ? w * 64 + 63 - (int)__builtin_clzll(below) : prevs[w];
    const int e = ((above != 0ull) ? w * 64 + (int)__builtin_ctzll(above) : nexts[w]) - 1;
    const uint32_t row = ord[k];
    const uint32_t lo = (uint32_t)(n - 1 - e), hi = (uint32_t)(n - 1 - s);
    const uint32_t rv = (uint32_t)(n - 1 - k) | (lo << 16);
    if (staged) { rec_s[row] = rv; hi_s[row] = (uint16_t)hi; }
    else { rec[2 * row] = rv; hirow[2 * row] = (uint16_t)hi; }
    // the group's place in tgroups (ascending in lo: groups that start after it -- descending -- come first), which is
    // the index its rows carry in girow
    uint32_t gi = GIROW_NONE;
    if (e > s) {
      const int sw = s >> 6;
      const unsigned long long les = ((s & 63) < 63) ? ((2ull << (s & 63)) - 1ull) : ~0ull;   // bits <= s
      gi = (uint32_t)(msuf[sw] + (int)__popcll(multi(sw) & ~les));
      if (k == s) tgl[gi] = lo | (hi << 16);
    }
    if (staged) gi_s[row] = (uint16_t)gi;
    else girow[2 * row] = (uint16_t)gi;
  }
  if (staged) {
    __syncthreads();
    for (int r = (int)threadIdx.x; r < n; r += 64 * KX_WAVES) {
      rec[2 * r] = rec_s[r];
      hirow[2 * r] = hi_s[r];
      girow[2 * r] = gi_s[r];
    }
  }
  if (pv.tp_stride > 0) {   // (half-wave kernels only: W <= 287 words)
    __syncthreads();
    unsigned long long* gfl = reinterpret_cast<unsigned long long*>(prevs);   // the scan arrays are free now: 1032 ints = 516 words
    for (int w = (int)threadIdx.x; w <= W; w += 64 * KX_WAVES) gfl[w] = gf[w];
    __syncthreads();
    k0_tie_program(gfl, kx_E, kx_list, kx_cross, kx_cnt, n, W, pv.tprog + (int64_t)c * pv.tp_stride,
                   pv.tprog + (int64_t)c * pv.tp_stride + (pv.tp_stride - 2 * TPROG_MARKS), ord,
                   pv.srow + (int64_t)c * pv.sr_steps * 64, pv.smask + (int64_t)c * pv.sr_steps * 32, pv.sr_steps,
                   (uint32_t)pv.n_pad, (int)threadIdx.x, 64 * KX_WAVES);
  }
}

// ------------------------------------------------------------------------------------------------
// K1: column pairs on wavefronts -- one pair per wave, or two (one per 32-lane half, sharing the gathered
// column, each streaming its own)
// ------------------------------------------------------------------------------------------------
// Column B (= the pairs' common pi) is the random-access side: rec[row] = q | lo << 16, q = the row's
// position in B's ascending stable order (unique), lo = first position of its tie group.
// Columns A_0..A_{NP-1} (= pj of the task's pairs) are the streamed side: order[k] = row at position k
// of A's DESCENDING order, gflag bit k = "position k starts a new tie group of A".
//
// The wave walks A from the largest value down, 64 positions per step, and keeps per pair in LDS
//   seen : bitset over B-positions of every row whose A-group is strictly above the current one
//   spre : per-64-bit-word exclusive prefix popcounts of seen
// so that   #{rows j : a_j > a_l, b_j < b_l} = spre[lo_l >> 6] + popc(seen[lo_l >> 6] & below(lo_l)).
// Rows that enter `seen` together are compared all-pairs in registers: DPP shifts inside a 16-lane row and
// balanced rotations between rows (wave_allpairs: 64 rows of one pair; half_allpairs: 32 rows of each of
// two pairs), one v_sub_co_u32_dpp + v_addc_co_u32 per compare.
// The rows of an A tie group are queried while `seen` stands still and inserted together when the group closes (GROUP
// steps); its joint ties (compare_both, kendallc.cpp:33-51) come from range counts of `seen` before and after.
// (Rounds 1-2 kept the open group's rows in a second bitset `pend` -- in LDS, or in per-wave global slots with a
// persistent grid for long columns -- and a general out-of-line step that re-sorted the lanes of mixed steps.)
struct WaveLds {
  unsigned long long* seen;
  uint16_t* spre;
};

// ---- half-wave hot step: one pair per 32-lane half, 32 rows per sub-step --------------------------
// In-step all-pairs without a shift chain: VOP2 instructions take a DPP control on src0, and inside a
// 16-lane row DPP shifts by any immediate 1..15, so "is the q of the lane b below me less than my lo" is ONE
// instruction: v_sub_co_u32_dpp computes src0_dpp - src1 and leaves the borrow (src1 > src0_dpp) in VCC.
// (VOPC has no DPP form on gfx9, and v_subrev_co_u32_dpp ignores the DPP control: tools/ubench/dpp_sem.hip.)
//  * inside a DPP row (15 compares, row_shr:1..15): a lane without a source reads 0 (bound_ctrl:0) and would
//    count whenever lo > 0: that is a known number per lane (15 - position in the row), subtracted afterwards.
//  * between the half's two rows (256 pairs, all "lower row earlier"): BOTH rows work, 8 compares instead of
//    16.  An upper lane asks "lo_me > q_l" of 8 lower lanes; a lower lane asks the same question about
//    itself, "lo_u > q_me", of the other 8 upper lanes, as "~q_me > ~lo_u" so that the same instruction
//    serves both.  xa holds what a lane's DPP sources offer (upper lanes: the lower row's q; lower lanes: the
//    upper row's ~lo rotated by one position; half_count builds it), xb = what the lane compares with (upper: lo,
//    lower: ~q); with
//    row_ror:0..7 (lane p reads p - r) the upper lanes cover (lower - upper) mod 16 in {0, 15, .., 9} and
//    the lower lanes {1, .., 8}.
// 23 subtract-with-borrow + 23 add-with-carry per 32 rows of a pair.
#define ICIKT_HSHR(b) "v_sub_co_u32_dpp %1, vcc, %2, %3 row_shr:" #b " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
                      "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
#define ICIKT_HROR(b) "v_sub_co_u32_dpp %1, vcc, %4, %5 row_ror:" #b " row_mask:0xf bank_mask:0xf\n\t" \
                      "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
__device__ __forceinline__ uint32_t half_allpairs(uint32_t q, uint32_t lo, uint32_t xa, uint32_t xb, uint32_t lane) {
  uint32_t acc = 0, junk;
  // s_nop 1: a DPP read needs two wait states after a VALU write of its source, and hipcc pads nothing
  // inside an asm statement (q is typically produced by the instruction just before it)
  asm volatile("s_nop 1\n\t"
               ICIKT_HSHR(1) ICIKT_HSHR(2) ICIKT_HSHR(3) ICIKT_HSHR(4) ICIKT_HSHR(5) ICIKT_HSHR(6) ICIKT_HSHR(7)
               ICIKT_HSHR(8) ICIKT_HSHR(9) ICIKT_HSHR(10) ICIKT_HSHR(11) ICIKT_HSHR(12) ICIKT_HSHR(13)
               ICIKT_HSHR(14) ICIKT_HSHR(15)
               "v_cmp_lt_u32_e32 vcc, %4, %5\n\tv_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
               ICIKT_HROR(1) ICIKT_HROR(2) ICIKT_HROR(3) ICIKT_HROR(4) ICIKT_HROR(5) ICIKT_HROR(6) ICIKT_HROR(7)
               : "+v"(acc), "=&v"(junk)
               : "v"(q), "v"(lo), "v"(xa), "v"(xb)
               : "vcc");
  // shifts 1..15 ran past the row start for 15 - p of the steps (p = position in the row) and read 0
  return acc - ((lo != 0u) ? (15u - (lane & 15u)) : 0u);
}

// per-lane share of #{rows a before row j in the same 32-lane half : q_a < lo_j}; only the sum over the half
// is meaningful (lower-row lanes carry part of the upper row's counts)
__device__ __forceinline__ uint32_t half_count(uint32_t q, uint32_t lo, uint32_t lane) {
  const uint32_t offer = (lane & 16u) ? ~lo : q;
  // the rows of a half exchanged in registers (the LDS is the busiest unit of this kernel): v_permlane16_swap
  // of two copies gives r[0] = the lower rows' offers in both rows, r[1] = the upper rows'; the lower rows read
  // the upper row rotated by one position
  const auto r = __builtin_amdgcn_permlane16_swap(offer, offer, false, false);
  const uint32_t rot = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r[1], 0x121 /*row_ror:1*/, 0xf, 0xf, false);
  const uint32_t xa = (lane & 16u) ? r[0] : rot;
  return half_allpairs(q, lo, xa, ~offer, lane);
}

// ---- both 32-row sub-steps of a half-wave step in one chain (hot steps) -----------------------------------------
// The in-step pairs of a sub-step need registers only, so the two sub-steps of a 64-row step are counted TOGETHER,
// one in each 16-bit half of the operands (positions of a half-wave kernel -- the guard position 64 W included -- are
// below 2^15: n <= 30 656):
//     A = 0x7FFF - q   (what a row offers)        B = lo   (what a row compares with)
//     A_a + B_j = 0x7FFF + lo_j - q_a :  bit 15 set  <=>  q_a < lo_j,  and nothing carries into the upper half.
// One DPP add per shift distance serves 2 x 64 compares, and its two flag bits (15, 31) are shifted into a bit
// vector by two full-rate instructions (v_lshrrev, v_and_or) -- the carry forms of half_allpairs (v_sub_co_dpp +
// v_addc) are both half-rate.  Every flag of a lane counts for the lane's pair, so the vectors are only
// popcounted.  A lane without a source (row_shr past the row start, bound_ctrl:0) adds 0: B < 0x8000, no flag, no
// correction.  Between the half's two rows the sum is symmetric: an upper lane adds the lower rows' A to its B, a
// lower lane the upper rows' B to its A, so `offer` is both what a lane offers and what it adds (layout of the
// partners as in half_count).  At most 16 flags per field and vector: one vector for the 15 in-row distances, one
// for the 8 between rows.
// (In a mixed instruction stream every VALU instruction of this kernel costs about one 4-cycle issue slot, whatever
// its class -- tools/ubench/valu_rate.hip -- so what counts is the NUMBER of instructions.)  The flag bits of two
// sums are gathered by one v_perm_b32 (bytes 1 and 3 of both: four flags at bit 7 of a byte) and shifted into the
// vector together: 2.5 instructions per 256 compares.
#define ICIKT_PACC  "v_perm_b32 %1, %1, %2, %6\n\tv_lshrrev_b32_e32 %0, 1, %0\n\tv_and_or_b32 %0, %1, %5, %0\n\t"
#define ICIKT_PSHR2(a, b) "v_add_u32_dpp %1, %3, %4 row_shr:" #a " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
                          "v_add_u32_dpp %2, %3, %4 row_shr:" #b " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" ICIKT_PACC
#define ICIKT_PROR2(a, b) "v_add_u32_dpp %1, %3, %4 row_ror:" #a " row_mask:0xf bank_mask:0xf\n\t" \
                          "v_add_u32_dpp %2, %3, %4 row_ror:" #b " row_mask:0xf bank_mask:0xf\n\t" ICIKT_PACC
// byte address, for ds_bpermute, of the lane whose offer a lane reads at distance 0 between the rows: an upper
// lane the lower row's lane at its own position, a lower lane the upper row's lane one position to the left
__device__ __forceinline__ uint32_t half_partner_addr(uint32_t lane) {
  const uint32_t p = lane & 15u, h = lane & 32u;
  return ((lane & 16u) ? (h + p) : (h + 16u + ((p + 15u) & 15u))) * 4u;
}
// (in three parts for the long-column kernels, which spread the chain over the LDS waits of a step's two sub-steps: the
//  operands with the partners' offers -- one ds_bpermute --, the in-row sums, the sums between the rows)
struct HalfChain { uint32_t A, B, offer, xa; };
__device__ __forceinline__ HalfChain half_step_prep(uint32_t sw0, uint32_t sw1, uint32_t lane, uint32_t partner_addr) {
  HalfChain h;
  const uint32_t Q = __builtin_amdgcn_perm(sw1, sw0, 0x05040100u);    // q of sub-step 0 | q of sub-step 1 << 16
  const uint32_t LO = __builtin_amdgcn_perm(sw1, sw0, 0x07060302u);   // lo likewise
  h.A = 0x7FFF7FFFu - Q; h.B = LO;
  h.offer = (lane & 16u) ? h.B : h.A;
  h.xa = (uint32_t)__builtin_amdgcn_ds_bpermute((int)partner_addr, (int)h.offer);
  return h;
}
__device__ __forceinline__ uint32_t half_step_part_a(const HalfChain& h, uint32_t acc) {
  const uint32_t A = h.A, B = h.B, xa = h.xa, offer = h.offer;
  const uint32_t M8 = 0x80808080u, SEL = 0x07050301u;
  uint32_t v1, t1, t2;
  asm volatile("s_nop 1\n\t"
               "v_add_u32_dpp %1, %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
               "v_add_u32_dpp %2, %3, %4 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
               "v_perm_b32 %1, %1, %2, %6\n\tv_and_b32_e32 %0, %5, %1\n\t"
               ICIKT_PSHR2(3, 4) ICIKT_PSHR2(5, 6) ICIKT_PSHR2(7, 8) ICIKT_PSHR2(9, 10) ICIKT_PSHR2(11, 12)
               ICIKT_PSHR2(13, 14)
               "v_add_u32_dpp %1, %3, %4 row_shr:15 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
               "v_add_u32_e32 %2, %7, %8\n\t" ICIKT_PACC
               : "=&v"(v1), "=&v"(t1), "=&v"(t2)
               : "v"(A), "v"(B), "s"(M8), "s"(SEL), "v"(xa), "v"(offer));
  return bcnt_acc(v1, acc);
}
__device__ __forceinline__ uint32_t half_step_part_b(const HalfChain& h, uint32_t acc) {
  const uint32_t xa = h.xa, offer = h.offer;
  const uint32_t M8 = 0x80808080u, SEL = 0x07050301u, M16 = 0x80008000u;
  uint32_t v2, t1, t2;
  asm volatile("s_nop 1\n\t"
               "v_add_u32_dpp %1, %3, %4 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_u32_dpp %2, %3, %4 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
               "v_perm_b32 %1, %1, %2, %6\n\tv_and_b32_e32 %0, %5, %1\n\t"
               ICIKT_PROR2(3, 4) ICIKT_PROR2(5, 6)
               "v_add_u32_dpp %1, %3, %4 row_ror:7 row_mask:0xf bank_mask:0xf\n\t"
               "v_lshrrev_b32_e32 %0, 1, %0\n\tv_and_or_b32 %0, %1, %7, %0\n\t"
               : "=&v"(v2), "=&v"(t1), "=&v"(t2)
               : "v"(xa), "v"(offer), "s"(M8), "s"(SEL), "s"(M16));
  return bcnt_acc(v2, acc);
}
__device__ __forceinline__ uint32_t half_step_count(uint32_t sw0, uint32_t sw1, uint32_t lane, uint32_t partner_addr) {
  const uint32_t Q = __builtin_amdgcn_perm(sw1, sw0, 0x05040100u);    // q of sub-step 0 | q of sub-step 1 << 16
  const uint32_t LO = __builtin_amdgcn_perm(sw1, sw0, 0x07060302u);   // lo likewise
  const uint32_t A = 0x7FFF7FFFu - Q, B = LO;
  const uint32_t offer = (lane & 16u) ? B : A;
  // the partners' offers come through the LDS crossbar (no memory is touched): the LDS unit has slack, the vector
  // unit has none (permlane16_swap + rotate + select were 7 issue slots)
  const uint32_t xa = (uint32_t)__builtin_amdgcn_ds_bpermute((int)partner_addr, (int)offer);
  const uint32_t M8 = 0x80808080u, SEL = 0x07050301u, M16 = 0x80008000u;
  uint32_t v1, v2, t1, t2;
  // vector 1: the 15 in-row distances and distance 0 between the rows (8 x 4 flags: bits 0..7 of every byte)
  asm volatile("s_nop 1\n\t"
               "v_add_u32_dpp %1, %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
               "v_add_u32_dpp %2, %3, %4 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
               "v_perm_b32 %1, %1, %2, %6\n\tv_and_b32_e32 %0, %5, %1\n\t"
               ICIKT_PSHR2(3, 4) ICIKT_PSHR2(5, 6) ICIKT_PSHR2(7, 8) ICIKT_PSHR2(9, 10) ICIKT_PSHR2(11, 12)
               ICIKT_PSHR2(13, 14)
               "v_add_u32_dpp %1, %3, %4 row_shr:15 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
               "v_add_u32_e32 %2, %7, %8\n\t" ICIKT_PACC
               : "=&v"(v1), "=&v"(t1), "=&v"(t2)
               : "v"(A), "v"(B), "s"(M8), "s"(SEL), "v"(xa), "v"(offer));
  // vector 2: distances 1..7 between the rows (the last sum alone: its flags, bits 15 and 31, go in unpermuted)
  asm volatile("s_nop 1\n\t"
               "v_add_u32_dpp %1, %3, %4 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_u32_dpp %2, %3, %4 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
               "v_perm_b32 %1, %1, %2, %6\n\tv_and_b32_e32 %0, %5, %1\n\t"
               ICIKT_PROR2(3, 4) ICIKT_PROR2(5, 6)
               "v_add_u32_dpp %1, %3, %4 row_ror:7 row_mask:0xf bank_mask:0xf\n\t"
               "v_lshrrev_b32_e32 %0, 1, %0\n\tv_and_or_b32 %0, %1, %7, %0\n\t"
               : "=&v"(v2), "=&v"(t1), "=&v"(t2)
               : "v"(xa), "v"(offer), "s"(M8), "s"(SEL), "s"(M16));
  return bcnt_acc(v2, bcnt_acc(v1, 0u));
}

// The same chain, returning the flag vectors themselves (layout: k0_tie_program) instead of their popcount, for steps
// whose rows form tie groups of the streamed column (MIXED): `src` is what a row offers to the rows behind it inside
// its DPP row, `own` what a row adds; between the rows `offer` is both.  (src, own, offer) = (A, B, upper ? B : A)
// gives [q_earlier < lo_later] per pair, (B, A, upper ? A : B) the other direction, [q_later < lo_earlier].
__device__ __forceinline__ void half_step_flags(uint32_t src, uint32_t own, uint32_t offer, uint32_t partner_addr,
                                                uint32_t& v1, uint32_t& v2) {
  const uint32_t xa = (uint32_t)__builtin_amdgcn_ds_bpermute((int)partner_addr, (int)offer);
  const uint32_t M8 = 0x80808080u, SEL = 0x07050301u, M16 = 0x80008000u;
  uint32_t t1, t2;
  asm volatile("s_nop 1\n\t"
               "v_add_u32_dpp %1, %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
               "v_add_u32_dpp %2, %3, %4 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
               "v_perm_b32 %1, %1, %2, %6\n\tv_and_b32_e32 %0, %5, %1\n\t"
               ICIKT_PSHR2(3, 4) ICIKT_PSHR2(5, 6) ICIKT_PSHR2(7, 8) ICIKT_PSHR2(9, 10) ICIKT_PSHR2(11, 12)
               ICIKT_PSHR2(13, 14)
               "v_add_u32_dpp %1, %3, %4 row_shr:15 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
               "v_add_u32_e32 %2, %7, %8\n\t" ICIKT_PACC
               : "=&v"(v1), "=&v"(t1), "=&v"(t2)
               : "v"(src), "v"(own), "s"(M8), "s"(SEL), "v"(xa), "v"(offer));
  asm volatile("s_nop 1\n\t"
               "v_add_u32_dpp %1, %3, %4 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_u32_dpp %2, %3, %4 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
               "v_perm_b32 %1, %1, %2, %6\n\tv_and_b32_e32 %0, %5, %1\n\t"
               ICIKT_PROR2(3, 4) ICIKT_PROR2(5, 6)
               "v_add_u32_dpp %1, %3, %4 row_ror:7 row_mask:0xf bank_mask:0xf\n\t"
               "v_lshrrev_b32_e32 %0, 1, %0\n\tv_and_or_b32 %0, %1, %7, %0\n\t"
               : "=&v"(v2), "=&v"(t1), "=&v"(t2)
               : "v"(xa), "v"(offer), "s"(M8), "s"(SEL), "s"(M16));
}

// The same flags for the NEAR distances only (round 4).  A MIXED step runs the chain twice: [q_earlier < lo_later] at
// every distance -- rows of different groups are discordant or not wherever they sit in the sub-step -- and the other
// direction, [q_later < lo_earlier], which only tells JOINT TIES apart and is looked at under the same-group masks
// alone: pairs at most (largest group of the step) - 1 lanes apart.  The pre-pass knows that size and passes its class
// with the masks (k0_step_masks): C = 0: groups of <= 3 rows (distances 1, 2), 1: <= 5 (1 .. 4), 2: <= 9 (1 .. 8; between
// the rows of a half every rotation can hold a pair up to 8 lanes apart).  The flags land where half_step_flags puts
// them (the remaining shifts of the chain in one), so the masks apply unchanged; the flags of the distances left out are 0
// and no mask bit of such a step sits there.  10 / 22 / 39 instructions instead of 58.
template <int C>
__device__ __forceinline__ void half_step_flags_near(uint32_t src, uint32_t own, uint32_t offer, uint32_t partner_addr,
                                                     uint32_t& v1, uint32_t& v2) {
  const uint32_t xa = (uint32_t)__builtin_amdgcn_ds_bpermute((int)partner_addr, (int)offer);
  const uint32_t M8 = 0x80808080u, SEL = 0x07050301u, M16 = 0x80008000u;
  uint32_t t1, t2;
#define ICIKT_NEAR_HEAD "s_nop 1\n\t"                                                                          \
               "v_add_u32_dpp %1, %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"                \
               "v_add_u32_dpp %2, %3, %4 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"                \
               "v_perm_b32 %1, %1, %2, %6\n\tv_and_b32_e32 %0, %5, %1\n\t"
  if constexpr (C == 0) {
    asm volatile(ICIKT_NEAR_HEAD "v_lshrrev_b32_e32 %0, 7, %0\n\t"
                 : "=&v"(v1), "=&v"(t1), "=&v"(t2) : "v"(src), "v"(own), "s"(M8), "s"(SEL));
  } else if constexpr (C == 1) {
    asm volatile(ICIKT_NEAR_HEAD ICIKT_PSHR2(3, 4) "v_lshrrev_b32_e32 %0, 6, %0\n\t"
                 : "=&v"(v1), "=&v"(t1), "=&v"(t2) : "v"(src), "v"(own), "s"(M8), "s"(SEL));
  } else {
    asm volatile(ICIKT_NEAR_HEAD ICIKT_PSHR2(3, 4) ICIKT_PSHR2(5, 6) ICIKT_PSHR2(7, 8) "v_lshrrev_b32_e32 %0, 4, %0\n\t"
                 : "=&v"(v1), "=&v"(t1), "=&v"(t2) : "v"(src), "v"(own), "s"(M8), "s"(SEL));
  }
#undef ICIKT_NEAR_HEAD
#define ICIKT_NEAR_HEAD2 "s_nop 1\n\t"                                                                         \
               "v_add_u32_dpp %1, %3, %4 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"                             \
               "v_add_u32_dpp %2, %3, %4 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"                             \
               "v_perm_b32 %1, %1, %2, %6\n\tv_and_b32_e32 %0, %5, %1\n\t"
  if constexpr (C == 0) {
    asm volatile(ICIKT_NEAR_HEAD2 "v_lshrrev_b32_e32 %0, 3, %0\n\t"
                 : "=&v"(v2), "=&v"(t1), "=&v"(t2) : "v"(xa), "v"(offer), "s"(M8), "s"(SEL));
  } else if constexpr (C == 1) {
    asm volatile(ICIKT_NEAR_HEAD2 ICIKT_PROR2(3, 4) "v_lshrrev_b32_e32 %0, 2, %0\n\t"
                 : "=&v"(v2), "=&v"(t1), "=&v"(t2) : "v"(xa), "v"(offer), "s"(M8), "s"(SEL));
  } else {
    asm volatile(ICIKT_NEAR_HEAD2 ICIKT_PROR2(3, 4) ICIKT_PROR2(5, 6)
                 "v_add_u32_dpp %1, %3, %4 row_ror:7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_lshrrev_b32_e32 %0, 1, %0\n\tv_and_or_b32 %0, %1, %7, %0\n\t"
                 : "=&v"(v2), "=&v"(t1), "=&v"(t2) : "v"(xa), "v"(offer), "s"(M8), "s"(SEL), "s"(M16));
  }
#undef ICIKT_NEAR_HEAD2
}

// ---- one pair on the whole wave: all-pairs of a 64-row step ----------------------------------------
// Same construction as the half-wave step, for four DPP rows: 15 in-row compares (row_shr) and, for the six
// row pairs, three rounds in which every row has one partner row (1: 0-1 2-3, 2: 0-2 1-3, 3: 0-3 1-2) and both
// partners work, 8 compares per round: 39 compare + carry pairs per 64 rows instead of a 63-step shift chain.
// In round c a lane of the later row reads the earlier row's q at its own position, a lane of the earlier row
// the later row's ~lo one position to the left (wave_cross_idx), through ds_bpermute.
__device__ __forceinline__ uint32_t wave_cross_idx(uint32_t lane, uint32_t c) {
  const uint32_t later = (c == 1u) ? (lane & 16u) : (lane & 32u);
  const uint32_t prow = (lane ^ (c << 4)) & 48u;
  return (prow | ((later ? lane : lane - 1u) & 15u)) << 2;
}
#define ICIKT_WROR(b, xa, xb) "v_sub_co_u32_dpp %1, vcc, " xa ", " xb " row_ror:" #b " row_mask:0xf bank_mask:0xf\n\t" \
                              "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
#define ICIKT_WROUND(xa, xb)                                                                  \
  "v_cmp_lt_u32_e32 vcc, " xa ", " xb "\n\tv_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"        \
  ICIKT_WROR(1, xa, xb) ICIKT_WROR(2, xa, xb) ICIKT_WROR(3, xa, xb) ICIKT_WROR(4, xa, xb)     \
  ICIKT_WROR(5, xa, xb) ICIKT_WROR(6, xa, xb) ICIKT_WROR(7, xa, xb)
// per-lane share of #{rows a before row j in the step : q_a < lo_j}; only the sum over the wave is meaningful
__device__ __forceinline__ uint32_t wave_allpairs(uint32_t q, uint32_t lo, uint32_t lane) {
  const uint32_t nlo = ~lo;
  const uint32_t off1 = (lane & 16u) ? nlo : q;   // round 1: rows 1, 3 are the later ones
  const uint32_t off2 = (lane & 32u) ? nlo : q;   // rounds 2, 3: rows 2, 3
  const uint32_t xa1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)wave_cross_idx(lane, 1u), (int)off1);
  const uint32_t xa2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)wave_cross_idx(lane, 2u), (int)off2);
  const uint32_t xa3 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)wave_cross_idx(lane, 3u), (int)off2);
  const uint32_t xb1 = ~off1, xb2 = ~off2;
  uint32_t acc = 0, junk;
  asm volatile("s_nop 1\n\t"
               ICIKT_HSHR(1) ICIKT_HSHR(2) ICIKT_HSHR(3) ICIKT_HSHR(4) ICIKT_HSHR(5) ICIKT_HSHR(6) ICIKT_HSHR(7)
               ICIKT_HSHR(8) ICIKT_HSHR(9) ICIKT_HSHR(10) ICIKT_HSHR(11) ICIKT_HSHR(12) ICIKT_HSHR(13)
               ICIKT_HSHR(14) ICIKT_HSHR(15)
               ICIKT_WROUND("%4", "%5") ICIKT_WROUND("%6", "%7") ICIKT_WROUND("%8", "%7")
               : "+v"(acc), "=&v"(junk)
               : "v"(q), "v"(lo), "v"(xa1), "v"(xb1), "v"(xa2), "v"(xb2), "v"(xa3)
               : "vcc");
  return acc - ((lo != 0u) ? (15u - (lane & 15u)) : 0u);
}

// Two 64-row steps of a one-pair kernel counted together (positions below 2^15, i.e. n <= 32 768): step t in the low
// and step t + 1 in the high 16 bits of every operand, A = 0x7FFF - q, B = lo, A_a + B_j = 0x7FFF + lo_j - q_a with
// bit 15 set <=> q_a < lo_j (see half_step_count).  15 in-row distances and three rounds of row partnerships with 8
// distances each: 39 sums for 2 x 2 016 pairs; the flags of two sums are gathered by one v_perm_b32 and shifted
// into bit vectors (16 sums per vector).  The in-step pairs of a step do not depend on `seen`, so the hot loop
// simply keeps the values of every other step and counts two steps at once: ~105 instead of 2 x 90 instructions.
#define ICIKT_WPACC "v_perm_b32 %1, %1, %2, %6\n\tv_lshrrev_b32_e32 %0, 1, %0\n\tv_and_or_b32 %0, %1, %5, %0\n\t"
#define ICIKT_WPSHR2(a, b) "v_add_u32_dpp %1, %3, %4 row_shr:" #a " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
                           "v_add_u32_dpp %2, %3, %4 row_shr:" #b " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" ICIKT_WPACC
#define ICIKT_WPROR2(a, b) "v_add_u32_dpp %1, %3, %4 row_ror:" #a " row_mask:0xf bank_mask:0xf\n\t" \
                           "v_add_u32_dpp %2, %3, %4 row_ror:" #b " row_mask:0xf bank_mask:0xf\n\t" ICIKT_WPACC
// one round of row partnerships: distance 0 (plain add) and row_ror 1..7 of the partners' offers, 8 sums = 4 gathers
__device__ __forceinline__ uint32_t wave_round_packed(uint32_t xa, uint32_t offer, uint32_t M8, uint32_t SEL) {
  uint32_t v, t1, t2;
  asm volatile("s_nop 1\n\t"
               "v_add_u32_e32 %1, %3, %4\n\t"
               "v_add_u32_dpp %2, %3, %4 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
               "v_perm_b32 %1, %1, %2, %6\n\tv_and_b32_e32 %0, %5, %1\n\t"
               ICIKT_WPROR2(2, 3) ICIKT_WPROR2(4, 5) ICIKT_WPROR2(6, 7)
               : "=&v"(v), "=&v"(t1), "=&v"(t2)
               : "v"(xa), "v"(offer), "s"(M8), "s"(SEL));
  return v;
}
// (in two parts: the operands with the partners' offers -- three ds_bpermute, which queue with the wave's other LDS
//  operations -- and the sums, which need registers only and can run while LDS atomics are in flight)
struct Packed2 { uint32_t A, B, off1, off2, xa1, xa2, xa3; };
__device__ __forceinline__ Packed2 wave_allpairs_packed2_prep(uint32_t rka, uint32_t rkb, uint32_t lane) {
  Packed2 p;
  const uint32_t Q = __builtin_amdgcn_perm(rkb, rka, 0x05040100u);    // q of step t | q of step t + 1 << 16
  const uint32_t LO = __builtin_amdgcn_perm(rkb, rka, 0x07060302u);   // lo likewise
  p.A = 0x7FFF7FFFu - Q; p.B = LO;
  p.off1 = (lane & 16u) ? p.B : p.A;   // round 1: rows 1, 3 are the later ones
  p.off2 = (lane & 32u) ? p.B : p.A;   // rounds 2, 3: rows 2, 3
  p.xa1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)wave_cross_idx(lane, 1u), (int)p.off1);
  p.xa2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)wave_cross_idx(lane, 2u), (int)p.off2);
  p.xa3 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)wave_cross_idx(lane, 3u), (int)p.off2);
  return p;
}
__device__ __forceinline__ uint32_t wave_allpairs_packed2_count(const Packed2& p) {
  const uint32_t A = p.A, B = p.B, off1 = p.off1, off2 = p.off2, xa1 = p.xa1, xa2 = p.xa2, xa3 = p.xa3;
  const uint32_t M8 = 0x80808080u, SEL = 0x07050301u, M16 = 0x80008000u;
  uint32_t v1, t1, t2;
  // the 15 in-row distances: seven gathers of two sums, the last sum alone (its flags, bits 15 and 31, unpermuted)
  asm volatile("s_nop 1\n\t"
               "v_add_u32_dpp %1, %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
               "v_add_u32_dpp %2, %3, %4 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
               "v_perm_b32 %1, %1, %2, %6\n\tv_and_b32_e32 %0, %5, %1\n\t"
               ICIKT_WPSHR2(3, 4) ICIKT_WPSHR2(5, 6) ICIKT_WPSHR2(7, 8) ICIKT_WPSHR2(9, 10) ICIKT_WPSHR2(11, 12)
               ICIKT_WPSHR2(13, 14)
               "v_add_u32_dpp %1, %3, %4 row_shr:15 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
               "v_lshrrev_b32_e32 %0, 1, %0\n\tv_and_or_b32 %0, %1, %7, %0\n\t"
               : "=&v"(v1), "=&v"(t1), "=&v"(t2)
               : "v"(A), "v"(B), "s"(M8), "s"(SEL), "s"(M16));
  const uint32_t r1 = wave_round_packed(xa1, off1, M8, SEL);
  const uint32_t r2 = wave_round_packed(xa2, off2, M8, SEL);
  const uint32_t r3 = wave_round_packed(xa3, off2, M8, SEL);
  return bcnt_acc(r3, bcnt_acc(r2, bcnt_acc(r1, bcnt_acc(v1, 0u))));
}
__device__ __forceinline__ uint32_t wave_allpairs_packed2(uint32_t rka, uint32_t rkb, uint32_t lane) {
  return wave_allpairs_packed2_count(wave_allpairs_packed2_prep(rka, rkb, lane));
}

// ---- two-level prefix for one pair per wave (long columns): O(1) work per step -------------------------------
// The flat prefix above is rebuilt over all W words after every 64-row step: O(n / 64) per step, O(n^2) per pair.
// Here lane l OWNS the IT words [l * IT, (l + 1) * IT) of `seen` (IT = words per lane, even, <= 16) and the count
// below a position is split in three:
//     #{seen bits below pos} = lb[o] + loc[o][j] + popc(seen[w] & below(pos))      w = pos >> 6, o = w / IT, j = w % IT
//   lb[o]      bits in the words of owners < o          (64 x u32)
//   loc[o][j]  bits in words [o * IT, o * IT + j)       (64 x 16 x u16, 32 bytes per owner)
// and all three are kept up to date by the lanes that INSERT, with LDS atomics whose number does not depend on n:
//   * seen: one 32-bit OR per row (as before);
//   * lb:   a 64-bin histogram of the rows' owners (one atomic add per row into `hist`), an exclusive wave scan of it
//           (read-and-clear: one ds_wrxchg per lane), one atomic add per lane into lb;
//   * loc:  a row in word j of its owner adds 1 to loc[o][j'] for every j' > j: the 16 u16 counters of an owner are
//           four u64 words, and "+1 in all fields above f" is ONE 64-bit atomic add of a constant pattern (no carry
//           between the fields: every counter stays below 16 * 64).  ceil(IT / 4) such adds per step.
// A step therefore costs the same at n = 12 000 and at n = 65 535.  After a merge of `pend` into `seen` (a tie group
// of the streamed column closes) everything is recomputed once from the words (tl_rebuild).
struct TwoLevel {
  unsigned long long* seen;
  uint8_t* loc;     // [64][4][4]: per owner and group of four words, bits in the group's words below word jj (<= 192: a byte)
  uint32_t* lb;     // [64]
  uint32_t* hist;   // [64], all zero between steps
  uint16_t* locg;   // [64][4]: per owner, bits in its groups below group g
};
__device__ __forceinline__ TwoLevel tl_view(unsigned long long* seen, uint16_t* spre) {
  TwoLevel t;
  t.seen = seen;
  t.loc = reinterpret_cast<uint8_t*>(spre);
  t.lb = reinterpret_cast<uint32_t*>(spre + 512);
  t.hist = t.lb + 64;
  t.locg = reinterpret_cast<uint16_t*>(t.hist + 64);
  return t;
}
constexpr int TL_BYTES = K1_TL_BYTES;  // loc + lb + hist + locg
// Words per owner: ALWAYS 16 (round 3).  Owner o owns words [16 o, 16 o + 16) whatever the column length, so a word
// splits into (owner, index) by two bit operations and loc / locg / lb are indexed by w, w >> 2 and w >> 4 directly;
// the owners past the column's last word simply own nothing (n = 50 000: 49 of 64 are used).  Rounds 2 used
// ceil(W / 64) words per owner (balanced, but a multiply-shift-multiply-subtract per split, twice per row, and address
// arithmetic on top: ~20 of the hot step's ~134 vector instructions).
__device__ __forceinline__ int tl_items(int) { return 16; }
__device__ __forceinline__ uint32_t tl_magic(int) { return 0u; }
__device__ __forceinline__ void tl_split(uint32_t w, int, uint32_t, uint32_t& o, uint32_t& j) {
  o = w >> 4;
  j = w & 15u;
}

// The counters inside an owner are kept on two levels (round 2b): its <= 16 words are four groups of four, locg[o]
// = four u16 "bits in the groups below g", loc[o][g] = four u16 "bits in the group's words below jj".  A row then
// costs exactly two 64-bit atomic adds whatever IT is -- "+1 in all fields above f" is the constant
// 0x0001000100010000 << 16 f (what leaves the top is the fields that do not exist) -- where one level of sixteen
// counters cost up to four adds in a predicated loop.  A query reads one u16 more.
__device__ __forceinline__ uint32_t tl_query(const TwoLevel& T, uint32_t pos, int IT, uint32_t magic) {
  const uint32_t w = pos >> 6;
  uint32_t o, j;
  tl_split(w, IT, magic, o, j);
  const uint32_t below_group = (IT > 4) ? (uint32_t)T.locg[o * 4u + (j >> 2)] : 0u;   // one group: nothing below it
  return T.lb[o] + below_group + (uint32_t)T.loc[w] + (uint32_t)__popcll(T.seen[w] & low_mask64(pos & 63u));
}

// Rows with ins == true have just been OR-ed into seen at position q (by these lanes): bring lb and loc up to date.
// Every lane of the wave takes part (scan, read-and-clear of its histogram bin).
__device__ __forceinline__ void tl_update(const TwoLevel& T, bool ins, uint32_t q, int IT, uint32_t magic, uint32_t lane) {
  uint32_t o, j;
  tl_split((q & 0xFFFFu) >> 6, IT, magic, o, j);
  if (ins) {
    atomicAdd(&T.hist[o], 1u);
    const unsigned long long ABOVE = 0x0001000100010000ull;
    // "+1 in the bytes above byte f" of the group's four counters: one 32-bit add (what leaves the top: nothing to count)
    atomicAdd(reinterpret_cast<uint32_t*>(T.loc) + o * 4u + (j >> 2), 0x01010100u << (8u * (j & 3u)));
    if (IT > 4) atomicAdd(reinterpret_cast<unsigned long long*>(T.locg) + o, ABOVE << (16u * (j >> 2)));
  }
  wave_lds_fence();
  const uint32_t h = atomicExch(&T.hist[lane], 0u);
  const uint32_t below = wave_incl_scan(h) - h;
  if (below != 0u) atomicAdd(&T.lb[lane], below);
}

// the per-row part of tl_update alone (the histogram is collected by the caller: two pairs per wave issue theirs together)
__device__ __forceinline__ void tl_update_rows(const TwoLevel& T, uint32_t q) {
  const uint32_t w = (q & 0xFFFFu) >> 6, o = w >> 4, j = w & 15u;
  atomicAdd(&T.hist[o], 1u);
  atomicAdd(reinterpret_cast<uint32_t*>(T.loc) + (w >> 2), 0x01010100u << (8u * (j & 3u)));
  atomicAdd(reinterpret_cast<unsigned long long*>(T.locg) + o, 0x0001000100010000ull << (16u * (j >> 2)));
}

// Recompute loc, locg and lb from the words of seen (after a long tie group's rows have been OR-ed into it).
// NW: words of seen that exist (even: k1_lds_stride); owners past them own nothing.
__device__ __forceinline__ void tl_rebuild(const TwoLevel& T, int NW, uint32_t lane) {
  constexpr int IT = 16;
  const uint32_t base = lane * (uint32_t)IT;
  ulonglong2* b2 = reinterpret_cast<ulonglong2*>(T.seen + base);
  uint16_t* l16 = reinterpret_cast<uint16_t*>(T.loc + lane * 16u);
  uint32_t run = 0, grun = 0;          // bits in the owner's words so far; of them, in the groups before this one
  unsigned long long gpack = 0ull;     // locg[lane]: four u16
  for (int i = 0; i < (IT >> 1); ++i) {
    const int w = (int)base + 2 * i;
    ulonglong2 v = make_ulonglong2(0ull, 0ull);
    if (w < NW) v = b2[i];
    if ((i & 1) == 0) {  // words 2i, 2i+1 open group i / 2
      grun = run;
      gpack |= (unsigned long long)grun << (16 * (i >> 1));
    }
    const uint32_t c0 = run - grun;
    run += (uint32_t)__popcll(v.x);
    const uint32_t c1 = run - grun;
    run += (uint32_t)__popcll(v.y);
    l16[i] = (uint16_t)(c0 | (c1 << 8));
  }
  reinterpret_cast<unsigned long long*>(T.locg)[lane] = gpack;
  T.lb[lane] = wave_incl_scan(run) - run;
}

// inclusive prefix sum inside each 32-lane half
__device__ __forceinline__ uint32_t half_incl_scan(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142 /*row_bcast:15*/, 0xa, 0xf, false);
  return v;
}

// Prefix of one bitset per half: lane l (0..31) of a half owns words [l*HI, (l+1)*HI) and keeps their INCLUSIVE
// prefix counts (bits in words 0 .. w) in ONE aligned 16-byte slot of `pre` (8 x u16, HI <= 5 of them used): HI u16
// values at a stride of 2*HI bytes made one misaligned 8-byte store plus a 2-byte one, and that store alone kept
// the LDS unit busy.  Word w lives in slot w / HI, entry w % HI.  A query is
// `entry - popcount(word >> bit)`: the shift takes its count from the low six bits of the position, no mask is built.
template <int HI> constexpr int k1_half_slot() { return HI > 8 ? 16 : 8; }   // u16 entries per lane slot
template <int HI> constexpr int k1_half_pre_bytes() { return 32 * k1_half_slot<HI>() * 2; }
template <int HI>
__device__ __forceinline__ uint32_t half_pre_index(uint32_t w) {
  const uint32_t o = w / (uint32_t)HI;
  return o * (uint32_t)k1_half_slot<HI>() + (w - o * (uint32_t)HI);
}
template <int HI>
__device__ __forceinline__ uint32_t prefix_query_half(const unsigned long long* bits, const uint16_t* pre, uint32_t pos) {
  const uint32_t w = pos >> 6;
  // the entry's BYTE address, as the hot step computes it: slot w / HI at 2 * slot-size bytes, entry w % HI at 2 ->
  // 2 w + (2 slot - 2 HI) (w / HI), with w / HI = (w * (65536 / HI + 1)) >> 16 -- exact for w < 32 HI, HI <= 15 (the
  // generic form, a division by a constant and a remainder, cost four vector instructions more per query; a GROUP step
  // in row mode makes eight to sixteen queries)
  const uint32_t d = (w * (65536u / (uint32_t)HI + 1u)) >> 16;
  const uint32_t e = (uint32_t)*reinterpret_cast<const uint16_t*>(
      reinterpret_cast<const unsigned char*>(pre) + 2u * w + (2u * (uint32_t)k1_half_slot<HI>() - 2u * (uint32_t)HI) * d);
  return e - (uint32_t)__popcll(bits[w] >> (pos & 63u));
}
// counts of the HI words a lane holds in registers -> the lane's slot; cw[i] = bits in words 0 .. i of the lane,
// incl = the half's inclusive scan of cw[HI - 1]
template <int HI>
__device__ __forceinline__ void half_pre_store(uint16_t* pre, uint32_t l, uint32_t incl, const uint32_t (&cw)[HI]) {
  const uint32_t excl = incl - cw[HI - 1];
  const uint32_t both = excl * 0x10001u;              // added to two packed counts at once (no carry: a count is at most n <= 18 336 < 2^16)
  uint32_t v[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (HI == 2 * i + 1) v[i] = incl;                                     // the lane's last word, alone in its dword
    if (HI >= 2 * i + 2) v[i] = pack16(cw[(2 * i + 1) % HI], cw[(2 * i) % HI]) + both;
  }
  uint4* slot = reinterpret_cast<uint4*>(pre + l * (uint32_t)k1_half_slot<HI>());
  if (HI > 4) slot[0] = make_uint4(v[0], v[1], v[2], v[3]);
  else *reinterpret_cast<uint2*>(slot) = make_uint2(v[0], v[1]);
  if (HI > 8) slot[1] = make_uint4(v[4], v[5], v[6], v[7]);
}
template <int HI>
__device__ __forceinline__ void rebuild_prefix_half(const unsigned long long* bits, uint16_t* pre, uint32_t l) {
  const uint32_t base = l * (uint32_t)HI;
  uint32_t c[HI], run = 0;
#pragma unroll
  for (int i = 0; i < HI; ++i) {
    run = bcnt64_acc(bits[base + i], run);
    c[i] = run;
  }
  half_pre_store<HI>(pre, l, half_incl_scan(run), c);
}

// state of one pair inside a wave
struct PairState {
  WaveLds L;
  uint32_t dis, tie, tie2;
};

// ---- tie steps of the whole-wave kernels ------------------------------------------------------------------------
//   MIXED  up to 64 rows of COMPLETE groups of at most k1_ks(false) rows each (seg_mixed_step).  The plain all-pairs
//          count of the step also counts pairs inside a group; those are the pairs (lane - d, lane), d = 1 .. group
//          size - 1, so d whole-wave shifts of q and lo give, per lane, the spurious count (q_prev < lo_me) and the
//          joint ties (lo_prev == lo_me).  The loop ends with the largest group of the step: no sort.
//   GROUP  up to 64 rows of ONE group (a piece of it, or all of it): in k1_pairs, the design of the half-wave kernels.
typedef __attribute__((address_space(3))) unsigned long long* lds_u64p;
typedef __attribute__((address_space(3))) uint16_t* lds_u16p;
// MIXED steps take groups of up to this many rows (the correction loop costs ~10 instructions per row of the
// step's largest group); longer groups get GROUP steps of their own.  A whole wave amortises a longer loop.
__host__ __device__ constexpr int k1_ks(bool half) { return half ? 32 : 32; }

struct SegState {   // LDS views of a pair
  lds_u64p seen;
  lds_u16p spre;
};
struct SegCounts { uint32_t dis, neg, tie, tie2, cfill; };

template <int SEG, int HI>
__device__ __forceinline__ uint32_t seg_query(unsigned long long* seen, uint16_t* spre, uint32_t lo, int IT, uint32_t magic) {
  return (SEG == 64) ? tl_query(tl_view(seen, spre), lo, IT, magic) : prefix_query_half<(HI > 0 ? HI : 1)>(seen, spre, lo);
}

// (inlined since the end of round 4: as a function of its own it began, like every callee, by waiting for ALL outstanding loads
//  -- the next step's gather and the ring reload the caller had just issued: a memory round trip per MIXED step)
template <int SEG, int HI>
__device__ __forceinline__ SegCounts seg_mixed_step(const SegState st, const unsigned long long F_in, const int nact_in,
                                                              const uint32_t rk, const int IT_in, const uint32_t magic_in,
                                                              const uint32_t lane) {
  // (when this was an out-of-line function its arguments arrived in vector registers; the wave-uniform ones go back to scalar
  //  registers -- harmless now -- or everything derived from them would run on the vector unit)
  const unsigned long long F = uniform_u64(F_in);
  const int nact = __builtin_amdgcn_readfirstlane(nact_in);
  const int IT = __builtin_amdgcn_readfirstlane(IT_in);
  const uint32_t magic = (uint32_t)__builtin_amdgcn_readfirstlane((int)magic_in);
  unsigned long long* seen = (unsigned long long*)st.seen;
  uint16_t* spre = (uint16_t*)st.spre;
  const uint32_t sl = (SEG == 32) ? (lane & 31u) : lane;
  const bool valid = (int)sl < nact;
  const uint32_t q = valid ? (rk & 0xFFFFu) : 0xFFFFFFFFu;  // never "below" anything
  const uint32_t lo = valid ? (rk >> 16) : 0u;              // nothing is below 0
  SegCounts c;
  c.tie2 = 0; c.cfill = 0;
  const uint32_t cnt = seg_query<SEG, HI>(seen, spre, lo, IT, magic);
  c.dis = (valid ? cnt : 0u) + ((SEG == 64) ? wave_allpairs(q, lo, lane) : half_count(q, lo, lane));
  // Pairs inside a group: lane l against lanes l - 1 .. l - (its group's rows before it).  Both operands carry the
  // group in their upper half, counted DOWN (63 - number of the group), so that a lane of an earlier group that is
  // shifted in compares as greater and never counts: no mask per shift.  The loop runs to the step's largest group.
  const unsigned long long Fs = (SEG == 32) ? (F & 0xFFFFFFFFull) : F;
  const unsigned long long upto = Fs & ((2ull << sl) - 1ull);                   // group starts at or before me
  // 64 - (group number 1 .. 64); the first half of a half-wave kernel is tagged above it, so that what the shifts
  // carry from pair 0's lanes into pair 1's compares as greater too
  const uint32_t gdown = ((64u - (uint32_t)__popcll(upto)) << 16) | ((SEG == 32 && lane < 32u) ? 0x00800000u : 0u);
  const uint32_t klo = valid ? (gdown | lo) : 0u;
  // longest run of rows that continue a group = largest group - 1 (wave-uniform, from the flags)
  // (without a loop: runs of 2, 4, .. 32 by doubling, then the longest run bit by bit from the top -- R(a + b) = R(a) & R(b) >> a)
  int dmax = 0;
  {
    const unsigned long long z = ~Fs & ((nact >= 64) ? ~0ull : ((1ull << nact) - 1ull));
    const unsigned long long p2 = z & (z >> 1), p4 = p2 & (p2 >> 2), p8 = p4 & (p4 >> 4), p16 = p8 & (p8 >> 8), p32 = p16 & (p16 >> 16);
    unsigned long long cur = ~0ull, t;
    t = cur & p32;             if (t != 0ull) { cur = t; dmax = 32; }
    t = cur & (p16 >> dmax);   if (t != 0ull) { cur = t; dmax += 16; }
    t = cur & (p8 >> dmax);    if (t != 0ull) { cur = t; dmax += 8; }
    t = cur & (p4 >> dmax);    if (t != 0ull) { cur = t; dmax += 4; }
    t = cur & (p2 >> dmax);    if (t != 0ull) { cur = t; dmax += 2; }
    t = cur & (z >> dmax);     if (t != 0ull) { cur = t; dmax += 1; }
  }
  // (round 4: the tie groups of the gathered column are disjoint ranges of positions and lo is the start of a row's range, so
  //  q_prev < lo_me <=> lo_prev < lo_me: ONE shifted operand serves both counts.  The shift runs in place -- lane 0 is never
  //  written and keeps the "greater than anything" of the first shift, which the later shifts carry upwards -- and the loop is
  //  five vector instructions per distance; the compiler's version of the two-operand loop took ten, four of them copies.)
  uint32_t slo = dpp_wave_shr1(0xFFFFFFFFu, valid ? klo : 0xFFFFFFFFu);   // lane l holds lane l - 1
  uint32_t spur = 0, tie = 0;
  asm volatile("s_nop 1" : "+v"(slo));      // (a DPP operand written by the previous vector instruction: two wait states)
  // (four distances per turn: a distance beyond the step's largest group counts nothing -- only lanes of earlier groups arrive)
#define ICIKT_SHIFT_DIST "v_cmp_lt_u32_e32 vcc, %0, %3\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t" /* same group (equal upper halves), the earlier row's tie group below mine */ \
                         "v_cmp_eq_u32_e32 vcc, %0, %3\n\tv_addc_co_u32_e32 %2, vcc, 0, %2, vcc\n\t" /* same group, same tie group of the gathered column */ \
                         "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"               /* the next distance */
  for (int d = 0; d < dmax; d += 4) {
    asm volatile(ICIKT_SHIFT_DIST ICIKT_SHIFT_DIST ICIKT_SHIFT_DIST ICIKT_SHIFT_DIST
                 : "+v"(slo), "+v"(spur), "+v"(tie) : "v"(klo) : "vcc");
  }
#undef ICIKT_SHIFT_DIST
  if (!valid) { spur = 0; tie = 0; }
  c.neg = spur;
  c.tie = tie;
  wave_lds_fence();
  if (valid) seen_insert(seen, q);
  if (SEG == 64) {
    tl_update(tl_view(seen, spre), valid, q, IT, magic, lane);
  } else {
    wave_lds_fence();
    rebuild_prefix_half<(HI > 0 ? HI : 1)>(seen, spre, sl);
  }
  wave_lds_fence();
  return c;
}

// Diagnostic build only (-DICIKT_STEP_STATS, tools/step_stats.py; in the product build no stamp executes): per step
// kind the steps taken, their rows and the wave cycles (s_memtime) spent in them, summed over all waves.
//   kind 0 hot loop   1 hot step met in the main loop   2 MIXED   3 GROUP (with its closes)   4 general step (pend in
//   global memory)    5 closed-form tail   6 task set-up and final reductions
#ifdef ICIKT_STEP_STATS
__device__ unsigned long long g_step_stats[24];
#define ICIKT_ST_DECL unsigned long long st_t = __builtin_amdgcn_s_memtime(); \
  unsigned long long st_c0 = 0, st_c1 = 0, st_c2 = 0, st_c3 = 0, st_c4 = 0, st_c5 = 0, st_c6 = 0, st_c7 = 0; \
  unsigned st_s0 = 0, st_s1 = 0, st_s2 = 0, st_s3 = 0, st_s4 = 0, st_s5 = 0, st_s6 = 0, st_s7 = 0; \
  unsigned st_r0 = 0, st_r1 = 0, st_r2 = 0, st_r3 = 0, st_r4 = 0, st_r5 = 0, st_r6 = 0, st_r7 = 0;
#define ICIKT_ST_MARK(K, ROWS) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_c##K += t_ - st_t; st_t = t_; \
  st_s##K += 1u; st_r##K += (unsigned)(ROWS); }
#define ICIKT_ST_FLUSH1(K) { atomicAdd(&g_step_stats[K], (unsigned long long)st_s##K); atomicAdd(&g_step_stats[8 + K], (unsigned long long)st_r##K); \
  atomicAdd(&g_step_stats[16 + K], st_c##K); }
#define ICIKT_ST_FLUSH if (lane == 0u) { ICIKT_ST_FLUSH1(0) ICIKT_ST_FLUSH1(1) ICIKT_ST_FLUSH1(2) ICIKT_ST_FLUSH1(3) ICIKT_ST_FLUSH1(4) \
  ICIKT_ST_FLUSH1(5) ICIKT_ST_FLUSH1(6) ICIKT_ST_FLUSH1(7) }
#else
#define ICIKT_ST_DECL
#define ICIKT_ST_MARK(K, ROWS)
#define ICIKT_ST_FLUSH
#endif

// Variants: <2, HI> two pairs per wave, one per 32-lane half, HI = 1..ICIKT_HALF_ITEMS_MAX words per lane in a half's
// prefix rebuild (n <= 18 336); <2, 0> two pairs per wave, one after the other on the whole wave (long columns); <1, 0>
// one pair on the whole wave (any n).
// A task is (pair, pair or -1).  The two pairs of a task share their STREAMED column (pj) and their gathered
// columns (pi) are the two columns of one rec block, so one 8-byte gather per row serves both (host:
// build_units).
template <int NP, int HI> __host__ __device__ constexpr bool half_mode_of() { return (HI > 0) && (NP == 2); }
// a pair's counter table (count mode): LDS atomics are 32 bits wide, so 2-byte counters are added to as halves of a dword (a
// count stays below 2^16: n <= 30 656)
#if ICIKT_CNT_BYTES == 2
typedef uint16_t cnt_t;
__device__ __forceinline__ void cnt_inc(cnt_t* t, uint32_t g) { atomicAdd(reinterpret_cast<uint32_t*>(t) + (g >> 1), 1u << ((g & 1u) << 4)); }
#else
typedef uint32_t cnt_t;
__device__ __forceinline__ void cnt_inc(cnt_t* t, uint32_t g) { atomicAdd(t + g, 1u); }
#endif
template <int NP, int HI>
__global__ void __launch_bounds__(512, (NP == 2 && HI == 0) ? 3 : (HI > 9) ? 5 : 6)  // 6 waves per SIMD (<= 80 VGPRs); the LDS state of a pair (seen + prefix slots) allows that up to HI = 9.
                                                                     // Two long-column pairs per wave: the LDS state allows 2-3 waves per SIMD, 3 leave 168 VGPRs
k1_pairs(PrepView pv, const int32_t* __restrict__ tasks, int n_tasks,
         const int32_t* __restrict__ pi, const int32_t* __restrict__ pj, PairRaw* __restrict__ raw,
         int perpair_bytes, int* __restrict__ task_ctr, int opts) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  // XCD-aware mapping: consecutive tasks share their gathered block, so keep them on one XCD
  // (workgroups are dealt round-robin over the 8 XCDs).  Bijective for any grid size.
  int blk;
  {
    const int nwg = (int)gridDim.x, orig = (int)blockIdx.x;
    const int qd = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    blk = ((xcd < r) ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + (orig >> 3);
  }
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // SGPR: keeps every per-wave pointer scalar
  const int wpb = blockDim.x >> 6;
  const int gwave = blk * wpb + wave;             // this wave's slot among the launched waves
  const int nwaves = (int)gridDim.x * wpb;

  const int n = pv.n, W = pv.W, Wp = pv.Wp;
  // (from the exec-mask count, not from threadIdx.x: the thread id would stay live -- and spill -- to the task's end)
  const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int Wp4 = k1_lds_stride(Wp, HI);                    // stride of the per-pair arrays (host: plan_k1)
  const int IT = tl_items(Wp);                              // one pair per wave: words of seen owned by a lane
  const uint32_t magic = tl_magic(IT);
  static_assert((NP == 1 && HI == 0) || NP == 2, "k1_pairs variants");
  constexpr bool half_mode = (HI > 0) && (NP == 2);         // HI = words per lane when a half rebuilds a prefix
  // joint ties of a tie group of more than 32 rows, by the gathered column's tie groups: list mode (range counts per listed
  // group) up to tg_list of them (<= 128), count mode (half-wave kernels: a counter per tie group) up to tg_max, else row mode
  const int tg_list = (opts >> 8) & 0x3FF;
  // (whole-wave kernels, round 4: a counter table too -- count mode for gathered columns of more than tg_list and more than
  //  K1_CNT_MIN_GROUPS tie groups, up to the table's entries; the pre-pass writes girow for exactly those long columns)
  const int tg_max = half_mode_of<NP, HI>() ? (opts >> 18) : max(tg_list, opts >> 18);
  const int cnt_cap = opts >> 18;   // entries of a pair's counter table (0: none)
  // listed tie groups per lane of a half in list mode: 4 (tg_list <= 128), 8 in the kernels of 11 .. 15 words per lane
  // (tg_list <= 256: they have the registers -- five waves per SIMD -- and the LDS beside their 4.9 KB of state per pair
  // leaves count mode 64 counters)
  constexpr int LISTK = (HI > 9) ? 8 : 4;

  // Half-wave kernels: the grid covers the task list and a wave takes exactly one task -- without a task loop the
  // compiler has nothing to hoist out of it, and the ~20 per-lane addresses it used to keep across the loop in scratch
  // (5.6 KB of spill stores per wave, 1.5 GB per c4 launch) are gone.
  // Whole-wave kernels with more tasks than the chip holds waves (the host passes task counters): PERSISTENT waves, the
  // grid is what the chip holds and a wave FETCHES its next task from a counter -- a wave that has finished goes on at
  // once instead of waiting for the other waves of its workgroup to retire.  Workgroups are dealt round-robin over the 8
  // XCDs, so the workgroups with equal blockIdx % 8 share an XCD: each such group owns one contiguous eighth of the task
  // list and its own counter.  Tasks are then started in order, as a grid that covers the task list starts its workgroups
  // in order: the waves of an XCD stay within a few hundred consecutive tasks = one or two gathered blocks (400 KB each
  // at n = 50 000) that live in its 4 MB L2.  (Striding over the task list instead lets the waves drift apart by tens of
  // rounds on a 2-million-task list; measured on the full c5 matrix: 55 % L2 misses, 7 TB fetched, 2.0e6 pairs/s,
  // against 2 % misses for a grid that covers the list.)
  // The kernel has no workgroup barrier, so the waves of a workgroup run independently.
  int task = gwave, t_lo = 0, t_hi = n_tasks;
  int* my_ctr = task_ctr;
  const bool persist = (HI == 0) && task_ctr != nullptr;   // wave-uniform
  if (persist) {
    const int ng = min(8, (int)gridDim.x);
    const int xg = (int)blockIdx.x % ng;
    const int chunk = (n_tasks + ng - 1) / ng;
    t_lo = xg * chunk;
    t_hi = min(n_tasks, t_lo + chunk);
    my_ctr = task_ctr + xg;
    int t = 0;
    if (lane == 0u) t = atomicAdd(my_ctr, 1);
    task = t_lo + __builtin_amdgcn_readfirstlane(t);
  }
  // Half-wave kernels, a task list that leaves the chip half empty (the host decides: opts bits 4..5): every task is CUT in
  // 2 or 4 SEGMENTS at the streamed column's segment marks (k0_tie_program) and a wave takes one segment -- it first inserts
  // the rows in front of its segment into `seen` without counting them (an eighth of a step's work per 64 rows), then walks
  // its part; the segments' counts are added up in `raw` (zeroed by the host).  A launch of 2 280 tasks on 1 024 SIMDs lasts
  // as long as ONE task, whatever else is done to the steps: two segments halve that.
  const int split = half_mode_of<NP, HI>() ? (1 << ((opts >> 4) & 3)) : 1;
  int seg = 0;
  if (split > 1) {
    seg = gwave / n_tasks;
    task = gwave - seg * n_tasks;
    if (seg >= split) return;
  }
  if (task >= t_hi) return;
  do {
  ICIKT_ST_DECL
  int pidx[NP];
  pidx[0] = __builtin_amdgcn_readfirstlane(tasks[2 * task]);
  const int p_second = __builtin_amdgcn_readfirstlane(tasks[2 * task + 1]);
  const int np = (NP == 2 && p_second >= 0) ? 2 : 1;
  if (NP == 2) pidx[NP - 1] = (p_second >= 0) ? p_second : pidx[0];  // an unused slot repeats the pair; not written

  // streamed column (shared by the task's pairs): rows in descending order, group-start flags
  const int scol = __builtin_amdgcn_readfirstlane(pj[pidx[0]]);
  const uint16_t* ord = pv.order + (int64_t)scol * pv.n_ord;
  const uint32_t* ord_w = pv.order_w ? pv.order_w + (int64_t)scol * pv.n_ord : nullptr;   // (long columns: the same as 32-bit words)
  const unsigned long long* gf = pv.col_gflag(scol);
  const unsigned long long* ma = pv.col_mask(scol);
  const unsigned long long* fa = pv.col_fillmask(scol);

  // gathered (random-access) columns: the pairs' pi, both in one block of the interleaved rec table.  The waves
  // of a workgroup, and the workgroups of an XCD, mostly share the block, which keeps it in L1 / L2.
  uint32_t comp[NP];           // which column of the block a pair reads
  const uint16_t* hiG[NP];
  const uint32_t* tgB[NP];     // few tie groups in the gathered column: joint ties of multi-step groups are
  int ntgB[NP];                // counted at group close from this list (-1: row mode)
  bool cntB[NP];               // half-wave kernels: ... or in the pair's counter table (count mode), one counter per listed group
  PairState S[NP];
  uint32_t cb[NP], gg[NP];
  bool g_oddtie = false;       // a gathered column has a tie group that starts at an odd position
  const uint32_t* rec_blk;
  const uint32_t* hi_blk;   // the block's tie-group ends: hi of column 2a | hi of column 2a + 1 << 16 per row
  const uint32_t* gi_blk;   // the block's tie-group indices (PrepView::girow), interleaved the same way
  {
    const int g0 = __builtin_amdgcn_readfirstlane(pi[pidx[0]]);
    rec_blk = pv.rec + ((int64_t)(g0 >> 1) * pv.rec_rows) * 2;
    hi_blk = reinterpret_cast<const uint32_t*>(pv.hirow + ((int64_t)(g0 >> 1) * pv.rec_rows) * 2);
    gi_blk = reinterpret_cast<const uint32_t*>(pv.girow + ((int64_t)(g0 >> 1) * pv.rec_rows) * 2);
  }
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int gcol = __builtin_amdgcn_readfirstlane(pi[pidx[k]]);
    comp[k] = (uint32_t)(gcol & 1);
    hiG[k] = pv.hirow + ((int64_t)(gcol >> 1) * pv.rec_rows) * 2 + (gcol & 1);   // interleaved like rec: index 2 * row
    tgB[k] = pv.tgroups + (int64_t)gcol * pv.tg_stride;
    const int ntg_raw = __builtin_amdgcn_readfirstlane((int)pv.col_stats(gcol)->ntg);
    // half-wave kernels: COUNT MODE while the gathered column's tie groups have a counter each in the pair's table (tg_max =
    // its entries, sized by the host to what the LDS holds at the launch's occupancy); whole-wave kernels: list mode with
    // at most two listed groups per lane
    cntB[k] = half_mode ? (ntg_raw > tg_list) : (ntg_raw > max(tg_list, K1_CNT_MIN_GROUPS));    // (read where ntgB[k] >= 0)
    ntgB[k] = (!(opts & 2) && ntg_raw <= ((half_mode || cntB[k]) ? max(tg_list, tg_max) : tg_list)) ? ntg_raw : -1;
    g_oddtie = g_oddtie || (__builtin_amdgcn_readfirstlane(pv.col_stats(gcol)->flags) & COL_ODD_TIE) != 0;
    const unsigned long long* mb = pv.col_mask(gcol);
    const unsigned long long* fb = pv.col_fillmask(gcol);

    unsigned char* wbase = smem + ((size_t)wave * NP + k) * (size_t)perpair_bytes;
    S[k].L.seen = reinterpret_cast<unsigned long long*>(wbase);
    if (HI == 0) {
      // a pair on the whole wave: seen | two-level counts (loc, lb, hist, locg: TL_BYTES); L.spre = the counts.  No `pend`
      // (round 3): as in the half-wave kernels a tie group of the streamed column is queried first and inserted when it closes
      S[k].L.spre = reinterpret_cast<uint16_t*>(S[k].L.seen + Wp4);
      for (int w = lane; w < Wp4; w += 64) S[k].L.seen[w] = 0ull;
      uint32_t* tl32 = reinterpret_cast<uint32_t*>(S[k].L.spre);
      // (... | the counters of count mode, as in the half-wave kernels: cnt_cap + 2 entries, all zero between groups)
      for (int w = lane; w < (TL_BYTES + (cnt_cap > 0 ? (cnt_cap + 2) * ICIKT_CNT_BYTES + 3 : 0)) / 4; w += 64) tl32[w] = 0u;
    } else {
      // half-wave kernels: seen | prefix slots of seen (32 lanes x 16 B).  No `pend`: the rows of a tie group of the
      // streamed column are queried first and inserted when the group closes (GROUP steps below)
      S[k].L.spre = reinterpret_cast<uint16_t*>(S[k].L.seen + Wp4);
      for (int w = lane; w < Wp4; w += 64) S[k].L.seen[w] = 0ull;
      for (int w = lane; w < k1_half_pre_bytes<(HI > 0 ? HI : 1)>() / 2; w += 64) S[k].L.spre[w] = 0;
      // ... | the counters of count mode: tg_max + 1 u16 (the last one takes the rows that are their own tie group)
      uint32_t* cz = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(S[k].L.spre) + k1_half_pre_bytes<(HI > 0 ? HI : 1)>());
      for (int w = lane; w < ((tg_max + 2) * ICIKT_CNT_BYTES + 3) / 4; w += 64) cz[w] = 0u;   // (the host rounds a pair's bytes up to 16)
    }
    S[k].dis = 0; S[k].tie = 0; S[k].tie2 = 0;
    // both-missing count and the (fill group, fill group) cell: bitset AND + popcount
    cb[k] = 0; gg[k] = 0;
    if (seg == 0) {
      for (int w = lane; w < W; w += 64) {
        cb[k] += (uint32_t)__popcll(ma[w] & mb[w]);
        gg[k] += (uint32_t)__popcll(fa[w] & fb[w]);
      }
    }
  }
  wave_lds_fence();

  // Rows of the streamed column are loaded THREE steps ahead (ring r0, r1, r2: positions pos, pos + 64, pos + 128
  // + lane) and the rec gather of a step is issued one step ahead, while the previous step is still being
  // counted: with the LDS state of a pair limiting long columns to 3 .. 4 waves per SIMD, a step's chain
  // "order row -> rec gather -> LDS query" would otherwise expose two memory latencies per step (measured: 356 ns
  // per step and SIMD at n = 50 000 against 266 ns of issue time).  The ring assumes 64-row steps; a shorter step
  // reloads it (order[] is zero padded for the run-ahead: PrepView::n_ord).
  uint32_t r0 = gload_u16(ord, lane), r1 = gload_u16(ord, 64u + lane), r2 = gload_u16(ord, 128u + lane);
  uint32_t rk_pre[NP];   // the rec values of the step that starts at pos, gathered during the previous step
  uint2 rv_pre = make_uint2(0u, 0u);   // (two pairs per wave: the 8-byte gather as it arrives; a pair takes its column only where
                                       //  the values are USED, so that no wait for the gather sits right behind its issue)
  bool rk_ok = false;
#pragma unroll
  for (int k = 0; k < NP; ++k) rk_pre[k] = 0u;
  uint32_t hi_pre = 0;   // fast tie steps: hirow of the next step's rows, gathered one step ahead in tie regions
  uint32_t gi_pre = 0;   // half-wave kernels: likewise girow (pairs in count mode)
  bool hi_ok = false;

  // The LAST tie group of the streamed column (on data with missing values: the fill group) in closed form.
  // Every row outside it is above it, so for a row r of the group
  //   #{j above : b_j < b_r} = lo_r - #{j in the group : b_j < b_r},
  // and summed over the group:  sum(lo_r) - (C(m, 2) - T),  m = rows of the group, T = its joint ties.
  // The group's rows then only gather and add lo; T needs no insertion either: every row that is not in `seen` when
  // the group begins belongs to it, so it has (size of g) - (rows of g in seen) rows in a listed tie group g of the
  // gathered column.  Used when the gathered columns are in list mode and the group is longer than one step: the
  // step loop ends at its first position last_start and a gather-only loop runs the rest.
  // half-wave kernels: both gathered columns have a counter per tie group in their pair's table
  // (SOLO steps read the table as bytes: solo_cap counters)
  const uint32_t solo_cap = (uint32_t)max(0, (tg_max + 1) * ICIKT_CNT_BYTES);
  bool solo_ok = half_mode && !(opts & 2) && !(opts & 8);
#pragma unroll
  for (int k = 0; k < NP; ++k) solo_ok = solo_ok && (pv.col_stats(__builtin_amdgcn_readfirstlane(pi[pidx[k]]))->ntg <= solo_cap);
  int last_start;
  bool closed_form = true;
#pragma unroll
  for (int k = 0; k < NP; ++k) closed_form = closed_form && (ntgB[k] >= 0);
  int hot_until;  // steps [pos, pos + 64) with pos + 64 <= hot_until hold singleton groups only: no flag work at all
  {
    int best = 0, first_cont = n;  // highest group start; first position that continues a group
    for (int w = lane; w < W; w += 64) {
      unsigned long long f = gf[w];
      unsigned long long z = ~f;
      if (w == W - 1 && (n & 63)) { f &= (1ull << (n & 63)) - 1ull; z &= (1ull << (n & 63)) - 1ull; }
      if (f != 0ull) best = max(best, w * 64 + 63 - (int)__builtin_clzll(f));
      if (z != 0ull) first_cont = min(first_cont, w * 64 + (int)__builtin_ctzll(z));
    }
    last_start = __builtin_amdgcn_readfirstlane(wave_max_i32(best));
    first_cont = -__builtin_amdgcn_readfirstlane(wave_max_i32(-first_cont));
    hot_until = (first_cont < n) ? first_cont - 1 : n;
    closed_form = closed_form && (n - last_start > 64);
  }
  // the wave's segment of the walk: [seg_begin, seg_end), seg_step = the program step at seg_begin (-1: the segment begins
  // in the singleton region, or at the program's first step)
  int seg_begin = 0, seg_end = n, seg_step = -1;
  if constexpr (half_mode) {
    if (split > 1) {
      const uint32_t* tail = pv.tprog + (int64_t)scol * pv.tp_stride + (pv.tp_stride - 2 * TPROG_MARKS);
      const uint32_t mv = gload_u32(tail, lane & 7u);
      // split 2: mark 1; split 4: marks 0, 2, 3 (a missing mark, 0xFFFFFFFF: the walk ends before it -> n)
      auto mark = [&](int j, int& mp, int& ms) {
        mp = (int)min((uint32_t)__builtin_amdgcn_readlane((int)mv, j), (uint32_t)n);
        ms = __builtin_amdgcn_readlane((int)mv, 4 + j);
      };
      int b1 = n, b2 = n, b3 = n, s1 = -1, s2 = -1, s3 = -1;
      if (split == 2) mark(1, b1, s1);
      else { mark(0, b1, s1); mark(2, b2, s2); mark(3, b3, s3); }
      seg_begin = (seg == 0) ? 0 : (seg == 1) ? b1 : (seg == 2) ? b2 : b3;
      seg_end = (seg == 0) ? b1 : (seg == 1) ? b2 : (seg == 2) ? b3 : n;
      seg_step = (seg == 0) ? -1 : (seg == 1) ? s1 : (seg == 2) ? s2 : s3;
      if (seg > 0 && seg_begin >= seg_end) return;   // an empty segment (segment 0 also writes the pair's missing-row counts)
      closed_form = closed_form && (seg_end >= n);
      hot_until = min(hot_until, seg_end);           // (a mark inside the singleton region is a multiple of 64)
    }
  }
  const int end_main = min(closed_form ? last_start : n, seg_end);

  // Steps are cut at tie-group boundaries of the streamed column: a step holds up to 64 rows of COMPLETE
  // groups (HOT / MIXED), or a piece of ONE longer group (GROUP).  pos = first position of the step, nact = its
  // rows, F = group-start flags of its rows, Fn = "the row after the step starts a group".
  const uint32_t partner_addr = half_partner_addr(lane);
  uint32_t dis_half = 0, dis_half_neg = 0;  // half-wave steps: lane (h, l) counts dis_half - dis_half_neg for pair h
  // fast tie steps: per lane, for the lane's pair (half-wave kernels: lanes >= 32 belong to the second pair)
  // (their counts and subtrahends go into dis_half / dis_half_neg: the same per-lane convention, two registers fewer)
  uint32_t seg_tie = 0, seg_tie2 = 0;
  bool grp_open = false;   // half-wave kernels: likewise; grp_start = first position of the open group
  int grp_start = 0, grp_entries = 0;   // grp_entries: steps of the open group so far
  uint32_t sv_k0 = 0, sv_k1 = 0, sv_h = 0;   // the open group's first step: its rows' values, still in registers when the second step closes it
  uint32_t sv_kw[NP], sv_hw[NP], sv_gw[NP], sneg[NP];     // whole-wave kernels: the same per pair (lane = row; sv_gw: the rows' counters); sneg: what is subtracted from a pair's dis
#pragma unroll
  for (int k = 0; k < NP; ++k) { sv_kw[k] = 0u; sv_hw[k] = 0u; sv_gw[k] = 0u; sneg[k] = 0u; }
  int pos = 0;
  if constexpr (half_mode) {
    if (seg_begin > 0) {
      // a later segment of a cut task: the rows in front of it enter `seen` uncounted -- a gather and two insertions per lane
      // and 64 rows -- and the prefix is built once
      constexpr int HP = (HI > 0 ? HI : 1);
      const bool hiP = lane >= 32u;
      const uint32_t l32p = lane & 31u;
      unsigned long long* seenP = hiP ? S[NP - 1].L.seen : S[0].L.seen;
      uint16_t* spreP = hiP ? S[NP - 1].L.spre : S[0].L.spre;
      const uint32_t GUARDP = (uint32_t)W << 6;
      // (256 rows per turn: four gathers in flight, the next turn's rows loaded behind them -- the wave is alone on its SIMD,
      //  that is why it was cut, and one gather per turn made the insertions as slow as a counted walk)
      const uint32_t ord_last = (uint32_t)pv.n_ord - 1u;   // (the run-ahead stays inside the column's zero padding)
      uint32_t rw[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) rw[u] = gload_u16(ord, min((uint32_t)u * 64u + lane, ord_last));
      for (int p = 0; p < seg_begin; p += 256) {
        uint2 rv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) rv[u] = gload_rec2(rec_blk, rw[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) rw[u] = gload_u16(ord, min((uint32_t)p + 256u + (uint32_t)u * 64u + lane, ord_last));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int pu = p + 64 * u;
          const auto sw = __builtin_amdgcn_permlane32_swap(comp[0] ? rv[u].y : rv[u].x, comp[NP - 1] ? rv[u].y : rv[u].x, false, false);
          const bool w0 = pu + (int)l32p < seg_begin, w1 = pu + 32 + (int)l32p < seg_begin;
          seen_insert(seenP, w0 ? (sw[0] & 0xFFFFu) : GUARDP);
          seen_insert(seenP, w1 ? (sw[1] & 0xFFFFu) : GUARDP);
        }
      }
      wave_lds_fence();
      rebuild_prefix_half<HP>(seenP, spreP, l32p);
      wave_lds_fence();
      pos = seg_begin;
      r0 = gload_u16(ord, (uint32_t)pos + lane);
      r1 = gload_u16(ord, (uint32_t)pos + 64u + lane);
      r2 = gload_u16(ord, (uint32_t)pos + 128u + lane);
    }
  }
  ICIKT_ST_MARK(6, 0)
  unsigned long long fw0 = 0ull, fw1 = 0ull;  // flag words fw_word, fw_word + 1 of the window (gf[W] is a zero guard)
  int fw_word = -1;
  // half-wave kernels: the streamed column's tie program and group offsets (PrepView::tprog / gidx)
  const uint32_t* tprog_col = half_mode ? pv.tprog + (int64_t)scol * pv.tp_stride : nullptr;
  // ... and the steps' records: their rows in lane layout (read three steps ahead, like the rows of the singleton region:
  // from the first program step on the ring r0 r1 r2 holds records, not positions of `ord`) and the MIXED steps' masks
  const uint16_t* srow_col = half_mode ? pv.srow + (int64_t)scol * pv.sr_steps * 64 : nullptr;
  const uint2* smask_col = half_mode ? pv.smask + (int64_t)scol * pv.sr_steps * 32 : nullptr;
  uint32_t tp_a = 0u, tp_b = 0u;   // entries of this step and the next, fetched two steps ahead (vector loads: see below)
  int tp_step = 0;                 // index of the program step that starts now
  bool tp_ok = false;
  // the ring moves on by a 64-row step: next step's rows are in r1 already -> gather its rec values now
  // (the gather is issued BEFORE the row load: vector loads return in order, so waiting for the gather at the top of
  //  the next step -- vmcnt(1) -- leaves the three-steps-ahead row load in flight)
  auto advance64 = [&]() {
    r0 = r1; r1 = r2;
    if (NP == 2) rv_pre = gload_rec2(rec_blk, r0);
    else rk_pre[0] = gload_u32(rec_blk, 2u * r0 + comp[0]);
    r2 = gload_u16(ord, (uint32_t)pos + 128u + lane);
    rk_ok = true;
  };
  auto take_rk = [&](uint32_t (&rk)[NP]) {   // the prefetched values of the step that starts now
    if (NP == 2) {
#pragma unroll
      for (int k = 0; k < NP; ++k) rk[k] = comp[k] ? rv_pre.y : rv_pre.x;
    } else {
      rk[0] = rk_pre[0];
    }
  };
  // ---- hot step: 64 rows, each its own tie group of the streamed column, nothing open ---------------------------
  // one-pair kernels on columns of at most 32 768 rows count the in-step pairs of two hot steps together
  // (wave_allpairs_packed2): inside the hot loop the step itself skips them
  bool defer_allpairs = false;
  uint32_t mix_sg1 = 0u, mix_sg2 = 0u;   // MIXED: the lane's same-group flag masks (both sub-steps); 0 in a plain hot step
  int mix_cls = 3;                       // MIXED: class of the step's largest group (half_step_flags_near), wave-uniform
  uint32_t solo_g0 = 0u, solo_g1 = 0u;   // SOLO: the tie-group indices (girow) of the lane's two rows
  // mode_tag: 0 = a hot step; 1 = MIXED; 2 = SOLO: each sub-step is ONE tie group of the streamed column -- no pair of its
  // rows is discordant, so there is no in-step chain at all, and its joint ties are counted like a GROUP step's in count mode:
  // a row adds one to the counter of its tie group in the gathered column, reads it back (c: rows of the sub-step in that
  // group; c - 1 partners, every pair twice) and clears it
  auto hot_step = [&](const uint32_t (&rk)[NP], auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    constexpr bool MIXED = MODE == 1;
    if constexpr (half_mode) {
      // lanes 0..31 run pair 0, lanes 32..63 pair 1, two 32-row sub-steps.  One 8-byte gather per row;
      // permlane32_swap turns the two 64-row registers (pair 0, pair 1) into the two sub-steps' operands
      // [pair 0 rows 0..31 | pair 1 rows 0..31], [.. rows 32..63].
      const auto sw = __builtin_amdgcn_permlane32_swap(rk[0], rk[NP - 1], false, false);
      const bool hi = lane >= 32u;
      unsigned long long* seenH = hi ? S[NP - 1].L.seen : S[0].L.seen;
      uint16_t* spreH = hi ? S[NP - 1].L.spre : S[0].L.spre;
      const uint32_t l32 = lane & 31u;
      constexpr int H = (HI > 0 ? HI : 1);
      // the in-step pairs of both sub-steps, while sub-step 0's LDS reads are in flight
      uint32_t inpairs = 0;
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        const uint32_t r = sub ? sw[1] : sw[0];
        const uint32_t qh = r & 0xFFFFu, loh = r >> 16;
        // All LDS traffic of the sub-step is ISSUED first, in the order in which it must take effect -- the query's
        // two reads, the rows' OR into seen, the reads of the prefix rebuild (one wave's DS operations execute in
        // order, so the query sees the bitset before, the rebuild after the insertion) -- and the all-pairs count,
        // which needs registers only, runs while they are in flight.
        // (addresses in bytes: slot w / H at 16 bytes, entry w % H at 2 -> 2 w + (16 - 2 H) (w / H); the word index of q
        //  is a bit field of the gathered value)
        const uint32_t wlo = r >> 22;
        const uint32_t dlo = (wlo * (65536u / (uint32_t)H + 1u)) >> 16;     // w / H for w < 32 H, H <= 13
        const uint32_t pre_lo = (uint32_t)*reinterpret_cast<const uint16_t*>(
            reinterpret_cast<const unsigned char*>(spreH) + 2u * wlo + (2u * (uint32_t)k1_half_slot<H>() - 2u * (uint32_t)H) * dlo);
        const unsigned long long word_lo = seenH[wlo];
        wave_lds_fence();
        atomicOr(reinterpret_cast<uint32_t*>(seenH) + __builtin_amdgcn_ubfe(r, 5, 10), 1u << (qh & 31u));
        wave_lds_fence();
        unsigned long long wv[H];
#pragma unroll
        for (int i = 0; i < H; ++i) wv[i] = seenH[l32 * (uint32_t)H + (uint32_t)i];
        // SOLO: the pair's counter table read as BYTES -- a sub-step holds at most 32 rows, and the table is all zeros
        // between its uses -- so that it serves columns of ICIKT_CNT_BYTES times the tie groups a GROUP step's counters do
        uint32_t solo_c = 1u;
        uint8_t* cntS = nullptr;
        if constexpr (MODE == 2) {
          const uint32_t g = sub ? solo_g1 : solo_g0;
          cntS = reinterpret_cast<uint8_t*>(spreH) + k1_half_pre_bytes<H>();
          if (g < solo_cap) atomicAdd(reinterpret_cast<uint32_t*>(cntS) + (g >> 2), 1u << ((g & 3u) << 3));
          wave_lds_fence();
          solo_c = (g < solo_cap) ? (uint32_t)cntS[g] : 1u;
        }
        if (sub == 0 && MODE != 2) {
          if constexpr (!MIXED) {
            // (issuing the chain's ds_bpermute in FRONT of the sub-step's reads, so that the chain waits for it alone --
            //  lgkmcnt(6) -- and runs under the reads' latency, was measured in round 4: no difference at six waves per SIMD)
            inpairs = half_step_count(sw[0], sw[1], lane, partner_addr);
          } else {
            const uint32_t Q = __builtin_amdgcn_perm(sw[1], sw[0], 0x05040100u);    // q of sub-step 0 | q of sub-step 1 << 16
            const uint32_t LO = __builtin_amdgcn_perm(sw[1], sw[0], 0x07060302u);   // lo likewise
            // (guard lanes: q = 64 W <= 0x47C0 and lo = 0 keep every field inside 15 bits)
            const uint32_t A = 0x7FFF7FFFu - Q, B = LO;
            uint32_t f1, f2, g1, g2;
            half_step_flags(A, B, (lane & 16u) ? B : A, partner_addr, f1, f2);      // [q_earlier < lo_later]
            // [q_later < lo_earlier]: looked at inside the groups only -- as far as the step's largest group reaches
            if (mix_cls == 0) half_step_flags_near<0>(B, A, (lane & 16u) ? A : B, partner_addr, g1, g2);
            else if (mix_cls == 1) half_step_flags_near<1>(B, A, (lane & 16u) ? A : B, partner_addr, g1, g2);
            else if (mix_cls == 2) half_step_flags_near<2>(B, A, (lane & 16u) ? A : B, partner_addr, g1, g2);
            else half_step_flags(B, A, (lane & 16u) ? A : B, partner_addr, g1, g2);
            inpairs = bcnt_acc(f2 & ~mix_sg2, bcnt_acc(f1 & ~mix_sg1, 0u));         // pairs of different groups only
            seg_tie = bcnt_acc(mix_sg2 & ~(f2 | g2), bcnt_acc(mix_sg1 & ~(f1 | g1), seg_tie));   // joint ties inside the groups
          }
        }
        // (the loaded words are consumed only behind the all-pairs block: an empty asm pins that order, or the
        //  scheduler hoists the popcounts -- and the wait for the loads -- in front of it)
        uint32_t pre_lo_k = pre_lo;
        unsigned long long word_lo_k = word_lo;
        asm volatile("" : "+v"(pre_lo_k), "+v"(word_lo_k));
#pragma unroll
        for (int i = 0; i < H; ++i) asm volatile("" : "+v"(wv[i]));
        // (the prefix entries are inclusive: what is subtracted is the part of the word at and above the position)
        dis_half += pre_lo_k + (sub == 0 ? inpairs : 0u);
        dis_half_neg = bcnt64_acc(word_lo_k >> (loh & 63u), dis_half_neg);
        // prefix of the half's bitset: lane l owns words [l*H, (l+1)*H); one aligned 16-byte store per lane
        uint32_t cw[H], run = 0;
#pragma unroll
        for (int i = 0; i < H; ++i) {
          run = bcnt64_acc(wv[i], run);
          cw[i] = run;
        }
        half_pre_store<H>(spreH, l32, half_incl_scan(run), cw);
        if constexpr (MODE == 2) {
          const uint32_t g = sub ? solo_g1 : solo_g0;
          seg_tie2 += solo_c - 1u;
          wave_lds_fence();
          if (g < solo_cap) cntS[g] = 0;
        }
        wave_lds_fence();
      }
    } else {
      // one pair on the whole wave: gather, count, insert into `seen`, update its counts.  <2, true, 0>: TWO pairs, one
      // after the other, each on the whole wave -- they share the streamed column and the rec block, so ONE 8-byte
      // gather per row serves both (long columns are bound by the L2's request rate: 64 requests per wave and gather,
      // the 400 KB block of a 50 000-row column never sits in a CU's L1), and their LDS phases are issued together:
      // the queries of both, then the insertions of both, then the count updates of both
      TwoLevel T[NP];
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        T[k] = tl_view(S[k].L.seen, S[k].L.spre);
        S[k].dis += tl_query(T[k], rk[k] >> 16, IT, magic);
      }
      if (!defer_allpairs) {
#pragma unroll
        for (int k = 0; k < NP; ++k) S[k].dis += wave_allpairs(rk[k] & 0xFFFFu, rk[k] >> 16, lane);
      }
      wave_lds_fence();
      if constexpr (NP == 1) {
        seen_insert(S[0].L.seen, rk[0] & 0xFFFFu);
        tl_update(T[0], true, rk[0] & 0xFFFFu, IT, magic, lane);
      } else {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          const uint32_t qk = rk[k] & 0xFFFFu;
          seen_insert(S[k].L.seen, qk);
          tl_update_rows(T[k], qk);
        }
        wave_lds_fence();
        uint32_t hh[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) hh[k] = atomicExch(&T[k].hist[lane], 0u);
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          const uint32_t below = wave_incl_scan(hh[k]) - hh[k];
          if (below != 0u) atomicAdd(&T[k].lb[lane], below);
        }
      }
      wave_lds_fence();
    }
  };
  while (pos < end_main) {
    if (pos + 64 <= hot_until) {
      // The singleton region (on continuous data: everything but the rows missing in the streamed column): a loop
      // of its own, so that the tie steps' state does not live in (or get merged into) the hot loop's registers.
      // ... and on longer columns too, with HALVED positions: the chain compares q >> 1 with (lo + 1) >> 1, 15 bits each,
      // and  q >> 1 < (lo + 1) >> 1  <=>  q < lo + (lo & 1):  exactly q < lo unless q == lo with lo odd -- another row at
      // the first position of the row's own tie group, i.e. a tie group of >= 2 rows that starts at an odd position.
      // Columns without one (K0 notes it: COL_ODD_TIE; continuous data never have one, the fill group starts at 0)
      // take the packed chain at any length; the others keep the two-instruction compare above 32 768 rows.
      const bool halved = (HI == 0) && (n > 32768) && !g_oddtie;
      const bool pack2 = (HI == 0) && ((n <= 32768) || halved);
      uint32_t rk_prev[NP];
#pragma unroll
      for (int k = 0; k < NP; ++k) rk_prev[k] = 0u;
      if (NP == 2 && HI == 0 && pack2 && (opts & 4)) {
        // Two long-column pairs in the HALF LAYOUT (round 3): lanes 0..31 take pair 0, lanes 32..63 pair 1, a 64-row step
        // is two 32-row sub-steps whose rows meet through `seen` -- the in-step pairs are then the half-wave chain
        // (half_step_count: 15 + 8 distances for both sub-steps and both pairs, ~57 instructions) instead of one packed
        // whole-wave chain per pair (~105 for two steps of ONE pair), and every LDS instruction of a sub-step serves both
        // pairs.  The count structure stays the two-level one; only its exchange changes: a half has 32 lanes for the 64
        // owner bins, so a lane reads-and-clears TWO bins (one 64-bit exchange) and adds their exclusive sums to two lb
        // entries (one 64-bit add).  Positions above 2^15 are compared halved, as in the packed whole-wave chain.
        const bool hiH = lane >= 32u;
        const uint32_t l32 = lane & 31u;
        const TwoLevel TH = tl_view(hiH ? S[NP - 1].L.seen : S[0].L.seen, hiH ? S[NP - 1].L.spre : S[0].L.spre);
        unsigned long long* const hist2 = reinterpret_cast<unsigned long long*>(TH.hist) + l32;
        unsigned long long* const lb2 = reinterpret_cast<unsigned long long*>(TH.lb) + l32;
        uint32_t dh = 0u;
        do {
          uint32_t rk[NP];
          if (rk_ok) {
            take_rk(rk);
          } else {
            const uint2 rv = gload_rec2(rec_blk, r0);
#pragma unroll
            for (int k = 0; k < NP; ++k) rk[k] = comp[k] ? rv.y : rv.x;
          }
          pos += 64;
          advance64();
          const auto sw = __builtin_amdgcn_permlane32_swap(rk[0], rk[NP - 1], false, false);
          HalfChain hc;
          {
            uint32_t ka = sw[0], kb = sw[1];
            if (halved) {   // q | lo << 16  ->  q >> 1 | (lo + 1) >> 1 << 16
              ka = (((ka + 0x10000u) >> 1) & 0x7FFF0000u) | ((ka >> 1) & 0x7FFFu);
              kb = (((kb + 0x10000u) >> 1) & 0x7FFF0000u) | ((kb >> 1) & 0x7FFFu);
            }
            hc = half_step_prep(ka, kb, lane, partner_addr);   // (its ds_bpermute queues with the first query's reads)
          }
#pragma unroll
          for (int sub = 0; sub < 2; ++sub) {
            const uint32_t r = sub ? sw[1] : sw[0];
            const uint32_t cnt = tl_query(TH, r >> 16, IT, magic);
            wave_lds_fence();
            seen_insert(TH.seen, r & 0xFFFFu);
            tl_update_rows(TH, r & 0xFFFFu);
            wave_lds_fence();
            const unsigned long long hh = atomicExch(hist2, 0ull);
            // the chain of the whole step, registers only, in two parts: each between the issue of a sub-step's atomics and
            // the wait for its exchange
            if (sub == 0) dh = half_step_part_a(hc, dh);
            else dh = half_step_part_b(hc, dh);
            dh += cnt;
            const uint32_t h0 = (uint32_t)hh, h1 = (uint32_t)(hh >> 32), hs = h0 + h1;
            const uint32_t below0 = half_incl_scan(hs) - hs, below1 = below0 + h0;
            if ((below0 | below1) != 0u) atomicAdd(lb2, (unsigned long long)below0 | ((unsigned long long)below1 << 32));
            wave_lds_fence();
          }
          ICIKT_ST_MARK(0, 64)
        } while (pos + 64 <= hot_until);
        S[0].dis += hiH ? 0u : dh;
        S[NP - 1].dis += hiH ? dh : 0u;
      } else
      if (HI == 0 && pack2) {
        // One pair per wave (or two, one after the other) with the packed two-step chain.  A step's LDS work is two
        // dependent round trips -- the query's reads, then the histogram exchange behind the insertions' atomics -- and at
        // 2 .. 4 waves per SIMD nothing else covers them, so the chain (registers only, ~105 instructions for two steps of
        // a pair) is placed BEHIND the issue of the atomics and the exchange and IN FRONT of the wait for them; its three
        // ds_bpermute are issued with the query's reads.  Two pairs take turns: pair 0 counts its steps (2m, 2m + 1)
        // during step 2m + 1, pair 1 counts (2m + 1, 2m + 2) during step 2m + 2 -- its step 0 alone, unpacked -- so that
        // every step has one chain to run.
        int ph[NP];   // per pair, wave-uniform: 0 = this step is counted alone, 1 = kept for the next step, 2 = counted with the kept one
#pragma unroll
        for (int k = 0; k < NP; ++k) ph[k] = (k == 0) ? 1 : 0;
        do {
          uint32_t rk[NP];
          if (rk_ok) {
            take_rk(rk);
          } else if (NP == 2) {
            const uint2 rv = gload_rec2(rec_blk, r0);
#pragma unroll
            for (int k = 0; k < NP; ++k) rk[k] = comp[k] ? rv.y : rv.x;
          } else {
            rk[0] = gload_u32(rec_blk, 2u * r0 + comp[0]);
          }
          pos += 64;
          advance64();
          TwoLevel T[NP];
          uint32_t cnt[NP];
#pragma unroll
          for (int k = 0; k < NP; ++k) {
            T[k] = tl_view(S[k].L.seen, S[k].L.spre);
            cnt[k] = tl_query(T[k], rk[k] >> 16, IT, magic);
          }
          Packed2 pk[NP];
#pragma unroll
          for (int k = 0; k < NP; ++k) {
            if (ph[k] == 2) {
              uint32_t ka = rk_prev[k], kb = rk[k];
              if (halved) {   // q | lo << 16  ->  q >> 1 | (lo + 1) >> 1 << 16
                ka = (((ka + 0x10000u) >> 1) & 0x7FFF0000u) | ((ka >> 1) & 0x7FFFu);
                kb = (((kb + 0x10000u) >> 1) & 0x7FFF0000u) | ((kb >> 1) & 0x7FFFu);
              }
              pk[k] = wave_allpairs_packed2_prep(ka, kb, lane);
            }
          }
#pragma unroll
          for (int k = 0; k < NP; ++k) {
            S[k].dis += cnt[k];
            if (ph[k] == 0) S[k].dis += wave_allpairs(rk[k] & 0xFFFFu, rk[k] >> 16, lane);
          }
          wave_lds_fence();
#pragma unroll
          for (int k = 0; k < NP; ++k) {
            seen_insert(S[k].L.seen, rk[k] & 0xFFFFu);
            tl_update_rows(T[k], rk[k] & 0xFFFFu);
          }
          wave_lds_fence();
          uint32_t hh[NP];
#pragma unroll
          for (int k = 0; k < NP; ++k) hh[k] = atomicExch(&T[k].hist[lane], 0u);
#pragma unroll
          for (int k = 0; k < NP; ++k)
            if (ph[k] == 2) S[k].dis += wave_allpairs_packed2_count(pk[k]);
#pragma unroll
          for (int k = 0; k < NP; ++k) {
            const uint32_t below = wave_incl_scan(hh[k]) - hh[k];
            if (below != 0u) atomicAdd(&T[k].lb[lane], below);
          }
          wave_lds_fence();
#pragma unroll
          for (int k = 0; k < NP; ++k) {
            if (ph[k] == 1) { rk_prev[k] = rk[k]; ph[k] = 2; }
            else ph[k] = 1;
          }
          ICIKT_ST_MARK(0, 64)
        } while (pos + 64 <= hot_until);
#pragma unroll
        for (int k = 0; k < NP; ++k)   // a kept step that met no partner
          if (ph[k] == 2) S[k].dis += wave_allpairs(rk_prev[k] & 0xFFFFu, rk_prev[k] >> 16, lane);
      } else {
      do {
        uint32_t rk[NP];
        if (rk_ok) {
          take_rk(rk);
        } else if (NP == 2) {
          const uint2 rv = gload_rec2(rec_blk, r0);
#pragma unroll
          for (int k = 0; k < NP; ++k) rk[k] = comp[k] ? rv.y : rv.x;
        } else {
          rk[0] = gload_u32(rec_blk, 2u * r0 + comp[0]);
        }
        pos += 64;
        advance64();
        hot_step(rk, std::integral_constant<int, 0>{});
        ICIKT_ST_MARK(0, 64)
      } while (pos + 64 <= hot_until);
      }
      defer_allpairs = false;
      hi_ok = false;
      continue;
    }
    if constexpr (half_mode) {
      // The step comes from the streamed column's TIE PROGRAM (k0_tie_program: cut and classified once per column by
      // the pre-pass, the list starts where the singleton loop above ends): one dword per step, fetched TWO steps
      // ahead by vector loads (a scalar load would make every wait for LDS data in the step a full drain), and the
      // step's RECORD: its rows in the lane layout it runs in (empty lanes: the guard row), three steps ahead.
      if (!tp_ok) {   // the first program step: the ring turns from positions of `ord` to records (one exposed gather per task)
        const int tp_base = (seg_step > 0) ? seg_step : 0;   // (a later segment of a cut task may begin inside the program)
        uint32_t off = (uint32_t)tp_base;
        asm volatile("" : "+v"(off));
        tp_a = gload_u32(tprog_col, off);
        tp_b = gload_u32(tprog_col, off + 1u);
        r0 = gload_u16(srow_col, (uint32_t)tp_base * 64u + lane);
        r1 = gload_u16(srow_col, (uint32_t)tp_base * 64u + 64u + lane);
        r2 = gload_u16(srow_col, (uint32_t)tp_base * 64u + 128u + lane);
        rv_pre = gload_rec2(rec_blk, r0);
        tp_step = tp_base;
        tp_ok = true;
        rk_ok = true;
        hi_ok = false;
      }
    }
    const uint32_t row = r0;
    const uint32_t hi_now = hi_pre;
    const uint32_t gi_now = gi_pre;
    const bool hi_now_ok = hi_ok;
    uint32_t rk[NP];
    if (rk_ok) {
      take_rk(rk);
    } else if (NP == 2) {  // after a short step: gather now, its latency runs behind the window logic
      const uint2 rv = gload_rec2(rec_blk, row);
#pragma unroll
      for (int k = 0; k < NP; ++k) rk[k] = comp[k] ? rv.y : rv.x;
    } else {
      rk[0] = gload_u32(rec_blk, 2u * row + comp[0]);
    }
    int nact;
    bool Fn;
    unsigned long long F;
    int kind = 0;          // 0: hot step or the general step (decided below), 1: MIXED, 2: GROUP (fast tie steps)
    bool closes = true;    // GROUP: the step holds the group's last row
    uint32_t e_next = 0u;  // half-wave kernels: the next step's program entry
    uint2 tmx = make_uint2(0u, 0u);   // MIXED (half-wave kernels): the lane's same-group flag masks
    if constexpr (half_mode) {
      const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)tp_a);
      e_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)tp_b);
      tp_a = tp_b;
      {
        uint32_t off = (uint32_t)(tp_step + 2);
        asm volatile("" : "+v"(off));
        tp_b = gload_u32(tprog_col, off);
      }
      nact = (int)tprog_rows(e);
      kind = (int)tprog_kind(e);
      closes = tprog_closes(e);
      Fn = (kind == 2) ? closes : true;
      F = (kind == 0) ? ~0ull : 0ull;     // (only `all_fast` below looks at it)
      // SOLO (kind 3): without the in-step chains when both gathered columns have their counters, else as a MIXED step
      if (kind == 3) kind = solo_ok ? 3 : 1;
      if (kind == 1) {   // which flags of the in-step compare belong to pairs inside a tie group (consumed behind the chains)
        tmx = smask_col[(uint32_t)tp_step * 32u + (lane & 31u)];
        mix_cls = (int)((e >> 16) & 3u);   // class of the step's largest group (k0_step_masks), carried by the program entry
      }
    } else
    if (pos + 64 <= hot_until) {
      nact = 64; Fn = true; F = ~0ull;
    } else {
      const int wc = pos >> 6, fb = pos & 63;
      if (fw_word != wc) {  // first step after the singleton region (or the very first)
        fw0 = gf[min(wc, W)];
        fw1 = gf[min(wc + 1, W)];
        fw_word = wc;
      }
      const unsigned long long w0 = uniform_u64(fw0), w1 = uniform_u64(fw1);
      F = fb ? ((w0 >> fb) | (w1 << (64 - fb))) : w0;
      const bool fnbit = ((w1 >> fb) & 1ull) != 0ull;          // position pos + 64 starts a group
      const int remaining = end_main - pos;
      const bool reach_end = remaining <= 64;
      if (remaining < 64) F &= (1ull << remaining) - 1ull;
      {
        // ---- tie steps: HOT (64 singleton rows), MIXED (complete groups of <= KS rows) or GROUP -----------
        constexpr int SW = half_mode ? 32 : 64;                  // rows a segment takes per step
        const int avail = reach_end ? remaining : 64;            // rows the window shows
        const int lim = min(remaining, SW);
        const bool endbit = fnbit || remaining == 64;            // row 64 of the window starts a group / is the end
        if (avail == 64 && F == ~0ull && endbit) {
          nact = 64; Fn = true;                                  // a hot step
        } else {
          // starts, with every position from the end of the data on marked as a start (it ends the last group)
          const unsigned long long Fz = (avail < 64) ? (F | (~0ull << avail)) : F;
          const unsigned long long Fr = Fz & ~1ull;
          const int next = (Fr != 0ull) ? (int)__builtin_ctzll(Fr) : (endbit ? 64 : 65);  // end of the group at pos
          constexpr int KS = k1_ks(half_mode);
          if ((F & 1ull) == 0ull || next > KS) {
            kind = 2;
            // (a half-wave kernel takes up to 64 rows of the group too: two 32-row pieces in one step of the ring)
            const int glim = half_mode ? min(remaining, 64) : lim;
            nact = min(next, glim);
            closes = next <= glim;
          } else {
            kind = 1;
            const unsigned long long z = ~Fz;                    // rows that continue a group
            unsigned long long r = z;
            // bit i: rows i .. i + KS - 1 continue a group: > KS rows
            if constexpr (KS == 32) {
              r &= r >> 1; r &= r >> 2; r &= r >> 4; r &= r >> 8; r &= r >> 16;
            } else {
#pragma unroll
              for (int i = 1; i < KS; ++i) r &= z >> i;
            }
            int s0 = 64;                                         // start of the first group the step must not take
            if (r != 0ull) s0 = 63 - (int)__builtin_clzll(Fz & ((1ull << __builtin_ctzll(r)) - 1ull));
            if (avail == 64 && !endbit) s0 = min(s0, 63 - (int)__builtin_clzll(Fz));  // cut by the window
            const int upper = min(min(remaining, 64), s0);      // (a MIXED step of a half-wave kernel takes 64 rows too)
            nact = (upper >= 64) ? 64 : (63 - (int)__builtin_clzll(Fz & ((2ull << upper) - 1ull) & ~1ull));
            if (nact < 64) F &= (1ull << nact) - 1ull;
          }
          Fn = closes;
        }
      }
      if (((pos + nact) >> 6) != wc) {  // the window moves on by one flag word at most (nact <= 64)
        fw0 = fw1;
        fw1 = gf[min(((pos + nact) >> 6) + 1, W)];
        fw_word = (pos + nact) >> 6;
      }
    }
    const int pos_next = pos + nact;
    const int pos_step = pos;
    const int kpos = pos + (int)lane;
    const bool valid = (int)lane < nact;
    const bool all_fast = (nact == 64) && (F == ~0ull) && Fn;
    pos = pos_next;
    if constexpr (half_mode) {
      // the ring moves on by one record, whatever the step held; the next step's rec values are gathered now
      ++tp_step;
      r0 = r1; r1 = r2;
      rv_pre = gload_rec2(rec_blk, r0);
      r2 = gload_u16(srow_col, (uint32_t)(tp_step + 2) * 64u + lane);
      rk_ok = true;
    } else if (nact == 64) {  // the ring stays aligned
      advance64();
    } else {
      // a shorter step: the next step's rows are in the ring already, nact lanes further on -> rotate the ring
      // (three cross-lane moves) instead of reloading it, and gather the rec values of the next step right away:
      // tie steps then hide their memory latencies like the hot steps do
      const int idx = (int)(((lane + (uint32_t)nact) & 63u) << 2);
      const uint32_t a0 = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)r0);
      const uint32_t a1 = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)r1);
      const uint32_t a2 = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)r2);
      const bool wrap = lane + (uint32_t)nact >= 64u;
      r0 = wrap ? a1 : a0;
      r1 = wrap ? a2 : a1;
      if (NP == 2) rv_pre = gload_rec2(rec_blk, r0);
      else rk_pre[0] = gload_u32(rec_blk, 2u * r0 + comp[0]);
      // (long columns: from the 32-bit copy of `order` -- behind the two-byte load the compiler puts its zero extension into this
      //  very block, and every tie step then began by waiting for a memory round trip; the word load lands in r2 and nothing waits)
      if (ord_w) r2 = gload_u32(ord_w, (uint32_t)pos + 128u + lane);
      else r2 = gload_u16(ord, (uint32_t)pos + 128u + lane);
      rk_ok = true;
    }
    if (!half_mode) {
      // a GROUP step of a pair in row mode needs the last position of every row's tie group in the gathered column ->
      // gathered one step ahead, beside the rec values (both columns of the block in one entry)
      // (a pair in count mode: the rows' tie-group indices likewise)
      bool any_row = false, any_cnt = false;
#pragma unroll
      for (int k = 0; k < NP; ++k) { any_row = any_row || (ntgB[k] < 0); any_cnt = any_cnt || (ntgB[k] >= 0 && cntB[k]); }
      hi_ok = kind == 2 && (any_row || any_cnt);   // after a GROUP step the next one is most likely a GROUP step too
      if (hi_ok && any_row) hi_pre = gload_u32(hi_blk, r0);
      if (hi_ok && any_cnt) gi_pre = gload_u32(gi_blk, r0);
    }
    if constexpr (half_mode) {
      // the next step is a GROUP step and a pair counts its joint ties row by row: the ends of the rows' tie groups
      // in the gathered column, one step ahead (lane (h, l): rows l and l + 32 of the next step, column of pair h)
      // ... or per tie group of the gathered column (count mode): the rows' tie-group indices
      hi_ok = tprog_kind(e_next) == TPROG_KIND_GROUP;
      if (hi_ok) {   // (lane = row: both columns of the block in one entry)
        if ((ntgB[0] < 0) || (ntgB[NP - 1] < 0)) hi_pre = gload_u32(hi_blk, r0);
        if ((ntgB[0] >= 0 && cntB[0]) || (ntgB[NP - 1] >= 0 && cntB[NP - 1])) gi_pre = gload_u32(gi_blk, r0);
      } else if (solo_ok && tprog_kind(e_next) == TPROG_KIND_SOLO) {
        gi_pre = gload_u32(gi_blk, r0);   // (sub-step layout: lanes l and l + 32 hold the lane's two rows)
        hi_ok = true;
      }
    }

    if (all_fast) {
      hot_step(rk, std::integral_constant<int, 0>{});
      ICIKT_ST_MARK(1, 64)
      continue;
    }

    if constexpr (half_mode) {
      // ---- tie step of a half-wave kernel: both pairs advance together, each on its 32 lanes ---------------------
      uint32_t lane_t = lane;
      asm volatile("" : "+v"(lane_t));   // keeps the per-lane choices below inside the tie step (see layout_rows)
      if (kind == 1) {
        // MIXED: the hot step on the step's rows (the record's empty lanes gathered the guard row: a position in the guard
        // word, lo = 0), then the pairs inside the groups
        mix_sg1 = tmx.x;
        mix_sg2 = tmx.y;
        hot_step(rk, std::integral_constant<int, 1>{});
        ICIKT_ST_MARK(2, nact)
      } else if (kind == 3) {
        // SOLO: the tie-group indices of the lane's two rows, column of the lane's pair
        const uint32_t gv = hi_now_ok ? gi_now : gload_u32(gi_blk, row);
        const auto gsw = __builtin_amdgcn_permlane32_swap(gv, gv, false, false);
        const bool compS = ((lane_t >= 32u) ? comp[NP - 1] : comp[0]) != 0u;
        solo_g0 = compS ? (gsw[0] >> 16) : (gsw[0] & 0xFFFFu);   // (GIROW_NONE: beyond any table)
        solo_g1 = compS ? (gsw[1] >> 16) : (gsw[1] & 0xFFFFu);
        hot_step(rk, std::integral_constant<int, 2>{});
        ICIKT_ST_MARK(1, nact)   // (diagnostic build: counted with the hot steps met in the main loop)
      } else {
        // GROUP: up to 64 rows of ONE tie group of the streamed column (a piece of it, or all of it).  Rows of one
        // group are never discordant with each other, so a group needs no all-pairs count at all -- and no second
        // bitset: `seen` does not change while the group's rows are QUERIED (phase A, step by step as the program
        // delivers them), and when the group closes its rows are INSERTED together (phase B: from registers when the
        // group is one or two steps, else by streaming its positions a second time) and the prefix is rebuilt once.
        // The group's joint ties, sum over the gathered column's tie groups g of C(rows of the group in g, 2):
        //   COUNT MODE (round 4, second half; the gathered column's tie groups have a counter each in the pair's table): a row adds one
        //     to the counter of its tie group (girow) in phase A; at the close every row reads its counter back, c, and
        //     counts c - 1 -- every pair twice: tie2 -- then clears it (one or two steps, rows still in registers), or the
        //     lanes read and clear the column's counters, C(c, 2) each (longer groups).  Two LDS operations per row
        //     instead of the eight prefix queries of a range count before and after;
        //   row mode (more tie groups than counters): per ROW of the group the rows in its cell [lo, hi], after - before
        //     - 1; "before" is taken in phase A, "after" in a third pass over the group's rows (phase C).  A range count
        //     is two prefix queries, whatever the width of the range.
        constexpr int H = (HI > 0 ? HI : 1);
        const bool hi_half = lane_t >= 32u;
        const uint32_t l32 = lane_t & 31u;
        unsigned long long* seenH = hi_half ? S[NP - 1].L.seen : S[0].L.seen;
        uint16_t* spreH = hi_half ? S[NP - 1].L.spre : S[0].L.spre;
        cnt_t* cntH = reinterpret_cast<cnt_t*>(reinterpret_cast<unsigned char*>(spreH) + k1_half_pre_bytes<H>());
        const uint32_t* tgH = hi_half ? tgB[NP - 1] : tgB[0];
        const int ntgH = hi_half ? ntgB[NP - 1] : ntgB[0];
        const bool compH = (hi_half ? comp[NP - 1] : comp[0]) != 0u;   // the lane's pair reads column 2a + 1 of the block
        const bool rowmode = ntgH < 0;                              // per lane = per pair
        const bool cntmode = !rowmode && (hi_half ? cntB[NP - 1] : cntB[0]);
        const bool listmode = !rowmode && !cntmode;
        const bool any_row = (ntgB[0] < 0) || (ntgB[NP - 1] < 0);   // wave-uniform
        const bool any_cnt = (ntgB[0] >= 0 && cntB[0]) || (ntgB[NP - 1] >= 0 && cntB[NP - 1]);
        // listed groups per lane (<= LISTK)
        const int kmax = (max((ntgB[0] >= 0 && !cntB[0]) ? ntgB[0] : 0, (ntgB[NP - 1] >= 0 && !cntB[NP - 1]) ? ntgB[NP - 1] : 0) + 31) >> 5;
        const uint32_t GUARD = (uint32_t)W << 6;                    // q = 64 W, lo = 0: queries 0, inserts into the guard word
        auto Q = [&](uint32_t p) -> uint32_t { return prefix_query_half<H>(seenH, spreH, p); };
        auto range = [&](uint32_t r) -> uint32_t { return Q((r >> 16) + 1u) - Q(r & 0xFFFFu); };   // r = lo | hi << 16
        // rows of the lane's pair in the cell [lo, hi] of a row (k = q | lo << 16), the row itself excluded
        auto cell_others = [&](uint32_t k, uint32_t h) -> uint32_t { return Q(h + 1u) - Q(k >> 16) - 1u; };
        // lane (h, l): rows l and l + 32 of a 64-row chunk, for pair h (a, b: the values of pair 0 / pair 1 by row)
        auto two_rows = [&](uint32_t a, uint32_t b, int cnt, uint32_t& x0, uint32_t& x1, bool& v0, bool& v1) {
          const auto sw = __builtin_amdgcn_permlane32_swap(a, b, false, false);
          v0 = (int)l32 < cnt; v1 = (int)l32 + 32 < cnt;
          x0 = sw[0]; x1 = sw[1];
        };
        auto two_his = [&](uint32_t hv, uint32_t& x0, uint32_t& x1) {   // hv: the block's hi (or tie-group index) entry by row
          const auto sw = __builtin_amdgcn_permlane32_swap(hv, hv, false, false);
          x0 = compH ? (sw[0] >> 16) : (sw[0] & 0xFFFFu);
          x1 = compH ? (sw[1] >> 16) : (sw[1] & 0xFFFFu);
        };
        // count mode: the counters are u16, two to a dword (LDS atomics are 32 bits wide; a count stays below 2^16)
        // (rows that are their own tie group and the guard rows of empty lanes stay out: they would all meet at one counter, and
        //  same-address atomics are served one lane at a time)
        auto cnt_add = [&](uint32_t g) {
          if (g < (uint32_t)tg_max) cnt_inc(cntH, g);
        };
        auto cnt_take = [&](uint32_t g) -> uint32_t {   // rows of the group in tie group g but one; the counter is cleared
          const uint32_t c = (uint32_t)cntH[g];
          return (g < (uint32_t)tg_max) ? c - 1u : 0u;
        };
        if (!grp_open) { grp_start = pos_step; grp_entries = 0; }
        uint32_t k0, k1;
        bool v0, v1;
        two_rows(rk[0], rk[NP - 1], nact, k0, k1, v0, v1);   // (the record's empty lanes gathered the guard row: k = GUARD)
        // h0, h1: row mode: the last position of the row's tie group in the gathered column; count mode: the group's index
        // (tg_max: the counter of the rows that are their own group -- and of the guard row)
        // (the cross-half moves run on the whole wave: any_row / any_cnt are wave-uniform, the choice per lane follows)
        uint32_t h0 = 0u, h1 = 0u;
        if (any_row) {
          uint32_t a0, a1;
          two_his(hi_now_ok ? hi_now : gload_u32(hi_blk, row), a0, a1);
          if (rowmode) { h0 = a0; h1 = a1; }
        }
        if (any_cnt) {
          uint32_t a0, a1;
          two_his(hi_now_ok ? gi_now : gload_u32(gi_blk, row), a0, a1);
          if (cntmode) { h0 = min(a0, (uint32_t)tg_max); h1 = min(a1, (uint32_t)tg_max); }
        }
        // phase A: rows of strictly higher groups below each row's tie group
        const uint32_t c0 = Q(k0 >> 16), c1 = Q(k1 >> 16);
        dis_half += c0 + c1;
        if (any_row && rowmode) {
          const uint32_t b0 = v0 ? Q(h0 + 1u) - c0 : 0u;
          const uint32_t b1 = v1 ? Q(h1 + 1u) - c1 : 0u;
          seg_tie2 -= b0 + b1;
        }
        if (any_cnt && cntmode) { cnt_add(h0); cnt_add(h1); }
        ICIKT_ST_MARK(4, 0)   // (diagnostic build: a GROUP step up to the end of phase A)
        if (closes) {
          // the group's rows: this step's alone (kept = 0), this and the previous step's, which are still in
          // registers (kept = 1), or more: those are streamed again from the group's first position
          const int kept = grp_entries;
          const int grp_end = pos;                                  // (pos has moved on to the next step)
          // list mode: rows of earlier groups inside each listed tie group of the gathered column
          uint32_t bef[LISTK];
#pragma unroll
          for (int i = 0; i < LISTK; ++i) {
            bef[i] = 0u;
            if (i < kmax) {
              const int g = (int)l32 + 32 * i;
              if (listmode && g < ntgH) bef[i] = range(tgH[g]);
            }
          }
          wave_lds_fence();
          // phase B: the group's rows enter `seen`
          if (kept <= 1) {
            if (kept == 1) {
              seen_insert(seenH, sv_k0 & 0xFFFFu);
              seen_insert(seenH, sv_k1 & 0xFFFFu);
            }
            seen_insert(seenH, k0 & 0xFFFFu);
            seen_insert(seenH, k1 & 0xFFFFu);
          } else {
            uint32_t rw = gload_u16(ord, (uint32_t)grp_start + lane);
            uint2 rv = gload_rec2(rec_blk, rw);
            uint32_t rw_n = gload_u16(ord, (uint32_t)grp_start + 64u + lane);
            for (int p = grp_start; p < grp_end; p += 64) {
              const uint2 rv_n = gload_rec2(rec_blk, rw_n);
              rw_n = gload_u16(ord, (uint32_t)p + 128u + lane);
              uint32_t a0, a1;
              bool w0, w1;
              two_rows(comp[0] ? rv.y : rv.x, comp[NP - 1] ? rv.y : rv.x, grp_end - p, a0, a1, w0, w1);
              seen_insert(seenH, w0 ? (a0 & 0xFFFFu) : GUARD);
              seen_insert(seenH, w1 ? (a1 & 0xFFFFu) : GUARD);
              rv = rv_n;
            }
          }
          wave_lds_fence();
          rebuild_prefix_half<H>(seenH, spreH, l32);
          wave_lds_fence();
          // list mode: C(rows of the group in the listed tie group, 2)
#pragma unroll
          for (int i = 0; i < LISTK; ++i) {
            if (i < kmax) {
              const int g = (int)l32 + 32 * i;
              if (listmode && g < ntgH) {
                const uint32_t c = range(tgH[g]) - bef[i];
                seg_tie += c * (c - 1u) / 2u;
              }
            }
          }
          // count mode: the counters hold, per tie group of the gathered column, the rows of this group
          if (any_cnt && cntmode) {
            if (kept <= 1) {
              uint32_t e = cnt_take(h0) + cnt_take(h1);
              if (kept == 1) e += cnt_take(sv_h & 0xFFFFu) + cnt_take(sv_h >> 16);
              seg_tie2 += e;
              wave_lds_fence();
              cntH[h0] = 0; cntH[h1] = 0;
              if (kept == 1) { cntH[sv_h & 0xFFFFu] = 0; cntH[sv_h >> 16] = 0; }
            } else {
              for (int g = (int)l32; g < ntgH; g += 32) {
                const uint32_t c = (uint32_t)cntH[g];
                seg_tie += c * (c - 1u) / 2u;
                cntH[g] = 0;
              }
            }
          }
          wave_lds_fence();
          ICIKT_ST_MARK(7, 0)   // (diagnostic build: a closing GROUP step's insertions, rebuild and counters)
          // phase C (row mode): rows of the group in each row's cell, the row itself excluded
          if (any_row) {
            if (kept <= 1) {
              uint32_t e = ((rowmode && v0) ? cell_others(k0, h0) : 0u) + ((rowmode && v1) ? cell_others(k1, h1) : 0u);
              if (kept == 1 && rowmode) e += cell_others(sv_k0, sv_h & 0xFFFFu) + cell_others(sv_k1, sv_h >> 16);
              seg_tie2 += e;
            } else {
              // (the gathers of the next chunk are in flight while this one is counted)
              uint32_t rw = gload_u16(ord, (uint32_t)grp_start + lane);
              uint2 rv = gload_rec2(rec_blk, rw);
              uint32_t hv = gload_u32(hi_blk, rw);
              uint32_t rw_n = gload_u16(ord, (uint32_t)grp_start + 64u + lane);
              for (int p = grp_start; p < grp_end; p += 64) {
                const uint2 rv_n = gload_rec2(rec_blk, rw_n);
                const uint32_t hv_n = gload_u32(hi_blk, rw_n);
                rw_n = gload_u16(ord, (uint32_t)p + 128u + lane);
                uint32_t a0, a1, g0, g1;
                bool w0, w1;
                two_rows(comp[0] ? rv.y : rv.x, comp[NP - 1] ? rv.y : rv.x, grp_end - p, a0, a1, w0, w1);
                two_his(hv, g0, g1);
                seg_tie2 += ((rowmode && w0) ? cell_others(a0, g0) : 0u) + ((rowmode && w1) ? cell_others(a1, g1) : 0u);
                rv = rv_n; hv = hv_n;
              }
            }
          }
          grp_open = false;
        } else {
          if (grp_entries == 0) { sv_k0 = k0; sv_k1 = k1; sv_h = h0 | (h1 << 16); }   // (a step that does not close has 64 rows)
          ++grp_entries;
          grp_open = true;
        }
        ICIKT_ST_MARK(3, nact)
      }
    } else {
      // ---- tie step of a whole-wave kernel: one pair, or two one after the other (they share the streamed column, hence
      //      the step; lane = row) -----------------------------------------------------------------------------------
      if (kind == 1) {
        // MIXED: complete groups of at most k1_ks(false) rows; seen and its counts only
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          SegState st;
          st.seen = (lds_u64p)S[k].L.seen;
          st.spre = (lds_u16p)S[k].L.spre;
          const SegCounts c = seg_mixed_step<64, HI>(st, F, nact, rk[k], IT, magic, lane);
          S[k].dis += c.dis; sneg[k] += c.neg; S[k].tie += c.tie;
        }
        ICIKT_ST_MARK(2, nact)
      } else {
        // GROUP: the design of the half-wave kernels (above) on 64 lanes and the two-level counts -- phase A queries while
        // `seen` stands still, the group's rows enter `seen` when it closes (one or two steps: from registers, each step's
        // rows with an incremental update of the counts; longer: streamed again, the counts rebuilt once), joint ties from
        // range counts before / after (list mode: <= 2 listed groups per lane; row mode: per row, phase C)
        // count mode (round 4; gathered columns of more than 128 tie groups, while a pair's counter table covers them): a
        // row adds one to the counter of its tie group (girow) in phase A, the close reads the counters back -- no third pass
        bool any_row = false, any_cnt = false;
#pragma unroll
        for (int k = 0; k < NP; ++k) { any_row = any_row || (ntgB[k] < 0); any_cnt = any_cnt || (ntgB[k] >= 0 && cntB[k]); }
        if (!grp_open) { grp_start = pos_step; grp_entries = 0; }
        TwoLevel T[NP];
        uint32_t hk[NP], gk[NP];
        cnt_t* cntT[NP];
        const uint32_t hv = any_row ? (hi_now_ok ? hi_now : gload_u32(hi_blk, row)) : 0u;
        const uint32_t gv = any_cnt ? (hi_now_ok ? gi_now : gload_u32(gi_blk, row)) : 0u;
        auto cnt_take = [&](cnt_t* t, uint32_t g) -> uint32_t {   // rows of the group in tie group g but one
          const uint32_t c = (uint32_t)t[min(g, (uint32_t)cnt_cap)];
          return (g < (uint32_t)cnt_cap) ? c - 1u : 0u;
        };
        auto rangeT = [&](const TwoLevel& Tk, uint32_t r) -> uint32_t {   // r = lo | hi << 16
          return tl_query(Tk, (r >> 16) + 1u, IT, magic) - tl_query(Tk, r & 0xFFFFu, IT, magic);
        };
        auto cell_others = [&](const TwoLevel& Tk, uint32_t kq, uint32_t h) -> uint32_t {
          return tl_query(Tk, h + 1u, IT, magic) - tl_query(Tk, kq >> 16, IT, magic) - 1u;
        };
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          T[k] = tl_view(S[k].L.seen, S[k].L.spre);
          cntT[k] = reinterpret_cast<cnt_t*>(reinterpret_cast<unsigned char*>(S[k].L.spre) + TL_BYTES);
          hk[k] = comp[k] ? (hv >> 16) : (hv & 0xFFFFu);
          // (GIROW_NONE, a row that is its own tie group, and the rows beyond the step: no counter)
          gk[k] = valid ? min(comp[k] ? (gv >> 16) : (gv & 0xFFFFu), (uint32_t)cnt_cap) : (uint32_t)cnt_cap;
          // phase A: rows of strictly higher groups below each row's tie group
          const uint32_t c = valid ? tl_query(T[k], rk[k] >> 16, IT, magic) : 0u;
          S[k].dis += c;
          if (ntgB[k] < 0) S[k].tie2 -= valid ? (tl_query(T[k], hk[k] + 1u, IT, magic) - c) : 0u;
          else if (cntB[k] && gk[k] < (uint32_t)cnt_cap) cnt_inc(cntT[k], gk[k]);
        }
        if (closes) {
          const int kept = grp_entries;
          const int grp_end = pos;                                  // (pos has moved on to the next step)
          uint32_t bef[NP][2];
#pragma unroll
          for (int k = 0; k < NP; ++k) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              bef[k][i] = 0u;
              const int g = (int)lane + 64 * i;
              if (!cntB[k] && g < ntgB[k]) bef[k][i] = rangeT(T[k], tgB[k][g]);
            }
          }
          wave_lds_fence();
          // phase B: the group's rows enter `seen`
          if (kept <= 1) {
            // (the rows of both steps and both pairs first -- bits and the counts inside their owners -- then ONE collection of
            //  the owners' histogram per pair: four incremental updates one after the other were a dozen LDS round trips per close)
#pragma unroll
            for (int k = 0; k < NP; ++k) {
              if (kept == 1) {
                seen_insert(S[k].L.seen, sv_kw[k] & 0xFFFFu);
                tl_update_rows(T[k], sv_kw[k]);
              }
              if (valid) {
                seen_insert(S[k].L.seen, rk[k] & 0xFFFFu);
                tl_update_rows(T[k], rk[k]);
              }
            }
            wave_lds_fence();
            uint32_t hcol[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) hcol[k] = atomicExch(&T[k].hist[lane], 0u);
#pragma unroll
            for (int k = 0; k < NP; ++k) {
              const uint32_t below = wave_incl_scan(hcol[k]) - hcol[k];
              if (below != 0u) atomicAdd(&T[k].lb[lane], below);
            }
            wave_lds_fence();
          } else {
            uint32_t rw = gload_u16(ord, (uint32_t)grp_start + lane);
            uint32_t rw_n = gload_u16(ord, (uint32_t)grp_start + 64u + lane);
            for (int p = grp_start; p < grp_end; p += 64) {
              uint32_t rkc[NP];
              if (NP == 2) {
                const uint2 rv = gload_rec2(rec_blk, rw);
#pragma unroll
                for (int k = 0; k < NP; ++k) rkc[k] = comp[k] ? rv.y : rv.x;
              } else {
                rkc[0] = gload_u32(rec_blk, 2u * rw + comp[0]);
              }
              rw = rw_n;
              rw_n = gload_u16(ord, (uint32_t)p + 128u + lane);
              if (p + (int)lane < grp_end) {
#pragma unroll
                for (int k = 0; k < NP; ++k) seen_insert(S[k].L.seen, rkc[k] & 0xFFFFu);
              }
            }
            wave_lds_fence();
#pragma unroll
            for (int k = 0; k < NP; ++k) tl_rebuild(T[k], Wp4, lane);
            wave_lds_fence();
          }
          // list mode: C(rows of the group in the listed tie group, 2)
#pragma unroll
          for (int k = 0; k < NP; ++k) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const int g = (int)lane + 64 * i;
              if (!cntB[k] && g < ntgB[k]) {
                const uint32_t c = rangeT(T[k], tgB[k][g]) - bef[k][i];
                S[k].tie += c * (c - 1u) / 2u;
              }
            }
          }
          // count mode: the counters hold, per tie group of the gathered column, the rows of this group: a group of one or two
          // steps reads them back by row (c - 1 each: every pair twice, tie2) and clears what it touched, a longer one by tie group
          if (any_cnt) {
#pragma unroll
            for (int k = 0; k < NP; ++k) {
              if (ntgB[k] >= 0 && cntB[k]) {
                if (kept <= 1) {
                  uint32_t e = cnt_take(cntT[k], gk[k]);
                  if (kept == 1) e += cnt_take(cntT[k], sv_gw[k]);
                  S[k].tie2 += e;
                } else {
                  for (int g = (int)lane; g < ntgB[k]; g += 64) {
                    const uint32_t c = (uint32_t)cntT[k][g];
                    S[k].tie += c * (c - 1u) / 2u;
                    cntT[k][g] = 0;
                  }
                }
              }
            }
            if (kept <= 1) {
              wave_lds_fence();
#pragma unroll
              for (int k = 0; k < NP; ++k) {
                if (ntgB[k] >= 0 && cntB[k]) {
                  cntT[k][gk[k]] = 0;                          // (entry cnt_cap: the scratch entry of the rows without a counter)
                  if (kept == 1) cntT[k][sv_gw[k]] = 0;
                }
              }
            }
            wave_lds_fence();
          }
          // phase C (row mode): rows of the group in each row's cell, the row itself excluded
          if (any_row) {
            if (kept <= 1) {
#pragma unroll
              for (int k = 0; k < NP; ++k) {
                if (ntgB[k] < 0) {
                  uint32_t e = valid ? cell_others(T[k], rk[k], hk[k]) : 0u;
                  if (kept == 1) e += cell_others(T[k], sv_kw[k], sv_hw[k]);
                  S[k].tie2 += e;
                }
              }
            } else {
              uint32_t rw = gload_u16(ord, (uint32_t)grp_start + lane);
              uint32_t rw_n = gload_u16(ord, (uint32_t)grp_start + 64u + lane);
              for (int p = grp_start; p < grp_end; p += 64) {
                uint32_t rkc[NP];
                if (NP == 2) {
                  const uint2 rv = gload_rec2(rec_blk, rw);
#pragma unroll
                  for (int k = 0; k < NP; ++k) rkc[k] = comp[k] ? rv.y : rv.x;
                } else {
                  rkc[0] = gload_u32(rec_blk, 2u * rw + comp[0]);
                }
                const uint32_t hvc = gload_u32(hi_blk, rw);
                rw = rw_n;
                rw_n = gload_u16(ord, (uint32_t)p + 128u + lane);
                const bool w = p + (int)lane < grp_end;
#pragma unroll
                for (int k = 0; k < NP; ++k)
                  if (ntgB[k] < 0) S[k].tie2 += w ? cell_others(T[k], rkc[k], comp[k] ? (hvc >> 16) : (hvc & 0xFFFFu)) : 0u;
              }
            }
          }
          grp_open = false;
        } else {
          if (grp_entries == 0) {   // (a step that does not close has 64 rows)
#pragma unroll
            for (int k = 0; k < NP; ++k) { sv_kw[k] = rk[k]; sv_hw[k] = hk[k]; sv_gw[k] = gk[k]; }
          }
          ++grp_entries;
          grp_open = true;
          // The group goes on.  Its further FULL steps only query (phase A): they run in a loop of their own, without
          // the window logic per step -- a group of thousands of rows (few distinct values; a fill group met in row
          // mode) is then a stream of gathers and queries.  The group's end: the first group start at or after pos,
          // 64 flag words per probe (lane l reads word pos / 64 + l).
          // (probed once the group has filled two steps: shorter groups close from registers and never get here)
          int gend = pos;
          if (grp_entries == 2) gend = end_main;
          for (int wb = pos >> 6; wb < W && grp_entries == 2; wb += 64) {
            const int w = wb + (int)lane;
            unsigned long long f = (w < W) ? gf[w] : 0ull;
            if (w == (pos >> 6)) f &= ~0ull << (pos & 63);
            const unsigned long long any = __ballot(f != 0ull);
            if (any != 0ull) {
              const int l0 = (int)__builtin_ctzll(any);
              const unsigned long long f0 =
                  (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)f, l0) |
                  ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(f >> 32), l0) << 32);
              gend = min(end_main, (wb + l0) * 64 + (int)__builtin_ctzll(f0));
              break;
            }
          }
          if (pos + 64 < gend) {
            uint32_t hvn = any_row ? (hi_ok ? hi_pre : gload_u32(hi_blk, r0)) : 0u;   // the next step's rows are in r0
            uint32_t gvn = any_cnt ? (hi_ok ? gi_pre : gload_u32(gi_blk, r0)) : 0u;
            do {
              uint32_t rkl[NP];
              take_rk(rkl);                           // (rk_ok: the ring moved on by a 64-row step)
              const uint32_t hvl = hvn, gvl = gvn;
              pos += 64;
              advance64();
              if (any_row) hvn = gload_u32(hi_blk, r0);
              if (any_cnt) gvn = gload_u32(gi_blk, r0);
#pragma unroll
              for (int k = 0; k < NP; ++k) {
                const uint32_t c = tl_query(T[k], rkl[k] >> 16, IT, magic);
                S[k].dis += c;
                if (ntgB[k] < 0) S[k].tie2 -= tl_query(T[k], (comp[k] ? (hvl >> 16) : (hvl & 0xFFFFu)) + 1u, IT, magic) - c;
                else if (cntB[k]) {
                  const uint32_t g = comp[k] ? (gvl >> 16) : (gvl & 0xFFFFu);
                  if (g < (uint32_t)cnt_cap) cnt_inc(cntT[k], g);
                }
              }
              ++grp_entries;
              ICIKT_ST_MARK(3, 64)
            } while (pos + 64 < gend);
            hi_pre = hvn;
            gi_pre = gvn;
            hi_ok = any_row || any_cnt;
            fw_word = -1;                             // the flag window is read again
          }
        }
        ICIKT_ST_MARK(3, nact)
      }
    }
  }

  // the rows of the last tie group of the streamed column: gather, add lo; then the group's joint ties T from range
  // counts of `seen` and the correction C(m, 2) - T of the summed lo's
  unsigned long long corr[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) corr[k] = 0ull;
  if (closed_form) {
    if constexpr (half_mode) {
      // Half-wave kernels: every row that is not in `seen` by now belongs to this last group, so the group has
      // (size of g) - (rows of g in seen) rows in a tie group g of the gathered column: T from one range count per
      // listed group, and the group's rows only gather and add their lo.
      uint32_t lane_t = lane;
      asm volatile("" : "+v"(lane_t));
      const bool hi_half = lane_t >= 32u;
      const uint32_t l32 = lane_t & 31u;
      constexpr int H = (HI > 0 ? HI : 1);
      unsigned long long* seenH = hi_half ? S[NP - 1].L.seen : S[0].L.seen;
      uint16_t* spreH = hi_half ? S[NP - 1].L.spre : S[0].L.spre;
      const uint32_t* tgH = hi_half ? tgB[NP - 1] : tgB[0];
      const int ntgH = hi_half ? ntgB[NP - 1] : ntgB[0];
      // (four listed groups per lane at a time: their list entries are loaded together -- a column of a thousand tie groups
      //  would otherwise pay a memory round trip per 32 of them, one after the other)
      uint32_t tl = 0;
      for (int g0 = (int)l32; g0 < ntgH; g0 += 128) {
        uint32_t rr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) rr[i] = (g0 + 32 * i < ntgH) ? tgH[g0 + 32 * i] : 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (g0 + 32 * i < ntgH) {
            const uint32_t r = rr[i];
            const uint32_t in_seen = prefix_query_half<H>(seenH, spreH, (r >> 16) + 1u) - prefix_query_half<H>(seenH, spreH, r & 0xFFFFu);
            const uint32_t c = (r >> 16) - (r & 0xFFFFu) + 1u - in_seen;
            tl += c * (c - 1u) / 2u;
          }
        }
      }
      seg_tie += tl;
      // (four 64-row chunks per turn: four gathers in flight and the next turn's rows behind them -- one chunk per turn, each gather
      //  waiting for its row, was a memory round trip per 64 rows: a fill group of 1 000 rows cost a task sixteen of them)
      {
        uint32_t rw[4];
        rw[0] = r0;
#pragma unroll
        for (int i = 1; i < 4; ++i) rw[i] = gload_u16(ord, min((uint32_t)(last_start + 64 * i) + lane, (uint32_t)n));   // (order[n ..]: zero padding)
        for (int p = last_start; p < n; p += 256) {
          uint2 rv[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) rv[i] = gload_rec2(rec_blk, rw[i]);
#pragma unroll
          for (int i = 0; i < 4; ++i) rw[i] = gload_u16(ord, min((uint32_t)(p + 256 + 64 * i) + lane, (uint32_t)n));
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (p + 64 * i + (int)lane < n) {
#pragma unroll
              for (int k = 0; k < NP; ++k) S[k].dis += (comp[k] ? rv[i].y : rv[i].x) >> 16;
            }
          }
        }
      }
      const unsigned long long m = (unsigned long long)(n - last_start);
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const bool mine = (lane >= 32u) == (k == NP - 1);
        corr[k] = m * (m - 1ull) / 2ull - wave_sum_u64(mine ? tl : 0u);
      }
    } else {
      // Whole-wave kernels: the same -- every row that is not in `seen` by now belongs to this last group, so it has
      // (size of g) - (rows of g in seen) rows in a listed tie group g of the gathered column; its rows only gather and
      // add their lo.
      uint32_t tl[NP];
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const TwoLevel Tk = tl_view(S[k].L.seen, S[k].L.spre);
        tl[k] = 0u;
        for (int g = (int)lane; g < ntgB[k]; g += 64) {
          const uint32_t r = tgB[k][g];
          const uint32_t in_seen = tl_query(Tk, (r >> 16) + 1u, IT, magic) - tl_query(Tk, r & 0xFFFFu, IT, magic);
          const uint32_t c = (r >> 16) - (r & 0xFFFFu) + 1u - in_seen;
          tl[k] += c * (c - 1u) / 2u;
        }
        S[k].tie += tl[k];
      }
      // (four 64-row chunks per turn, as in the half-wave kernels above)
      {
        uint32_t rw[4];
        rw[0] = r0;
#pragma unroll
        for (int i = 1; i < 4; ++i) rw[i] = gload_u16(ord, min((uint32_t)(last_start + 64 * i) + lane, (uint32_t)n));   // (order[n ..]: zero padding)
        for (int p = last_start; p < n; p += 256) {
          uint32_t rkt[4][NP];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (NP == 2) {
              const uint2 rv = gload_rec2(rec_blk, rw[i]);
#pragma unroll
              for (int k = 0; k < NP; ++k) rkt[i][k] = comp[k] ? rv.y : rv.x;
            } else {
              rkt[i][0] = gload_u32(rec_blk, 2u * rw[i] + comp[0]);
            }
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) rw[i] = gload_u16(ord, min((uint32_t)(p + 256 + 64 * i) + lane, (uint32_t)n));
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (p + 64 * i + (int)lane < n) {
#pragma unroll
              for (int k = 0; k < NP; ++k) S[k].dis += rkt[i][k] >> 16;
            }
          }
        }
      }
      const unsigned long long m = (unsigned long long)(n - last_start);
#pragma unroll
      for (int k = 0; k < NP; ++k) corr[k] = m * (m - 1ull) / 2ull - wave_sum_u64(tl[k]);
    }
    ICIKT_ST_MARK(5, n - last_start)
  }

#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const bool mine = (NP == 1) || ((lane >= 32u) == (k == NP - 1));   // per-lane accumulators of this pair
    const unsigned long long dis = wave_sum_u64(S[k].dis) + wave_sum_u64(mine ? dis_half : 0u) +
                                   0ull - wave_sum_u64(mine ? dis_half_neg : 0u) - wave_sum_u64(sneg[k]) - corr[k];
    const unsigned long long ntie = wave_sum_u64(S[k].tie) + wave_sum_u64(mine ? seg_tie : 0u) +
                                    ((wave_sum_u64(S[k].tie2) + wave_sum_u64(mine ? seg_tie2 : 0u)) >> 1);
    const unsigned long long cbs = wave_sum_u64(cb[k]);
    const unsigned long long ggs = wave_sum_u64(gg[k]);
    if (lane == 0 && k < np) {
      if (split > 1) {   // a segment of a cut task: the host has zeroed the pair's record
        atomicAdd(&raw[pidx[k]].dis, dis);
        atomicAdd(&raw[pidx[k]].ntie, ntie);
        if (seg == 0) { raw[pidx[k]].c_both = (uint32_t)cbs; raw[pidx[k]].g = (uint32_t)ggs; }
      } else {
        PairRaw o;
        o.dis = dis;
        o.ntie = ntie;
        o.c_both = (uint32_t)cbs;
        o.g = (uint32_t)ggs;
        raw[pidx[k]] = o;
      }
    }
  }
  wave_lds_fence();
  ICIKT_ST_MARK(6, 0)
  ICIKT_ST_FLUSH
    if (persist) {
      int t = 0;
      if (lane == 0u) t = atomicAdd(my_ctr, 1);
      task = t_lo + __builtin_amdgcn_readfirstlane(t);
    }
  } while (persist && task < t_hi);  // task loop
}


// ------------------------------------------------------------------------------------------------
// K2: epilogue, one pair per lane
// ------------------------------------------------------------------------------------------------
// R's pnorm (libR nmath pnorm_both): W. J. Cody, Math. Comp. 23 (1969) 631-637.
__device__ void pnorm_both_dev(double x, double& cum, double& ccum) {
  const double a[5] = {2.2352520354606839287, 161.02823106855587881, 1067.6894854603709582,
                       18154.981253343561249, 0.065682337918207449113};
  const double b[4] = {47.20258190468824187, 976.09855173777669322, 10260.932208618978205,
                       45507.789335026729956};
  const double c[9] = {0.39894151208813466764, 8.8831497943883759412, 93.506656132177855979,
                       597.27027639480026226,  2494.5375852903726711, 6848.1904505362823326,
                       11602.651437647350124,  9842.7148383839780218, 1.0765576773720192317e-8};
  const double d[8] = {22.266688044328115691, 235.38790178262499861, 1519.377599407554805,
                       6485.558298266760755,  18615.571640885098091, 34900.952721145977266,
                       38912.003286093271411, 19685.429676859990727};
  const double pp[6] = {0.21589853405795699,    0.1274011611602473639, 0.022235277870649807,
                        0.001421619193227893466, 2.9112874951168792e-5, 0.02307344176494017303};
  const double qq[5] = {1.28426009614491121,   0.468238212480865118, 0.0659881378689285515,
                        0.00378239633202758244, 7.29751555083966205e-5};
  if (x != x) { cum = x; ccum = x; return; }
  const double y = fabs(x);
  double xnum, xden, temp, xsq, del;
  if (y <= 0.67448975) {
    if (y > 1.1102230246251565e-16) {
      xsq = x * x;
      xnum = a[4] * xsq;
      xden = xsq;
      for (int i = 0; i < 3; ++i) { xnum = (xnum + a[i]) * xsq; xden = (xden + b[i]) * xsq; }
    } else {
      xnum = xden = 0.0;
    }
    temp = x * (xnum + a[3]) / (xden + b[3]);
    cum = 0.5 + temp;
    ccum = 0.5 - temp;
  } else if (y <= 5.656854249492380195206754896838) {
    xnum = c[8] * y;
    xden = y;
    for (int i = 0; i < 7; ++i) { xnum = (xnum + c[i]) * y; xden = (xden + d[i]) * y; }
    temp = (xnum + c[7]) / (xden + d[7]);
    xsq = trunc(y * 16) / 16;
    del = (y - xsq) * (y + xsq);
    cum = exp(-xsq * xsq * 0.5) * exp(-del * 0.5) * temp;
    ccum = 1.0 - cum;
    if (x > 0.) { temp = cum; cum = ccum; ccum = temp; }
  } else if ((-37.5193 < x && x < 8.2924) || (-8.2924 < x && x < 37.5193)) {
    xsq = 1.0 / (x * x);
    xnum = pp[5] * xsq;
    xden = xsq;
    for (int i = 0; i < 4; ++i) { xnum = (xnum + pp[i]) * xsq; xden = (xden + qq[i]) * xsq; }
    temp = xsq * (xnum + pp[4]) / (xden + qq[4]);
    temp = (0.398942280401432677939946059934 - temp) / y;
    xsq = trunc(x * 16) / 16;
    del = (x - xsq) * (x + xsq);
    cum = exp(-xsq * xsq * 0.5) * exp(-del * 0.5) * temp;
    ccum = 1.0 - cum;
    if (x > 0.) { temp = cum; cum = ccum; ccum = temp; }
  } else {
    if (x > 0) { cum = 1.; ccum = 0.; } else { cum = 0.; ccum = 1.; }
  }
}

__device__ __forceinline__ void pnorm_tails(double z, double& lower, double& upper) {
  if (isinf(z)) { lower = z > 0 ? 1.0 : 0.0; upper = 1.0 - lower; return; }
  pnorm_both_dev(z, lower, upper);
}

// element of count_rank_tie's three sums for one tie group of size t, int32 arithmetic as uint32
__device__ __forceinline__ void tie_terms32(int t, uint32_t& a0, uint32_t& a1, uint32_t& a2) {
  if (t < 2) { a0 = a1 = a2 = 0; return; }
  const uint32_t ut = (uint32_t)t, tt1 = ut * (ut - 1u);
  a0 = tt1; a1 = tt1 * (ut - 2u); a2 = tt1 * (2u * ut + 5u);
}
__device__ __forceinline__ void tie_terms64(long long t, long long& a0, long long& a1, long long& a2) {
  if (t < 2) { a0 = a1 = a2 = 0; return; }
  a0 = t * (t - 1); a1 = t * (t - 1) * (t - 2); a2 = t * (t - 1) * (2 * t + 5);
}

__global__ void __launch_bounds__(256)
k2_epilogue(PrepView pv, const int32_t* __restrict__ pi, const int32_t* __restrict__ pj,
            const PairRaw* __restrict__ raw, int64_t n_pairs, int perspective, int alternative,
            int continuity, int exact64, double* __restrict__ out4, int64_t* __restrict__ counts,
            int32_t* __restrict__ reasons) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  const double NA = __longlong_as_double(0x7FF00000000007A2ll);  // R's NA_real_
  const ColStats sx = *pv.col_stats(pi[p]);
  const ColStats sy = *pv.col_stats(pj[p]);
  const PairRaw rw = (pv.n > 0) ? raw[p] : PairRaw{0ull, 0ull, 0u, 0u};
  const long long n = pv.n;
  const long long cb = rw.c_both;
  const bool local = (perspective == ICIKT_PERSPECTIVE_LOCAL_);

  int reason = 0;
  double o_tau = NA, o_p = NA, o_tmax = NA, o_comp = NA;
  long long cnt[ICIKT_CNT_FIELDS_] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

  // kendallc.cpp:190-199; dropping the both-missing rows cannot change "all missing"
  if (sx.nna == n || sy.nna == n) {
    reason = 1;
  } else {
    const long long ne = local ? (n - cb) : n;                               // :180-185, :221
    const long long missing = (long long)sx.nna + sy.nna - (local ? 2 * cb : cb);  // :208-211
    const double completeness = 1.0 - (double)missing / (double)ne;
    // a fill group vanishes under "local" when every member was a both-missing row
    const int kx = sx.ngroups - ((local && sx.nna > 0 && sx.tfill == cb) ? 1 : 0);
    const int ky = sy.ngroups - ((local && sy.nna > 0 && sy.tfill == cb) ? 1 : 0);
    if (ne < 2) {
      reason = 2;                                                            // :224-231
    } else if (kx == 1 || ky == 1) {
      reason = 3;                                                            // :234-244
    } else {
      const long long shrink = local ? cb : 0;
      double xtie, x0, x1, ytie, y0, y1, ntie;
      const long long g = rw.g, g2 = g - shrink;
      const long long others = (long long)rw.ntie - g * (g - 1) / 2;          // joint ties outside the fill/fill cell
      if (exact64) {
        long long a0, a1, a2, b0, b1, b2;
        tie_terms64(sx.tfill, a0, a1, a2); tie_terms64(sx.tfill - shrink, b0, b1, b2);
        xtie = (double)((sx.e0 - a0 + b0) / 2); x0 = (double)((sx.e1 - a1 + b1) / 2); x1 = (double)(sx.e2 - a2 + b2);
        tie_terms64(sy.tfill, a0, a1, a2); tie_terms64(sy.tfill - shrink, b0, b1, b2);
        ytie = (double)((sy.e0 - a0 + b0) / 2); y0 = (double)((sy.e1 - a1 + b1) / 2); y1 = (double)(sy.e2 - a2 + b2);
        ntie = (double)(others + (g2 >= 2 ? g2 * (g2 - 1) / 2 : 0));
      } else {
        uint32_t a0, a1, a2, b0, b1, b2;
        tie_terms32(sx.tfill, a0, a1, a2); tie_terms32((int)(sx.tfill - shrink), b0, b1, b2);
        xtie = (double)((int32_t)(sx.s0 - a0 + b0) / 2); x0 = (double)((int32_t)(sx.s1 - a1 + b1) / 2);
        x1 = (double)(int32_t)(sx.s2 - a2 + b2);
        tie_terms32(sy.tfill, a0, a1, a2); tie_terms32((int)(sy.tfill - shrink), b0, b1, b2);
        ytie = (double)((int32_t)(sy.s0 - a0 + b0) / 2); y0 = (double)((int32_t)(sy.s1 - a1 + b1) / 2);
        y1 = (double)(int32_t)(sy.s2 - a2 + b2);
        // sum((cnt * (cnt - 1)) / 2) in int32 (:267): only a cell of >= 46342 rows can wrap, and a pair has at
        // most one (n <= 65535).  It is the cell of the two columns' largest tie groups: the (fill, fill) cell is
        // known (g); any other one is counted here, row by row -- both columns must have a group that large, rare
        const int32_t cell = (g2 >= 2) ? ((int32_t)((uint32_t)g2 * (uint32_t)(g2 - 1)) / 2) : 0;
        long long rest = others;
        int32_t cell2 = 0;
        const int mgx = (int)((uint32_t)sx.maxgroup >> 16), mgy = (int)((uint32_t)sy.maxgroup >> 16);
        const uint32_t lbx = (uint32_t)sx.maxgroup & 0xFFFFu, lby = (uint32_t)sy.maxgroup & 0xFFFFu;
        const bool fill_fill = (sx.nna > 0 && lbx == 0u) && (sy.nna > 0 && lby == 0u);
        if (mgx >= 46342 && mgy >= 46342 && !fill_fill) {
          const int cx = pi[p], cy = pj[p];
          const uint32_t* rx = pv.rec + ((int64_t)(cx >> 1) * pv.rec_rows) * 2 + (cx & 1);
          const uint32_t* ry = pv.rec + ((int64_t)(cy >> 1) * pv.rec_rows) * 2 + (cy & 1);
          long long c = 0;
          for (int r = 0; r < pv.n; ++r) c += ((rx[2 * r] >> 16) == lbx && (ry[2 * r] >> 16) == lby) ? 1 : 0;
          if (c >= 46342) {
            rest -= c * (c - 1) / 2;
            cell2 = (int32_t)((uint32_t)c * (uint32_t)(c - 1)) / 2;
          }
        }
        ntie = (double)(int32_t)((uint32_t)rest + (uint32_t)cell + (uint32_t)cell2);
      }
      const long long dis = (long long)rw.dis;  // both-missing rows are never discordant
      const long long tot = ne * (ne - 1) / 2;                                // :280
      cnt[0] = ne; cnt[1] = missing; cnt[2] = dis; cnt[3] = (long long)ntie;
      cnt[4] = (long long)xtie; cnt[5] = (long long)ytie; cnt[6] = (long long)x0; cnt[7] = (long long)x1;
      cnt[8] = (long long)y0; cnt[9] = (long long)y1; cnt[10] = tot;
      if (xtie == (double)tot || ytie == (double)tot) {
        reason = 4;                                                           // :291-298
      } else {
        const double dtot = (double)tot;
        const double con_minus_dis = dtot - xtie - ytie + ntie - 2.0 * (double)dis;  // :300
        const double den = sqrt((dtot - xtie) * (dtot - ytie));
        double tau = con_minus_dis / den;
        const double con_plus_dis = dtot - xtie - ytie + ntie;
        const double tau_max = con_plus_dis / den;                            // not clipped (:303)
        if (tau > 1) tau = 1; else if (tau < -1) tau = -1;
        const long long m = ne * (ne - 1);                                    // :310
        const double var = (((double)(m * (2 * ne + 5)) - x1 - y1) / 18 + (2 * xtie * ytie) / (double)m +
                            x0 * y0 / (double)(9 * m * (ne - 2)));            // :311-312
        double s_adj = tau * sqrt(((double)(m / 2) - xtie) * ((double)(m / 2) - ytie));  // :315
        if (continuity) {
          const double sg = s_adj > 0 ? 1.0 : (s_adj == 0 ? 0.0 : -1.0);
          s_adj = sg * (fabs(s_adj) - 1);                                     // :316-319
        }
        const double z = s_adj / sqrt(var);
        double pval = 0.0, plo, pup;
        pnorm_tails(z, plo, pup);
        if (alternative == 1) pval = plo;                                     // "less"
        else if (alternative == 2) pval = pup;                                // "greater"
        else if (alternative == 0) {                                          // "two.sided": 2 * min
          double mn = plo;
          if (!(plo != plo)) { if (pup != pup) mn = pup; else if (pup < mn) mn = pup; }
          pval = 2 * mn;
        }
        o_tau = tau; o_p = pval; o_tmax = tau_max; o_comp = completeness;
      }
    }
  }
  out4[4 * p + 0] = o_tau;
  out4[4 * p + 1] = o_p;
  out4[4 * p + 2] = o_tmax;
  out4[4 * p + 3] = o_comp;
  if (reasons) reasons[p] = reason;
  if (counts) {
    for (int f = 0; f < ICIKT_CNT_FIELDS_; ++f) counts[p * ICIKT_CNT_FIELDS_ + f] = cnt[f];
  }
}

// ------------------------------------------------------------------------------------------------
// pairwise missingness (pairwise_completeness, R/kendalltau.R:611-629): popcount of mask OR
// ------------------------------------------------------------------------------------------------
// The mask-only pre-pass: the missing-row bitset of a column and nothing else -- no sort, no statistics.  The
// reference's missing_either (R/kendalltau.R:626-629) never sorts either; the full pre-pass cost 13 ms of sorting per
// c5 matrix here for bitsets that one streaming pass delivers.  One workgroup per (column, slab of 64 words).
constexpr int KM_WORDS = 64;   // words (= waves' 64-row steps) per workgroup: 4 096 rows, 32 KB of the column
__global__ void __launch_bounds__(256)
k0_mask(PrepView pv, const double* __restrict__ X, int64_t ld, int col_begin) {
  const int c = col_begin + (int)blockIdx.x;
  const double* col = X + (int64_t)c * ld;
  unsigned long long* mask = pv.col_mask(c);
  const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
  const int w0 = (int)blockIdx.y * KM_WORDS;
  for (int w = w0 + wave; w < min(w0 + KM_WORDS, pv.W); w += 4) {
    const int i = w * 64 + lane;
    const double v = (i < pv.n) ? col[i] : 0.0;
    const unsigned long long b = __ballot(v != v);
    if (lane == 0) mask[w] = b;
  }
  if (blockIdx.y == 0 && threadIdx.x == 0) mask[pv.W] = 0ull;
}

__global__ void __launch_bounds__(256)
k_missingness(PrepView pv, const int32_t* __restrict__ pi, const int32_t* __restrict__ pj, int64_t n_pairs,
              int64_t* __restrict__ missing) {
  const int64_t p = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;  // one pair per wave
  if (p >= n_pairs) return;
  const uint32_t lane = lane_id();
  const unsigned long long* ma = pv.col_mask(pi[p]);
  const unsigned long long* mb = pv.col_mask(pj[p]);
  uint32_t c = 0;
  for (int w = lane; w < pv.W; w += 64) c += (uint32_t)__popcll(ma[w] | mb[w]);
  const unsigned long long tot = wave_sum_u64(c);
  if (lane == 0) missing[p] = (int64_t)tot;
}

// ------------------------------------------------------------------------------------------------
// kt_fast(use = "pairwise.complete.obs") (R/kendalltau.R:310-354, 448-545): per pair, rows with a missing value in
// EITHER vector are dropped.  Writes the pair's two columns with both entries of such rows missing; "local" then
// removes exactly those rows (src/kendallc.cpp:180-185) and nothing missing remains.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_mask_pairs(const double* __restrict__ X, int64_t ld, int n, const int32_t* __restrict__ pi,
             const int32_t* __restrict__ pj, int64_t first, double* __restrict__ Xp) {
  const int64_t p = blockIdx.y;   // pair of the chunk
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  const double a = X[(int64_t)pi[first + p] * ld + i], b = X[(int64_t)pj[first + p] * ld + i];
  const bool drop = (a != a) || (b != b);
  const double na = __longlong_as_double(0x7FF8000000000000ll);
  Xp[(2 * p) * (int64_t)n + i] = drop ? na : a;
  Xp[(2 * p + 1) * (int64_t)n + i] = drop ? na : b;
}

// ------------------------------------------------------------------------------------------------
// K1w: the pair kernel for wide columns (65 535 < n <= ICIKT_MAX_FEATURES_WIDE)
// ------------------------------------------------------------------------------------------------
// Same counting as k1_pairs -- walk the streamed column B in descending order, keep a bitset `seen` over the
// gathered column's positions of all rows whose B-group is strictly above, count "bits below lo" with a prefix over
// the words -- but with 32-bit positions from separate arrays and none of the fast paths: one pair per wave, one
// wave per workgroup, the flat prefix rebuilt after every batch (O(n / 64) per batch, as in round 1), in-batch pairs
// by whole-wave shifts.  A batch is either up to 64 rows of COMPLETE tie groups of B, or one group longer than 64
// rows, which is handled in two passes over its rows: (1) query `seen`, collect the rows in `pend`; (2) with a
// prefix over `pend`, every row counts the group's rows in its tie group [lo, hi] of A (k - 1 per row: every joint
// pair twice); then pend is merged into seen.  Throughput is not the point of this kernel (n = 100 000: ~10^5 pairs/s
// per GPU); it exists so that columns of any practical length are accepted, like the reference accepts them.
__device__ __forceinline__ void wide_rebuild(const unsigned long long* bits, uint32_t* pre, int Wp, uint32_t lane) {
  const int per = (Wp + 63) >> 6;                 // words per lane, contiguous
  const int w0 = (int)lane * per, w1 = min(Wp, w0 + per);
  uint32_t run = 0;
  for (int w = w0; w < w1; ++w) run += (uint32_t)__popcll(bits[w]);
  uint32_t below = wave_incl_scan(run) - run;
  for (int w = w0; w < w1; ++w) {
    pre[w] = below;
    below += (uint32_t)__popcll(bits[w]);
  }
  if (lane == 63u) pre[Wp] = below;               // total (the guard entry: a query at position 64 Wp)
}
__device__ __forceinline__ uint32_t wide_query(const unsigned long long* bits, const uint32_t* pre, uint32_t pos) {
  const uint32_t w = pos >> 6;
  return pre[w] + (uint32_t)__popcll(bits[w] & low_mask64(pos & 63u));
}

__global__ void __launch_bounds__(64)
k1_wide(PrepView pv, const int32_t* __restrict__ pi, const int32_t* __restrict__ pj, PairRaw* __restrict__ raw,
        long long n_pairs, int* __restrict__ task_ctr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
  const uint32_t lane = threadIdx.x & 63u;
  const int n = pv.n, W = pv.W, Wp = pv.Wp;
  unsigned long long* seen = reinterpret_cast<unsigned long long*>(wsm);          // Wp + 1 words each
  unsigned long long* pend = seen + (Wp + 1);
  uint32_t* pre = reinterpret_cast<uint32_t*>(pend + (Wp + 1));                   // Wp + 1 entries each
  uint32_t* ppre = pre + (Wp + 1);
  for (;;) {
    long long p = 0;
    if (lane == 0u) p = (long long)atomicAdd(task_ctr, 1);
    p = ((long long)__builtin_amdgcn_readfirstlane((int)(p >> 32)) << 32) |
        (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)p);
    if (p >= n_pairs) return;
    const int ca = __builtin_amdgcn_readfirstlane(pi[p]), cb_ = __builtin_amdgcn_readfirstlane(pj[p]);
    const uint32_t* ordB = pv.order32 + (int64_t)cb_ * pv.n_pad;
    const unsigned long long* gf = pv.col_gflag(cb_);
    const uint32_t* qA = pv.q32 + (int64_t)ca * pv.n_pad;
    const uint32_t* loA = pv.lo32 + (int64_t)ca * pv.n_pad;
    const uint32_t* hiA = pv.hi32 + (int64_t)ca * pv.n_pad;
    for (int w = (int)lane; w <= Wp; w += 64) { seen[w] = 0ull; pend[w] = 0ull; pre[w] = 0u; ppre[w] = 0u; }
    wave_lds_fence();
    unsigned long long dis = 0, tie = 0, tie2 = 0;   // per lane
    int pos = 0;
    while (pos < n) {
      // flags of positions pos .. pos + 63 and of position pos + 64 (gf[W] is a zero guard word)
      const int wc = pos >> 6, fb = pos & 63;
      const unsigned long long w0 = gf[min(wc, W)], w1 = gf[min(wc + 1, W)];
      unsigned long long F = fb ? ((w0 >> fb) | (w1 << (64 - fb))) : w0;
      const bool fnbit = ((w1 >> fb) & 1ull) != 0ull;
      const int remaining = n - pos;
      const int avail = min(remaining, 64);
      if (avail < 64) F &= (1ull << avail) - 1ull;
      // group starts after position pos inside the window; the position after the data counts as a start
      unsigned long long starts = F & ~1ull;
      if (avail < 64) starts |= 1ull << avail;
      const bool end_at_64 = (avail == 64) && (fnbit || remaining == 64);
      if (starts == 0ull && !end_at_64) {
        // ---- one group longer than the window: find its end, then two passes over its rows ------------------
        int gend = pos + 64;
        for (;;) {   // first start at or after gend (wave-uniform scalar scan of the flag words)
          if (gend >= n) { gend = n; break; }
          const int gw = gend >> 6, gb = gend & 63;
          const unsigned long long m = gf[gw] >> gb;
          if (m != 0ull) { gend = min(n, gend + (int)__builtin_ctzll(m)); break; }
          gend = (gw + 1) << 6;
        }
        for (int b = pos; b < gend; b += 64) {
          const int k = b + (int)lane;
          if (k < gend) {
            const uint32_t row = ordB[k];
            const uint32_t q = qA[row], lo = loA[row];
            dis += wide_query(seen, pre, lo);
            seen_insert(pend, q);
          }
        }
        wave_lds_fence();
        wide_rebuild(pend, ppre, Wp, lane);
        wave_lds_fence();
        for (int b = pos; b < gend; b += 64) {
          const int k = b + (int)lane;
          if (k < gend) {
            const uint32_t row = ordB[k];
            const uint32_t lo = loA[row], hi = hiA[row];
            const uint32_t cell = wide_query(pend, ppre, hi + 1u) - wide_query(pend, ppre, lo);
            tie2 += cell - 1u;     // the row itself is in its cell
          }
        }
        wave_lds_fence();
        for (int w = (int)lane; w < Wp; w += 64) { seen[w] |= pend[w]; pend[w] = 0ull; }
        wave_lds_fence();
        wide_rebuild(seen, pre, Wp, lane);
        wave_lds_fence();
        pos = gend;
        continue;
      }
      // ---- a batch of complete groups: rows pos .. pos + nact - 1 ------------------------------------------------
      int nact;
      if (end_at_64) nact = 64;
      else nact = 63 - (int)__builtin_clzll(starts);      // the last start inside the window ends the batch
      const bool valid = (int)lane < nact;
      uint32_t q = 0xFFFFFFFFu, lo = 0u;
      if (valid) {
        const uint32_t row = ordB[pos + (int)lane];
        q = qA[row];
        lo = loA[row];
      }
      const uint32_t gid = (uint32_t)__popcll(F & ((2ull << lane) - 1ull));   // number of the lane's group in the batch
      if (valid) dis += wide_query(seen, pre, lo);
      uint32_t sq = q, sg = gid, slo = lo;
      if (nact == 64 && F == ~0ull) {
        // 64 rows, each its own group (continuous data): the 39-compare all-pairs count of the one-pair kernels
        // (32-bit compares: positions of any width) instead of 63 shifts
        dis += wave_allpairs(q, lo, lane);
      } else
      for (int d = 1; d < nact; ++d) {
        sq = dpp_wave_shr1(0xFFFFFFFFu, sq);     // lane l now holds lane l - d
        sg = dpp_wave_shr1(0xFFFFFFFFu, sg);
        slo = dpp_wave_shr1(0xFFFFFFFFu, slo);
        if (valid && (int)lane >= d) {
          if (sg != gid) dis += (sq < lo) ? 1u : 0u;       // an earlier group of the batch: is its row below mine in A?
          else tie += (slo == lo) ? 1u : 0u;               // my own group: a joint tie?
        }
      }
      wave_lds_fence();
      if (valid) seen_insert(seen, q);
      wave_lds_fence();
      wide_rebuild(seen, pre, Wp, lane);
      wave_lds_fence();
      pos += nact;
    }
    // rows missing in both columns / in both fill groups
    const unsigned long long* ma = pv.col_mask(ca);
    const unsigned long long* mb = pv.col_mask(cb_);
    const unsigned long long* fa = pv.col_fillmask(ca);
    const unsigned long long* fbm = pv.col_fillmask(cb_);
    unsigned long long cboth = 0, gboth = 0;
    for (int w = (int)lane; w < W; w += 64) {
      cboth += (unsigned long long)__popcll(ma[w] & mb[w]);
      gboth += (unsigned long long)__popcll(fa[w] & fbm[w]);
    }
    const unsigned long long dis_t = wave_sum_u64(dis);
    const unsigned long long tie_t = wave_sum_u64(tie) + (wave_sum_u64(tie2) >> 1);
    const unsigned long long cb_t = wave_sum_u64(cboth), gg_t = wave_sum_u64(gboth);
    if (lane == 0u) {
      PairRaw o;
      o.dis = dis_t;
      o.ntie = tie_t;
      o.c_both = (uint32_t)cb_t;
      o.g = (uint32_t)gg_t;
      raw[p] = o;
    }
    wave_lds_fence();
  }
}

// ------------------------------------------------------------------------------------------------
// self-test of the DPP primitives
// ------------------------------------------------------------------------------------------------
__global__ void k_selftest(uint32_t* out) {
  const uint32_t lane = lane_id();
  out[lane] = wave_incl_scan(lane + 1u);                       // (lane+1)(lane+2)/2
  out[64 + lane] = dpp_wave_shr1(0xABCDu, lane * 3u);          // lane 0: 0xABCD, else 3*(lane-1)
  uint32_t v = dpp_wave_shr1(0xFFFFFFFFu, lane);
  for (int s = 1; s < 41; ++s) v = dpp_wave_shr1(v, v);        // in-place chain, crosses DPP rows
  out[128 + lane] = v;                                         // lanes < 41: ~0, else lane-41
  // half-wave pieces: q, lo from a fixed pseudo-random table; the host recomputes the counts
  const uint32_t q = (lane * 2654435761u >> 20) & 0xFFFu, lo = ((lane * 40503u + 977u) >> 3) & 0xFFFu;
  out[192 + lane] = half_count(q, lo, lane);  // summed per half by the host
  {  // the packed form of two sub-steps at once: rows (q, lo) and (q2, lo2) must give both counts together
    const uint32_t q2 = (lane * 1103515245u >> 17) & 0x27BFu, lo2 = ((lane * 69069u + 12345u) >> 5) & 0x27BFu;
    out[576 + lane] = half_step_count(q | (lo << 16), q2 | (lo2 << 16), lane, half_partner_addr(lane));
  }
  const auto sw = __builtin_amdgcn_permlane32_swap(lane, 100u + lane, false, false);
  out[256 + lane] = sw[0];                                     // lanes < 32: lane, else 100 + (lane - 32)
  out[320 + lane] = sw[1];                                     // lanes < 32: 32 + lane, else 100 + lane
  out[384 + lane] = half_incl_scan(lane + 1u);
  // lane_xor<J>(lane * 5 + 1) must be (lane ^ J) * 5 + 1 for every J; packed as a 6-bit pass mask
  const uint32_t t5 = lane * 5u + 1u;
  out[448 + lane] = (lane_xor<1>(t5, lane) == ((lane ^ 1u) * 5u + 1u) ? 1u : 0u) |
                    (lane_xor<2>(t5, lane) == ((lane ^ 2u) * 5u + 1u) ? 2u : 0u) |
                    (lane_xor<4>(t5, lane) == ((lane ^ 4u) * 5u + 1u) ? 4u : 0u) |
                    (lane_xor<8>(t5, lane) == ((lane ^ 8u) * 5u + 1u) ? 8u : 0u) |
                    (lane_xor<16>(t5, lane) == ((lane ^ 16u) * 5u + 1u) ? 16u : 0u) |
                    (lane_xor<32>(t5, lane) == ((lane ^ 32u) * 5u + 1u) ? 32u : 0u);
  out[512 + lane] = 0u;
}

// ------------------------------------------------------------------------------------------------
// launchers (called from icikt_capi.cpp; keeps <<<>>> syntax inside the .hip translation unit)
// ------------------------------------------------------------------------------------------------
// (every launcher first drops whatever error an earlier, unrelated HIP call of the calling thread left behind: the
//  hipGetLastError() after the launch must report THIS launch)
hipError_t launch_k0(const PrepView& pv, const double* dX, int64_t ld, int col_begin, int ncols, const MaskSpec* msp,
                     uint8_t* keep, int small_shape, hipStream_t s) {
  (void)hipGetLastError();
  MaskSpec ms{};
  if (msp) ms = *msp;
  if (pv.wide)
    hipLaunchKernelGGL(k0_prepare_wide, dim3(ncols), dim3(K0_THREADS), 0, s, pv, dX, ld, col_begin, ms, keep);
  else if (small_shape)   // 4 waves per column: fits beside a running pair kernel (the later chunks of the pipelined host path)
    hipLaunchKernelGGL(k0_prepare_small, dim3(ncols), dim3(K0_THREADS_SMALL), 0, s, pv, dX, ld, col_begin, ms, keep);
  else if (pv.npow2 >= 2 * K0_TILE)
    hipLaunchKernelGGL(k0_prepare_large8, dim3(ncols), dim3(K0_THREADS), 0, s, pv, dX, ld, col_begin, ms, keep);
  else
    hipLaunchKernelGGL(k0_prepare_large, dim3(ncols), dim3(K0_THREADS), 0, s, pv, dX, ld, col_begin, ms, keep);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Full-matrix assembly: scale_and_reshape (R/kendalltau.R:357-421) on the device
// ------------------------------------------------------------------------------------------------
// The reference row-binds the chunks' data.frames, divides raw by max(taumax, na.rm = TRUE) (:368-373), appends one
// row per sample for the diagonal (raw = cor = n_good / max(n_good), pvalue 0, taumax 1, completeness =
// n_good / n_feature, :375-386) and fills five S x S matrices symmetrically by name pair (:390-415): on the host that
// is a 523 776-row data.frame and ten indexed assignments at c4.  Here: one reduction kernel, one scatter kernel,
// one D2H of 5 S^2 doubles.
__device__ __forceinline__ unsigned long long dbl_sortable(double v) {   // monotone in v; 0 is below every double
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double dbl_unsortable(unsigned long long k) {
  return __longlong_as_double((long long)((k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k));
}
__device__ __forceinline__ long long col_n_good(const PrepView& pv, const int64_t* n_good, int c) {
  return n_good ? (long long)n_good[c] : (long long)pv.n - pv.col_stats(c)->nexcl;
}

// red[0] = max over the pairs of the sortable key of taumax (NaN skipped; 0 = no pair had one), red[1 + r] = pairs
// with reason code r (0..4), red[6] = max(n_good)
__global__ void __launch_bounds__(256)
k_out_stats(PrepView pv, const double* __restrict__ out4, const int32_t* __restrict__ reasons, int64_t n_pairs,
            const int64_t* __restrict__ n_good, unsigned long long* __restrict__ red) {
  __shared__ unsigned long long sh[8];
  if (threadIdx.x < 8) sh[threadIdx.x] = 0ull;
  __syncthreads();
  unsigned long long mx = 0ull, mg = 0ull;
  uint32_t rc[5] = {0u, 0u, 0u, 0u, 0u};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_pairs; p += stride) {
    const double t = out4[4 * p + 2];
    if (t == t) mx = max(mx, dbl_sortable(t));
    const int r = reasons ? reasons[p] : 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) rc[k] += (r == k) ? 1u : 0u;
  }
  for (int64_t cc = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; cc < pv.n_samp; cc += stride)
    mg = max(mg, (unsigned long long)max(0ll, col_n_good(pv, n_good, (int)cc)));
  atomicMax(&sh[0], mx);
  atomicMax(&sh[6], mg);
#pragma unroll
  for (int k = 0; k < 5; ++k)
    if (rc[k]) atomicAdd(&sh[1 + k], (unsigned long long)rc[k]);
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicMax(&red[0], sh[0]);
    atomicMax(&red[6], sh[6]);
  }
  if (threadIdx.x >= 1 && threadIdx.x <= 5 && sh[threadIdx.x]) atomicAdd(&red[threadIdx.x], sh[threadIdx.x]);
}

// pair p of combn(S, 2): row i holds the pairs (i, i+1 .. S-1) and starts at offset i (2S - i - 1) / 2
__device__ __forceinline__ void combn_pair(int64_t S, int64_t t, int64_t& i, int64_t& j) {
  const double b = 2.0 * (double)S - 1.0;
  i = (int64_t)((b - sqrt(b * b - 8.0 * (double)t)) * 0.5);
  i = max((int64_t)0, min(i, S - 2));
  while (i > 0 && i * (2 * S - i - 1) / 2 > t) --i;
  while ((i + 1) * (2 * S - i - 2) / 2 <= t) ++i;
  j = i + 1 + (t - i * (2 * S - i - 1) / 2);
}
// out5: five S x S matrices, zero-filled by the caller; a thread per pair writes both triangles, then a thread per
// sample the diagonal (the diagonal rows come AFTER the pairs in the reference: they win over a self pair of the list)
__global__ void __launch_bounds__(256)
k_assemble(PrepView pv, const double* __restrict__ out4, const int32_t* __restrict__ pi, const int32_t* __restrict__ pj,
           int64_t n_pairs, const int64_t* __restrict__ n_good, const unsigned long long* __restrict__ red, int scale_max,
           int diag_good, double* __restrict__ out5) {
  const int64_t S = pv.n_samp, SS = S * S;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n_pairs) {
    int64_t i, j;
    if (pi) {
      i = pi[t]; j = pj[t];
    } else {
      combn_pair(S, t, i, j);
    }
    const double raw = out4[4 * t + 0], pval = out4[4 * t + 1], tmax = out4[4 * t + 2], comp = out4[4 * t + 3];
    // max(numeric(0), na.rm = TRUE) is -Inf in R
    const double max_cor = red[0] ? dbl_unsortable(red[0]) : -__longlong_as_double(0x7FF0000000000000ll);
    const double cor = scale_max ? raw / max_cor : raw;
    const int64_t a = i + j * S, b2 = j + i * S;
    out5[a] = cor;            out5[b2] = cor;
    out5[SS + a] = raw;       out5[SS + b2] = raw;
    out5[2 * SS + a] = pval;  out5[2 * SS + b2] = pval;
    out5[3 * SS + a] = tmax;  out5[3 * SS + b2] = tmax;
    out5[4 * SS + a] = comp;  out5[4 * SS + b2] = comp;
  }
}
__global__ void __launch_bounds__(256)
k_assemble_diag(PrepView pv, const int64_t* __restrict__ n_good, const unsigned long long* __restrict__ red,
                double* __restrict__ out5) {
  const int64_t S = pv.n_samp, SS = S * S;
  const int64_t cc = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (cc >= S) return;
  const double g = (double)col_n_good(pv, n_good, (int)cc);
  const double d = g / (double)red[6];                       // n_good / max(n_good)  (0 / 0 = NaN, as in R)
  const int64_t a = cc + cc * S;
  out5[a] = d;
  out5[SS + a] = d;
  out5[2 * SS + a] = 0.0;
  out5[3 * SS + a] = 1.0;
  out5[4 * SS + a] = g / (double)pv.n;                       // frac_complete = n_good / nrow
}

__global__ void __launch_bounds__(256)
k_fill_combn(int32_t* __restrict__ pi, int32_t* __restrict__ pj, int64_t S, int64_t begin, int64_t count) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  int64_t i, j;
  combn_pair(S, begin + t, i, j);
  pi[t] = (int32_t)i;
  pj[t] = (int32_t)j;
}
hipError_t launch_fill_combn(int32_t* pi, int32_t* pj, int64_t S, int64_t begin, int64_t count, hipStream_t s) {
  if (count <= 0) return hipSuccess;
  (void)hipGetLastError();
  hipLaunchKernelGGL(k_fill_combn, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, pi, pj, S, begin, count);
  return hipGetLastError();
}

hipError_t launch_out_stats(const PrepView& pv, const double* out4, const int32_t* reasons, int64_t n_pairs,
                            const int64_t* n_good, unsigned long long* red, hipStream_t s) {
  (void)hipGetLastError();
  hipError_t e = hipMemsetAsync(red, 0, 8 * sizeof(unsigned long long), s);
  if (e != hipSuccess) return e;
  const int64_t work = std::max<int64_t>(n_pairs, pv.n_samp);
  const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(2048, (work + 255) / 256));
  hipLaunchKernelGGL(k_out_stats, dim3(blocks), dim3(256), 0, s, pv, out4, reasons, n_pairs, n_good, red);
  return hipGetLastError();
}

hipError_t launch_assemble(const PrepView& pv, const double* out4, const int32_t* pi, const int32_t* pj, int64_t n_pairs,
                           const int64_t* n_good, const unsigned long long* red, int scale_max, int diag_good,
                           double* out5, hipStream_t s) {
  const size_t S = (size_t)pv.n_samp;
  if (S == 0) return hipSuccess;
  (void)hipGetLastError();
  hipError_t e = hipMemsetAsync(out5, 0, 5 * S * S * sizeof(double), s);
  if (e != hipSuccess) return e;
  if (n_pairs > 0)
    hipLaunchKernelGGL(k_assemble, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, s, pv, out4, pi, pj, n_pairs,
                       n_good, red, scale_max, diag_good, out5);
  if (diag_good)
    hipLaunchKernelGGL(k_assemble_diag, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, s, pv, n_good, red, out5);
  return hipGetLastError();
}

hipError_t launch_k1_wide(const PrepView& pv, const int32_t* pi, const int32_t* pj, PairRaw* raw, int64_t n_pairs,
                          int blocks, size_t lds_bytes, int* task_ctr, hipStream_t s) {
  if (n_pairs <= 0 || blocks <= 0) return hipSuccess;
  (void)hipGetLastError();
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k1_wide), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k1_wide, dim3(blocks), dim3(64), lds_bytes, s, pv, pi, pj, raw, (long long)n_pairs, task_ctr);
  return hipGetLastError();
}
hipError_t k1_wide_blocks_per_cu(size_t lds_bytes, int* out) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k1_wide), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds_bytes);
  if (e != hipSuccess) return e;
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(out, reinterpret_cast<const void*>(k1_wide), 64, lds_bytes);
}

// the records of the pairs of a task range cleared: a launch whose tasks are cut in segments ADDS its counts up
__global__ void __launch_bounds__(256) k_zero_raw(const int32_t* __restrict__ tasks, int n_tasks, PairRaw* __restrict__ raw) {
  const int t = (int)(blockIdx.x * 256 + threadIdx.x);
  if (t >= n_tasks) return;
  PairRaw z;
  z.dis = 0ull; z.ntie = 0ull; z.c_both = 0u; z.g = 0u;
  const int p0 = tasks[2 * t], p1 = tasks[2 * t + 1];
  raw[p0] = z;
  if (p1 >= 0) raw[p1] = z;
}
hipError_t launch_zero_raw(const int32_t* tasks, int n_tasks, PairRaw* raw, hipStream_t s) {
  if (n_tasks <= 0) return hipSuccess;
  (void)hipGetLastError();
  hipLaunchKernelGGL(k_zero_raw, dim3((n_tasks + 255) / 256), dim3(256), 0, s, tasks, n_tasks, raw);
  return hipGetLastError();
}

typedef void (*k1_fn_t)(PrepView, const int32_t*, int, const int32_t*, const int32_t*, PairRaw*, int, int*, int);

// Half-wave kernels: one per half_items (1..ICIKT_HALF_ITEMS_MAX words per lane in a half's prefix rebuild:
// n <= 2 040 * half_items - 24); half_items == 0: pairs on the whole wave, one or two per wave
static k1_fn_t k1_select(int np, int half_items) {
  if (np == 2) {
    switch (half_items) {
      case 0: return &k1_pairs<2, 0>;
      case 1: return &k1_pairs<2, 1>;
      case 2: return &k1_pairs<2, 2>;
      case 3: return &k1_pairs<2, 3>;
      case 4: return &k1_pairs<2, 4>;
      case 5: return &k1_pairs<2, 5>;
      case 6: return &k1_pairs<2, 6>;
      case 7: return &k1_pairs<2, 7>;
      case 9: return &k1_pairs<2, 9>;
      case 11: return &k1_pairs<2, 11>;
      case 13: return &k1_pairs<2, 13>;
      case 15: return &k1_pairs<2, 15>;
      default: return nullptr;
    }
  }
  if (np != 1 || half_items != 0) return nullptr;
  return &k1_pairs<1, 0>;
}

hipError_t launch_k1(const PrepView& pv, const int32_t* tasks, int n_tasks, const int32_t* pi,
                     const int32_t* pj, PairRaw* raw, int np, int half_items, int wpb, int blocks,
                     size_t lds_bytes, int perpair_bytes, int* task_ctr, int opts, hipStream_t s) {
  if (n_tasks <= 0 || blocks <= 0) return hipSuccess;
  k1_fn_t fn = k1_select(np, half_items);
  if (!fn) return hipErrorInvalidValue;
  (void)hipGetLastError();
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(wpb * 64), lds_bytes, s, pv, tasks, n_tasks, pi, pj, raw,
                     perpair_bytes, task_ctr, opts);
  return hipGetLastError();
}

// resident workgroups per CU of the pair kernel for a launch shape (occupancy query)
hipError_t k1_blocks_per_cu(int np, int half_items, int wpb, size_t lds_bytes, int* out) {
  k1_fn_t fn = k1_select(np, half_items);
  if (!fn) return hipErrorInvalidValue;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds_bytes);
  if (e != hipSuccess) return e;
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(out, reinterpret_cast<const void*>(fn), wpb * 64, lds_bytes);
}

hipError_t launch_k2(const PrepView& pv, const int32_t* pi, const int32_t* pj, const PairRaw* raw,
                     int64_t n_pairs, int perspective, int alternative, int continuity, int exact64,
                     double* out4, int64_t* counts, int32_t* reasons, hipStream_t s) {
  if (n_pairs <= 0) return hipSuccess;
  (void)hipGetLastError();
  const int threads = 256;
  const int64_t blocks = (n_pairs + threads - 1) / threads;
  hipLaunchKernelGGL(k2_epilogue, dim3((unsigned)blocks), dim3(threads), 0, s, pv, pi, pj, raw, n_pairs,
                     perspective, alternative, continuity, exact64, out4, counts, reasons);
  return hipGetLastError();
}

hipError_t launch_k0_expand(const PrepView& pv, int col_begin, int ncols, hipStream_t s) {
  if (ncols <= 0 || pv.n <= 0) return hipSuccess;
  (void)hipGetLastError();
  const int staged = (pv.n_pad <= 12288) ? 1 : 0;
  const size_t lds = staged ? (size_t)pv.n_pad * 8 : 0;
  if (staged) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k0_expand), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k0_expand, dim3(ncols), dim3(64 * KX_WAVES), lds, s, pv, col_begin, ncols, staged);
  return hipGetLastError();
}

hipError_t launch_k0_mask(const PrepView& pv, const double* dX, int64_t ld, int col_begin, int ncols, hipStream_t s) {
  if (ncols <= 0) return hipSuccess;
  (void)hipGetLastError();
  const unsigned slabs = (unsigned)std::max(1, (pv.W + KM_WORDS - 1) / KM_WORDS);
  hipLaunchKernelGGL(k0_mask, dim3((unsigned)ncols, slabs), dim3(256), 0, s, pv, dX, ld, col_begin);
  return hipGetLastError();
}

hipError_t launch_missingness(const PrepView& pv, const int32_t* pi, const int32_t* pj, int64_t n_pairs,
                              int64_t* missing, hipStream_t s) {
  if (n_pairs <= 0) return hipSuccess;
  (void)hipGetLastError();
  const int threads = 256;
  const int64_t blocks = (n_pairs * 64 + threads - 1) / threads;
  hipLaunchKernelGGL(k_missingness, dim3((unsigned)blocks), dim3(threads), 0, s, pv, pi, pj, n_pairs, missing);
  return hipGetLastError();
}

hipError_t launch_mask_pairs(const double* dX, int64_t ld, int n, const int32_t* pi, const int32_t* pj, int64_t first,
                             int64_t npairs, double* dXp, hipStream_t s) {
  if (npairs <= 0 || n <= 0) return hipSuccess;
  (void)hipGetLastError();
  hipLaunchKernelGGL(k_mask_pairs, dim3((unsigned)((n + 255) / 256), (unsigned)npairs), dim3(256), 0, s, dX, ld, n, pi,
                     pj, first, dXp);
  return hipGetLastError();
}

// diagnostic build: copy out (and optionally clear) the step statistics; hipErrorNotSupported in the product build
hipError_t read_step_stats(unsigned long long* out24, int reset) {
#ifdef ICIKT_STEP_STATS
  hipError_t e = hipMemcpyFromSymbol(out24, HIP_SYMBOL(g_step_stats), 24 * sizeof(unsigned long long));
  if (e == hipSuccess && reset) {
    unsigned long long z[24] = {};
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_step_stats), z, sizeof(z));
  }
  return e;
#else
  (void)out24; (void)reset;
  return hipErrorNotSupported;
#endif
}

hipError_t launch_selftest(uint32_t* d_out, hipStream_t s) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(k_selftest, dim3(1), dim3(64), 0, s, d_out);
  return hipGetLastError();
}

}  // namespace icikt
