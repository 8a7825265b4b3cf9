// Microbenchmark of the in-step all-pairs loop variants (issue cost per step), gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 -o gpurun_out/allpairs_ubench tools/ubench/allpairs_ubench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t dpp_wave_shr1(uint32_t old, uint32_t src) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t dpp_row_shr1(uint32_t old, uint32_t src) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, 0x111, 0xf, 0xf, false);
}

#define REP2(x) x x
#define REP4(x) REP2(x) REP2(x)
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP8(x) REP8(x)
#define REP32(x) REP16(x) REP16(x)
#define REP62(x) REP32(x) REP16(x) REP8(x) REP4(x) REP2(x)

#define S1 "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_cmp_lt_u32_e32 vcc, %0, %2\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
#define S1ROW "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_cmp_lt_u32_e32 vcc, %0, %2\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
#define S1NOP "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_cmp_lt_u32_e32 vcc, %0, %2\n\ts_nop 0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
#define PK "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_pk_sub_u16 %2, %3, %0 clamp\n\tv_pk_min_u16 %2, %2, %4\n\tv_pk_add_u16 %1, %1, %2\n\t"
// two independent single chains interleaved (vcc and an SGPR pair)
#define S2 "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
           "v_cmp_lt_u32_e32 vcc, %0, %5\n\tv_cmp_lt_u32_e64 %4, %2, %6\n\t" \
           "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\tv_addc_co_u32_e64 %3, %4, 0, %3, %4\n\t"
// two packed chains interleaved
#define PK2 "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
            "v_pk_sub_u16 %4, %6, %0 clamp\n\tv_pk_sub_u16 %5, %7, %2 clamp\n\t" \
            "v_pk_min_u16 %4, %4, %8\n\tv_pk_min_u16 %5, %5, %8\n\t" \
            "v_pk_add_u16 %1, %1, %4\n\tv_pk_add_u16 %3, %3, %5\n\t"
// ballot form: cmp to SGPR pair, scalar popcount and add
#define BAL "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_cmp_lt_u32_e64 %3, %0, %2\n\ts_bcnt1_i32_b64 %4, %3\n\ts_add_u32 %1, %1, %4\n\t"

template <int V>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters) {
  const uint32_t lane = threadIdx.x & 63;
  uint32_t q = (lane * 2654435761u) >> 16, lo = (lane * 40503u + 77u) & 0xFFFF;
  uint32_t acc = 0, acc2 = 0, q2 = q ^ 0x5555, lo2 = lo ^ 0x3333, ones = 0x00010001u;
  asm volatile("" : "+v"(ones));
  uint32_t sacc = 0;
  for (int it = 0; it < iters; ++it) {
    uint32_t qs = q + it, qs2 = q2 + it;
    if (V == 0) {  // hipcc's own code for the loop
      uint32_t t = dpp_wave_shr1(0xFFFFFFFFu, qs);
      uint32_t c = (t < lo) ? 1u : 0u;
#pragma unroll
      for (int s = 2; s < 64; ++s) { t = dpp_wave_shr1(t, t); c += (t < lo) ? 1u : 0u; }
      acc += c;
    } else if (V == 1) {
      asm volatile(REP62(S1) : "+v"(qs), "+v"(acc) : "v"(lo) : "vcc");
    } else if (V == 2) {
      asm volatile(REP62(S1ROW) : "+v"(qs), "+v"(acc) : "v"(lo) : "vcc");
    } else if (V == 3) {
      asm volatile(REP62(S1NOP) : "+v"(qs), "+v"(acc) : "v"(lo) : "vcc");
    } else if (V == 4) {
      uint32_t d;
      asm volatile(REP62(PK) : "+v"(qs), "+v"(acc), "=&v"(d) : "v"(lo), "v"(ones));
    } else if (V == 5) {
      unsigned long long sp;
      asm volatile(REP62(S2) : "+v"(qs), "+v"(acc), "+v"(qs2), "+v"(acc2), "=&s"(sp) : "v"(lo), "v"(lo2) : "vcc");
    } else if (V == 6) {
      uint32_t d1, d2;
      asm volatile(REP62(PK2) : "+v"(qs), "+v"(acc), "+v"(qs2), "+v"(acc2), "=&v"(d1), "=&v"(d2) : "v"(lo), "v"(lo2), "v"(ones));
    } else if (V == 7) {
      unsigned long long sp; uint32_t st;
      asm volatile(REP62(BAL) : "+v"(qs), "+s"(sacc), "+v"(lo), "=&s"(sp), "=&s"(st));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + acc2 + sacc;
}

template <int V>
int run(const char* name, int pairs_per_step, int instr_per_step) {
  uint32_t* d;
  CHK(hipMalloc(&d, 256 * 8 * 256 * 4));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const int iters = 2000;
  for (int wpc : {4, 8, 16, 24, 32}) {  // waves per CU (256 CUs): blocks of 4 waves
    const int blocks = 256 * wpc / 4;
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, 10);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, iters);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    // steps executed per SIMD: waves per SIMD * iters * 62
    const double steps_per_simd = (double)wpc / 4.0 * iters * 62.0;
    const double ns_per_step = ms * 1e6 / steps_per_simd;
    printf("%-28s waves/CU %2d: %.3f ms, %.2f ns per step per SIMD = %.2f ns per pair-step, %.2f ns per instr\n", name, wpc, ms,
           ns_per_step, ns_per_step / pairs_per_step, ns_per_step / instr_per_step);
  }
  CHK(hipFree(d));
  return 0;
}

int main() {
  if (run<0>("hipcc loop (1 pair)", 1, 3)) return 1;
  if (run<1>("asm cmp/addc (1 pair)", 1, 3)) return 1;
  if (run<2>("asm row_shr cmp/addc", 1, 3)) return 1;
  if (run<3>("asm cmp/nop/addc", 1, 3)) return 1;
  if (run<4>("asm packed (2 pairs)", 2, 4)) return 1;
  if (run<5>("asm 2 chains (2 pairs)", 2, 6)) return 1;
  if (run<6>("asm 2 packed chains (4 pairs)", 4, 8)) return 1;
  if (run<7>("asm ballot+salu (1 pair)", 1, 4)) return 1;
  return 0;
}
