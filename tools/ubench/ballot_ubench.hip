// Does moving the accumulation of the in-step all-pairs loop to the scalar unit pay on gfx950?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define REP2(x) x x
#define REP4(x) REP2(x) REP2(x)
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP8(x) REP8(x)
#define REP32(x) REP16(x) REP16(x)
#define REP62(x) REP32(x) REP16(x) REP8(x) REP4(x) REP2(x)

#define PK "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_pk_sub_u16 %2, %3, %0 clamp\n\tv_pk_min_u16 %2, %2, %4\n\tv_pk_add_u16 %1, %1, %2\n\t"
// operands: %0 qs (v) %1 acc1 (s) %2 acc2 (s) %3 m1 (s64) %4 m2 (s64) %5 t (s) %6 lo1 (v) %7 lo2 (v)
#define SD2 "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
            "v_cmp_lt_u16_sdwa %3, %0, %6 src0_sel:WORD_0 src1_sel:WORD_0\n\t" \
            "v_cmp_lt_u16_sdwa %4, %0, %7 src0_sel:WORD_1 src1_sel:WORD_0\n\t" \
            "s_bcnt1_i32_b64 %5, %3\n\ts_add_u32 %1, %1, %5\n\ts_bcnt1_i32_b64 %5, %4\n\ts_add_u32 %2, %2, %5\n\t"
// single pair: %0 qs %1 acc (s) %2 m (s64) %3 t (s) %4 lo
#define SD1 "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_cmp_lt_u32_e64 %2, %0, %4\n\ts_bcnt1_i32_b64 %3, %2\n\ts_add_u32 %1, %1, %3\n\t"

template <int V>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters) {
  const uint32_t lane = threadIdx.x & 63;
  uint32_t q = (lane * 2654435761u) >> 17, lo = (lane * 40503u + 77u) & 0x7FFF, lo2 = lo ^ 0x1234;
  uint32_t acc = 0, ones = 0x00010001u, s1 = 0, s2 = 0;
  asm volatile("" : "+v"(ones));
  for (int it = 0; it < iters; ++it) {
    uint32_t qs = (q | (q << 16)) + it;
    if (V == 0) {
      uint32_t d;
      asm volatile(REP62(PK) : "+v"(qs), "+v"(acc), "=&v"(d) : "v"(lo | (lo2 << 16)), "v"(ones));
    } else if (V == 1) {
      unsigned long long m1, m2; uint32_t t;
      asm volatile(REP62(SD2) : "+v"(qs), "+s"(s1), "+s"(s2), "=&s"(m1), "=&s"(m2), "=&s"(t) : "v"(lo), "v"(lo2) : "scc");
    } else {
      unsigned long long m1; uint32_t t;
      asm volatile(REP62(SD1) : "+v"(qs), "+s"(s1), "=&s"(m1), "=&s"(t) : "v"(lo) : "scc");
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + s1 + s2;
}

template <int V>
int run(const char* name, int pairs, uint32_t* d) {
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const int iters = 3000;
  for (int wpc : {8, 16, 24, 32}) {
    const int blocks = 256 * wpc / 4;
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, iters);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, iters);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double steps_per_simd = (double)wpc / 4.0 * iters * 62.0;
    printf("%-34s waves/CU %2d: %7.3f ms  %.2f ns per step per SIMD = %.2f ns per pair-step\n", name, wpc, ms,
           ms * 1e6 / steps_per_simd, ms * 1e6 / steps_per_simd / pairs);
  }
  return 0;
}

int main() {
  uint32_t* d;
  CHK(hipMalloc(&d, 256 * 8 * 256 * 4));
  if (run<0>("packed v_pk (2 pairs)", 2, d)) return 1;
  if (run<1>("sdwa cmp x2 + salu count (2 pairs)", 2, d)) return 1;
  if (run<2>("cmp + salu count (1 pair)", 1, d)) return 1;
  return 0;
}
