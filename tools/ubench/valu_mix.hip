// How do full-rate (2-cycle) and half-rate (4-cycle) VALU forms share a SIMD's issue?  (gfx950; follow-up to
// valu_rate.hip, which found ONE half-rate instruction among fifteen full-rate ones making all sixteen cost 4 cycles.)
//   blocks   : runs of N full-rate instructions followed by runs of N half-rate ones, N = 1 .. 256, in ONE wave's stream
//   split    : half of the waves of a SIMD run only full-rate instructions, the other half only half-rate ones
//   waves    : the H1F15 mix at 1, 2, 4, 8 waves per SIMD
// Prints ns and (at the clock measured in-kernel with s_memtime / s_memrealtime) cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define REP2(x) x x
#define REP4(x) REP2(x) REP2(x)
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP8(x) REP8(x)
#define REP32(x) REP16(x) REP16(x)
#define REP64(x) REP32(x) REP32(x)
#define REP128(x) REP64(x) REP64(x)
#define REP256(x) REP128(x) REP128(x)
#define F4 "v_lshrrev_b32 %0, 1, %0\n\tv_xor_b32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_lshrrev_b32 %3, 1, %3\n\t"
#define H4 "v_and_or_b32 %0, %0, %4, %5\n\tv_add_u32_dpp %1, %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_bcnt_u32_b32 %2, %2, %4\n\tv_perm_b32 %3, %3, %4, %5\n\t"

// V: 0 pure F, 1 pure H, 2 F4 H4 alternating, 3 F16 H16, 4 F64 H64, 5 F256 H256, 6 split by wave parity, 7 split by wave >= half,
//    8 H1F15-like (1 H per 15 F)
template <int V>
__global__ void __launch_bounds__(256) k(uint32_t* out, unsigned long long* stamps, int iters) {
  uint32_t a = threadIdx.x, b = threadIdx.x * 3, c = threadIdx.x * 5, d = threadIdx.x * 7, e = 0x00010001u, f = 3;
  const int wave = threadIdx.x >> 6;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#define RUN(G) asm volatile(G : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f))
    if (V == 0) RUN(REP256(F4) REP256(F4));
    else if (V == 1) RUN(REP256(H4) REP256(H4));
    else if (V == 2) RUN(REP256(F4 H4));
    else if (V == 3) RUN(REP64(REP4(F4) REP4(H4)));
    else if (V == 4) RUN(REP16(REP16(F4) REP16(H4)));
    else if (V == 5) RUN(REP4(REP64(F4) REP64(H4)));
    else if (V == 6) { if ((blockIdx.x >> 8) & 1) RUN(REP256(F4) REP256(F4)); else RUN(REP256(H4) REP256(H4)); }
    else if (V == 7) { if (wave & 1) RUN(REP256(F4) REP256(F4)); else RUN(REP256(H4) REP256(H4)); }
    else if (V == 9) { if (it < iters / 2) RUN(REP256(F4) REP256(F4)); else RUN(REP256(H4) REP256(H4)); }
    else if (V == 10) { if ((it < iters / 2) == (((blockIdx.x >> 8) & 1) != 0)) RUN(REP256(F4) REP256(F4)); else RUN(REP256(H4) REP256(H4)); }
    else if (V == 11) RUN(REP128(F4 F4 F4 "s_cmp_eq_u32 0, 0\n\ts_cbranch_scc1 1f\n\tv_and_or_b32 %0, %0, %4, %5\n\t1:\n\t" F4));
    else if (V == 12) RUN(REP128(F4 F4 F4 "s_nop 0\n\t" F4));
    else if (V == 13) { if (it == 0) RUN(REP256(H4) REP256(H4)); else RUN(REP256(F4) REP256(F4)); }
    else if (V == 8) RUN(REP128("v_and_or_b32 %0, %0, %4, %5\n\t" F4 F4 F4 "v_lshrrev_b32 %0, 1, %0\n\tv_xor_b32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\t"));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int V>
int run(const char* name, uint32_t* d, unsigned long long* st, int wps /*waves per SIMD*/) {
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const int iters = 400, blocks = 256 * wps;   // 256-thread blocks: 4 waves, one per SIMD; wps blocks per CU
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, st, iters);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, st, iters);
  CHK(hipEventRecord(e1));
  CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  static unsigned long long h[2 * 256 * 8];
  CHK(hipMemcpy(h, st, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost));
  double clk = 0; for (int i = 0; i < blocks; ++i) clk += (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;  // GHz (memrealtime: 100 MHz)
  clk /= blocks;
  const double instr_per_simd = (double)wps * iters * 2048.0;   // every variant: 2048 instructions per iteration and wave
  const double ns = ms * 1e6 / instr_per_simd;
  printf("%-34s wps=%d  %.2f ms  %.3f ns = %.2f cycles per wave-instruction per SIMD at %.3f GHz (in-kernel)\n", name, wps, ms, ns, ns * clk, clk);
  return 0;
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  uint32_t* d; unsigned long long* st;
  CHK(hipMalloc(&d, 256 * 8 * 256 * 4)); CHK(hipMalloc(&st, 2 * 256 * 8 * 8));
  for (int wps : {8, 2}) {
    if (run<0>("pure full-rate", d, st, wps)) return 1;
    if (run<1>("pure half-rate", d, st, wps)) return 1;
    if (run<2>("F4 H4 alternating", d, st, wps)) return 1;
    if (run<3>("F16 H16 blocks", d, st, wps)) return 1;
    if (run<4>("F64 H64 blocks", d, st, wps)) return 1;
    if (run<5>("F256 H256 blocks", d, st, wps)) return 1;
    if (run<6>("split by workgroup layer (same SIMD)", d, st, wps)) return 1;
    if (run<7>("split by wave parity (other SIMDs)", d, st, wps)) return 1;
    if (run<8>("1 half per 15 full", d, st, wps)) return 1;
    if (run<9>("first half F, second half H", d, st, wps)) return 1;
    if (run<10>("same, layers in opposite order", d, st, wps)) return 1;
    if (run<11>("F with a skipped H every 16", d, st, wps)) return 1;
    if (run<12>("F with an s_nop every 16", d, st, wps)) return 1;
    if (run<13>("H in iteration 0 only, then F", d, st, wps)) return 1;
  }
  return 0;
}
