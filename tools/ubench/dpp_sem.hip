// Semantics probe: carry-out of v_sub_co / v_subrev_co with a DPP source, and bound_ctrl behaviour (gfx950).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void k(uint32_t* out) {
  const uint32_t lane = threadIdx.x & 63;
  uint32_t a = lane * 10u;          // src0 (DPP side): lane l sees a[l-1] = 10(l-1)
  uint32_t b = 95u;                 // src1
  uint32_t d, r1 = 0, r2 = 0, r3 = 0, r4 = 0;
  asm volatile("s_nop 1\n\tv_sub_co_u32_dpp %0, vcc, %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
               "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
               "v_subrev_co_u32_dpp %0, vcc, %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
               "v_addc_co_u32_e32 %2, vcc, 0, %2, vcc\n\t"
               "v_sub_co_u32_e32 %0, vcc, %5, %6\n\t"
               "v_addc_co_u32_e32 %3, vcc, 0, %3, vcc\n\t"
               "v_subrev_co_u32_e32 %0, vcc, %5, %6\n\t"
               "v_addc_co_u32_e32 %4, vcc, 0, %4, vcc\n\t"
               : "=&v"(d), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4) : "v"(a), "v"(b) : "vcc");
  out[lane] = r1; out[64 + lane] = r2; out[128 + lane] = r3; out[192 + lane] = r4;
}
int main() {
  uint32_t* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  uint32_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("lane: a[l-1] | sub_dpp subrev_dpp | a[l]: sub subrev   (b = 95)\n");
  for (int l = 0; l < 20; ++l) printf("%2d: %4d | %u %u | %4d: %u %u\n", l, l % 16 ? (l - 1) * 10 : -1, h[l], h[64 + l], l * 10, h[128 + l], h[192 + l]);
  return 0;
}
