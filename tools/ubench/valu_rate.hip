// Issue-rate microbenchmark for the instruction classes the pair kernel uses (gfx950).
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

// each variant: 64 copies of a small instruction group on 4 independent register sets
#define G_FMA   "v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5\n\t"
#define G_ADD   "v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4\n\t"
#define G_DPP   "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define G_DPPR  "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define G_MOV   "v_mov_b32 %0, %1\n\tv_mov_b32 %1, %2\n\tv_mov_b32 %2, %3\n\tv_mov_b32 %3, %0\n\t"
#define G_CMP   "v_cmp_lt_u32_e32 vcc, %0, %4\n\tv_cmp_lt_u32_e32 vcc, %1, %4\n\tv_cmp_lt_u32_e32 vcc, %2, %4\n\tv_cmp_lt_u32_e32 vcc, %3, %4\n\t"
#define G_ADDC  "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\tv_addc_co_u32_e32 %2, vcc, 0, %2, vcc\n\tv_addc_co_u32_e32 %3, vcc, 0, %3, vcc\n\t"
#define G_PKSUB "v_pk_sub_u16 %0, %0, %4 clamp\n\tv_pk_sub_u16 %1, %1, %4 clamp\n\tv_pk_sub_u16 %2, %2, %4 clamp\n\tv_pk_sub_u16 %3, %3, %4 clamp\n\t"
#define G_PKADD "v_pk_add_u16 %0, %0, %4\n\tv_pk_add_u16 %1, %1, %4\n\tv_pk_add_u16 %2, %2, %4\n\tv_pk_add_u16 %3, %3, %4\n\t"
#define G_CNDM  "v_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_cndmask_b32 %3, %3, %4, vcc\n\t"
#define G_BCNT  "v_bcnt_u32_b32 %0, %0, %4\n\tv_bcnt_u32_b32 %1, %1, %4\n\tv_bcnt_u32_b32 %2, %2, %4\n\tv_bcnt_u32_b32 %3, %3, %4\n\t"
#define G_SUBF  "v_sub_f32_e64 %0, %4, %0 clamp\n\tv_sub_f32_e64 %1, %4, %1 clamp\n\tv_sub_f32_e64 %2, %4, %2 clamp\n\tv_sub_f32_e64 %3, %4, %3 clamp\n\t"
#define G_ADDF  "v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %4\n\t"
#define G_ADDFDPP "v_add_f32_dpp %0, %0, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %2, %2, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define G_MIX1  "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_cmp_lt_u32_e32 vcc, %0, %4\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_add_u32 %3, %3, %2\n\t"
#define G_MIXF  "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_sub_f32_e64 %1, %4, %0 clamp\n\tv_add_f32 %2, %2, %1\n\tv_mov_b32_dpp %3, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"

#define DPPR " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
#define G_ALIGN "v_alignbit_b32 %0, %0, %4, 31\n\tv_alignbit_b32 %1, %1, %4, 31\n\tv_alignbit_b32 %2, %2, %4, 31\n\tv_alignbit_b32 %3, %3, %4, 31\n\t"
#define G_SUBDPP "v_sub_u32_dpp %0, %4, %5" DPPR "v_sub_u32_dpp %1, %4, %5" DPPR "v_sub_u32_dpp %2, %4, %5" DPPR "v_sub_u32_dpp %3, %4, %5" DPPR
#define G_SUBCODPP "v_sub_co_u32_dpp %0, vcc, %4, %5" DPPR "v_sub_co_u32_dpp %1, vcc, %4, %5" DPPR "v_sub_co_u32_dpp %2, vcc, %4, %5" DPPR "v_sub_co_u32_dpp %3, vcc, %4, %5" DPPR
#define G_MIXA "v_sub_co_u32_dpp %1, vcc, %4, %5" DPPR "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\tv_sub_co_u32_dpp %1, vcc, %4, %5" DPPR "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
#define G_MIXB "v_sub_u32_dpp %1, %4, %5" DPPR "v_alignbit_b32 %0, %0, %1, 31\n\tv_sub_u32_dpp %1, %4, %5" DPPR "v_alignbit_b32 %0, %0, %1, 31\n\t"
#define G_MIXC "v_sub_u32_dpp %1, %4, %5" DPPR "v_sub_u32_dpp %2, %4, %5" DPPR "v_alignbit_b32 %0, %0, %1, 31\n\tv_alignbit_b32 %3, %3, %2, 31\n\t"

#define G_ANDOR "v_and_or_b32 %0, %0, %4, %5\n\tv_and_or_b32 %1, %1, %4, %5\n\tv_and_or_b32 %2, %2, %4, %5\n\tv_and_or_b32 %3, %3, %4, %5\n\t"
#define G_LSHR "v_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\t"
#define G_ADD3 "v_add3_u32 %0, %0, %4, %5\n\tv_add3_u32 %1, %1, %4, %5\n\tv_add3_u32 %2, %2, %4, %5\n\tv_add3_u32 %3, %3, %4, %5\n\t"
#define G_PERM "v_perm_b32 %0, %0, %4, %5\n\tv_perm_b32 %1, %1, %4, %5\n\tv_perm_b32 %2, %2, %4, %5\n\tv_perm_b32 %3, %3, %4, %5\n\t"
#define G_BFI "v_bfi_b32 %0, %0, %4, %5\n\tv_bfi_b32 %1, %1, %4, %5\n\tv_bfi_b32 %2, %2, %4, %5\n\tv_bfi_b32 %3, %3, %4, %5\n\t"
#define G_LSHLADD "v_lshl_add_u32 %0, %0, 3, %4\n\tv_lshl_add_u32 %1, %1, 3, %4\n\tv_lshl_add_u32 %2, %2, 3, %4\n\tv_lshl_add_u32 %3, %3, 3, %4\n\t"
#define G_CNDS "v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n\tv_cndmask_b32_e64 %1, %1, %4, s[20:21]\n\tv_cndmask_b32_e64 %2, %2, %4, s[20:21]\n\tv_cndmask_b32_e64 %3, %3, %4, s[20:21]\n\t"
#define G_LSHL64 "v_lshlrev_b64 v[20:21], %4, v[20:21]\n\tv_lshlrev_b64 v[22:23], %4, v[22:23]\n\t"
#define G_MUL24 "v_mul_u32_u24 %0, %0, %4\n\tv_mul_u32_u24 %1, %1, %4\n\tv_mul_u32_u24 %2, %2, %4\n\tv_mul_u32_u24 %3, %3, %4\n\t"
#define G_MULLO16 "v_mul_lo_u16 %0, %0, %4\n\tv_mul_lo_u16 %1, %1, %4\n\tv_mul_lo_u16 %2, %2, %4\n\tv_mul_lo_u16 %3, %3, %4\n\t"
#define G_PL16 "v_permlane16_swap_b32 %0, %5\n\tv_permlane16_swap_b32 %1, %5\n\tv_permlane16_swap_b32 %2, %5\n\tv_permlane16_swap_b32 %3, %5\n\t"
#define G_PL32 "v_permlane32_swap_b32 %0, %5\n\tv_permlane32_swap_b32 %1, %5\n\tv_permlane32_swap_b32 %2, %5\n\tv_permlane32_swap_b32 %3, %5\n\t"
#define G_ADDDPP "v_add_u32_dpp %0, %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_add_u32_dpp %1, %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_add_u32_dpp %2, %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_add_u32_dpp %3, %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
#define G_DOT4 "v_dot4_u32_u8 %0, %4, %5, %0\n\tv_dot4_u32_u8 %1, %4, %5, %1\n\tv_dot4_u32_u8 %2, %4, %5, %2\n\tv_dot4_u32_u8 %3, %4, %5, %3\n\t"
#define G_SAD "v_sad_u32 %0, %4, %5, %0\n\tv_sad_u32 %1, %4, %5, %1\n\tv_sad_u32 %2, %4, %5, %2\n\tv_sad_u32 %3, %4, %5, %3\n\t"
#define G_MAD24 "v_mad_u32_u24 %0, %4, %5, %0\n\tv_mad_u32_u24 %1, %4, %5, %1\n\tv_mad_u32_u24 %2, %4, %5, %2\n\tv_mad_u32_u24 %3, %4, %5, %3\n\t"
#define G_XOR "v_xor_b32 %0, %0, %4\n\tv_xor_b32 %1, %1, %4\n\tv_xor_b32 %2, %2, %4\n\tv_xor_b32 %3, %3, %4\n\t"
#define G_MIXP "v_add_u32_dpp %1, %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_lshrrev_b32 %0, 1, %0\n\tv_and_or_b32 %0, %1, %4, %0\n\tv_add_u32_dpp %3, %4, %5 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_lshrrev_b32 %2, 1, %2\n\tv_and_or_b32 %2, %3, %4, %2\n\t"
#define G_MIXQ "v_add_u32_dpp %1, %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_add_u32_dpp %3, %4, %5 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_perm_b32 %1, %1, %3, %5\n\tv_lshrrev_b32 %0, 1, %0\n\tv_and_or_b32 %0, %1, %4, %0\n\t"
#define G_HFHF "v_and_or_b32 %0, %0, %4, %5\n\tv_lshrrev_b32 %1, 1, %1\n\tv_and_or_b32 %2, %2, %4, %5\n\tv_lshrrev_b32 %3, 1, %3\n\t"
#define G_HHFF "v_and_or_b32 %0, %0, %4, %5\n\tv_and_or_b32 %2, %2, %4, %5\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %3, 1, %3\n\t"
#define G_FFDEP "v_add_u32 %0, %0, %4\n\tv_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %1, %1, %4\n\t"
#define G_HFFdep "v_and_or_b32 %0, %0, %4, %5\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %1, %1, %4\n\tv_and_or_b32 %2, %2, %4, %5\n\t"
#define G_DFF "v_add_u32_dpp %0, %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_add_u32_dpp %3, %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
#define G_BFF "v_bcnt_u32_b32 %0, %0, %4\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_bcnt_u32_b32 %3, %3, %4\n\t"
#define G_FFFH "v_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_and_or_b32 %3, %3, %4, %5\n\t"
#define G_H1F7 "v_and_or_b32 %0, %0, %4, %5\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\t"
#define G_H1F15 "v_and_or_b32 %0, %0, %4, %5\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\t"
#define G_H2F14 "v_and_or_b32 %0, %0, %4, %5\n\tv_and_or_b32 %1, %1, %4, %5\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\t"
#define G_H4F12 "v_and_or_b32 %0, %0, %4, %5\n\tv_and_or_b32 %1, %1, %4, %5\n\tv_and_or_b32 %2, %2, %4, %5\n\tv_and_or_b32 %3, %3, %4, %5\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\t"
#define G_H8F8 "v_and_or_b32 %0, %0, %4, %5\n\tv_and_or_b32 %1, %1, %4, %5\n\tv_and_or_b32 %2, %2, %4, %5\n\tv_and_or_b32 %3, %3, %4, %5\n\tv_and_or_b32 %0, %0, %4, %5\n\tv_and_or_b32 %1, %1, %4, %5\n\tv_and_or_b32 %2, %2, %4, %5\n\tv_and_or_b32 %3, %3, %4, %5\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\t"
#define G_A1F7 "v_add_u32 %0, %0, %4\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\t"
#define G_D1F15 "v_add_u32_dpp %0, %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\t"
#define G_B1F15 "v_bcnt_u32_b32 %0, %0, %4\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\t"
#define G_C1F15 "v_cmp_lt_u32_e32 vcc, %0, %4\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\t"
#define G_P1F15 "v_perm_b32 %0, %0, %4, %5\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\t"
#define G_X1F15 "v_add3_u32 %0, %0, %4, %5\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\t"
#define G_FMA1F15 "v_fma_f32 %0, %0, %4, %5\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\t"
#define G_L1F15 "v_lshl_add_u32 %0, %0, 3, %4\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\t"

template <int V>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters) {
  uint32_t a = threadIdx.x, b = threadIdx.x * 3, c = threadIdx.x * 5, d = threadIdx.x * 7, e = 0x00010001u, f = 3;
  for (int it = 0; it < iters; ++it) {
#define RUN(G) asm volatile(REP64(G) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "vcc", "s20", "s21", "s22", "s23", "s24", "v20", "v21", "v22", "v23")
    if (V == 0) RUN(G_FMA); else if (V == 1) RUN(G_ADD); else if (V == 2) RUN(G_DPP); else if (V == 3) RUN(G_DPPR);
    else if (V == 4) RUN(G_MOV); else if (V == 5) RUN(G_CMP); else if (V == 6) RUN(G_ADDC); else if (V == 7) RUN(G_PKSUB);
    else if (V == 8) RUN(G_PKADD); else if (V == 9) RUN(G_CNDM); else if (V == 10) RUN(G_BCNT); else if (V == 11) RUN(G_SUBF);
    else if (V == 12) RUN(G_ADDF); else if (V == 13) RUN(G_ADDFDPP); else if (V == 14) RUN(G_MIX1); else if (V == 15) RUN(G_MIXF);
    else if (V == 16) RUN(G_ALIGN); else if (V == 17) RUN(G_SUBDPP); else if (V == 18) RUN(G_SUBCODPP);
    else if (V == 19) RUN(G_MIXA); else if (V == 20) RUN(G_MIXB); else if (V == 21) RUN(G_MIXC);
    else if (V == 100) RUN(G_D1F15);
    else if (V == 101) RUN(G_B1F15);
    else if (V == 102) RUN(G_C1F15);
    else if (V == 103) RUN(G_P1F15);
    else if (V == 104) RUN(G_X1F15);
    else if (V == 105) RUN(G_FMA1F15);
    else if (V == 106) RUN(G_L1F15);
    else if (V == 80) RUN(G_H1F7);
    else if (V == 81) RUN(G_H1F15);
    else if (V == 82) RUN(G_H2F14);
    else if (V == 83) RUN(G_H4F12);
    else if (V == 84) RUN(G_H8F8);
    else if (V == 85) RUN(G_A1F7);
    else if (V == 60) RUN(G_HFHF);
    else if (V == 61) RUN(G_HHFF);
    else if (V == 62) RUN(G_FFDEP);
    else if (V == 63) RUN(G_HFFdep);
    else if (V == 64) RUN(G_DFF);
    else if (V == 65) RUN(G_BFF);
    else if (V == 66) RUN(G_FFFH);
    else if (V == 22) RUN(G_ANDOR);
    else if (V == 23) RUN(G_LSHR);
    else if (V == 24) RUN(G_ADD3);
    else if (V == 25) RUN(G_PERM);
    else if (V == 26) RUN(G_BFI);
    else if (V == 27) RUN(G_LSHLADD);
    else if (V == 28) RUN(G_CNDS);
    else if (V == 29) RUN(G_LSHL64);
    else if (V == 30) RUN(G_MUL24);
    else if (V == 31) RUN(G_MULLO16);
    else if (V == 32) RUN(G_PL16);
    else if (V == 33) RUN(G_PL32);
    else if (V == 34) RUN(G_ADDDPP);
    else if (V == 35) RUN(G_DOT4);
    else if (V == 36) RUN(G_SAD);
    else if (V == 37) RUN(G_MAD24);
    else if (V == 38) RUN(G_XOR);
    else if (V == 39) RUN(G_MIXP);
    else if (V == 40) RUN(G_MIXQ);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}

template <int V>
int run(const char* name, uint32_t* d) {
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const int iters = 4000, wpc = 32, blocks = 256 * wpc / 4;
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, iters);  // warm clocks
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, iters);
  CHK(hipEventRecord(e1));
  CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  const double instr_per_simd = (double)wpc / 4.0 * iters * 256.0;
  printf("%-22s %.2f ms  %.3f ns per wave-instruction per SIMD\n", name, ms, ms * 1e6 / instr_per_simd);
  return 0;
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  uint32_t* d;
  CHK(hipMalloc(&d, 256 * 8 * 256 * 4));
  if (run<0>("v_fma_f32", d)) return 1;
  if (run<100>("D1F15 (per 4 of 16)", d)) return 1;
  if (run<101>("B1F15 (per 4 of 16)", d)) return 1;
  if (run<102>("C1F15 (per 4 of 16)", d)) return 1;
  if (run<103>("P1F15 (per 4 of 16)", d)) return 1;
  if (run<104>("X1F15 (per 4 of 16)", d)) return 1;
  if (run<105>("FMA1F15 (per 4 of 16)", d)) return 1;
  if (run<106>("L1F15 (per 4 of 16)", d)) return 1;

  if (run<80>("H1F7 (per 4)", d)) return 1;
  if (run<81>("H1F15 (per 4)", d)) return 1;
  if (run<82>("H2F14 (per 4)", d)) return 1;
  if (run<83>("H4F12 (per 4)", d)) return 1;
  if (run<84>("H8F8 (per 4)", d)) return 1;
  if (run<85>("A1F7 (per 4)", d)) return 1;

  if (run<1>("v_add_u32", d)) return 1;
  if (run<2>("v_mov_dpp wave_shr", d)) return 1;
  if (run<3>("v_mov_dpp row_shr", d)) return 1;
  if (run<4>("v_mov_b32", d)) return 1;
  if (run<5>("v_cmp_lt_u32", d)) return 1;
  if (run<6>("v_addc_co_u32", d)) return 1;
  if (run<7>("v_pk_sub_u16 clamp", d)) return 1;
  if (run<8>("v_pk_add_u16", d)) return 1;
  if (run<9>("v_cndmask_b32", d)) return 1;
  if (run<10>("v_bcnt_u32_b32", d)) return 1;
  if (run<11>("v_sub_f32 clamp", d)) return 1;
  if (run<12>("v_add_f32", d)) return 1;
  if (run<13>("v_add_f32_dpp", d)) return 1;
  if (run<14>("mix dpp/cmp/cndm/add", d)) return 1;
  if (run<15>("mix dpp/subf/addf/dpp", d)) return 1;
  if (run<16>("v_alignbit_b32", d)) return 1;
  if (run<17>("v_sub_u32_dpp row_shr", d)) return 1;
  if (run<18>("v_sub_co_u32_dpp", d)) return 1;
  if (run<19>("mix sub_co_dpp/addc", d)) return 1;
  if (run<20>("mix sub_dpp/alignbit", d)) return 1;
  if (run<21>("mix 2sub_dpp/2alignbit", d)) return 1;
  if (run<22>("andor", d)) return 1;
  if (run<23>("lshr", d)) return 1;
  if (run<24>("add3", d)) return 1;
  if (run<25>("perm", d)) return 1;
  if (run<26>("bfi", d)) return 1;
  if (run<27>("lshladd", d)) return 1;
  if (run<28>("cnds", d)) return 1;
  if (run<29>("lshl64", d)) return 1;
  if (run<30>("mul24", d)) return 1;
  if (run<31>("mullo16", d)) return 1;
  if (run<32>("pl16", d)) return 1;
  if (run<33>("pl32", d)) return 1;
  if (run<34>("adddpp", d)) return 1;
  if (run<35>("dot4", d)) return 1;
  if (run<36>("sad", d)) return 1;
  if (run<37>("mad24", d)) return 1;
  if (run<38>("xor", d)) return 1;
  if (run<39>("mixp", d)) return 1;
  if (run<40>("mixq", d)) return 1;
  if (run<60>("HFHF", d)) return 1;
  if (run<61>("HHFF", d)) return 1;
  if (run<62>("FFDEP", d)) return 1;
  if (run<63>("HFFdep", d)) return 1;
  if (run<64>("DFF", d)) return 1;
  if (run<65>("BFF", d)) return 1;
  if (run<66>("FFFH", d)) return 1;
  if (run<0>("v_fma_f32 (again)", d)) return 1;
  return 0;
}
