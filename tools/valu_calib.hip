// VALU issue-cost calibration for the roofline block: independent instruction streams of ONE
// opcode each, at 1 / 4 / 8 waves per SIMD, on every SIMD of the chip.  Round 2 calibrated v_fma_f32 only and charged every VALU
// instruction its 2 issue cycles; the megakernel's hot arithmetic is glibc's double-core libm (v_fma_f64 ...), IEEE divides
// (v_div_scale / v_rcp / v_div_fmas / v_div_fixup) and transcendentals, which are dearer.  This prints, per opcode, the time per
// wave-level instruction relative to v_fma_f32 in the same run -- "issue cycles" = 2 x that ratio, with v_fma_f32 = 2 cycles (0.5 per
// cycle per SIMD-32: MI355X_MICROARCH.md constants table; 157.3 TFLOP/s f32 = 1024 SIMDs x 32 lanes x 2 x 2.4 GHz) -- and writes
// profiles/valu_calib.json for tools/pmc_traffic.py, which weights the SQ_INSTS_VALU_* class counters with these costs.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_calib.hip -o /tmp/valu_calib && /tmp/valu_calib [out.json]
// (`rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES -- /tmp/valu_calib` gives the same costs in counted cycles:
//  GRBM_GUI_ACTIVE / 8 / (instructions per wave x waves per SIMD); tools/profile_round.sh keeps that CSV next to the JSON.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <algorithm>

#define PER_ITER 64   /* instructions of the opcode per loop trip (8 registers x 8) */

// eight independent registers, each the destination (and, where the opcode has one, a source) of every eighth instruction
#define STREAM8(ASM, CON) \
	asm volatile(ASM : "+v"(x0) : CON); asm volatile(ASM : "+v"(x1) : CON); asm volatile(ASM : "+v"(x2) : CON); asm volatile(ASM : "+v"(x3) : CON); \
	asm volatile(ASM : "+v"(x4) : CON); asm volatile(ASM : "+v"(x5) : CON); asm volatile(ASM : "+v"(x6) : CON); asm volatile(ASM : "+v"(x7) : CON);
#define STREAM8_VCC(ASM, CON) \
	asm volatile(ASM : "+v"(x0) : CON : "vcc"); asm volatile(ASM : "+v"(x1) : CON : "vcc"); asm volatile(ASM : "+v"(x2) : CON : "vcc"); asm volatile(ASM : "+v"(x3) : CON : "vcc"); \
	asm volatile(ASM : "+v"(x4) : CON : "vcc"); asm volatile(ASM : "+v"(x5) : CON : "vcc"); asm volatile(ASM : "+v"(x6) : CON : "vcc"); asm volatile(ASM : "+v"(x7) : CON : "vcc");
#define COMMA ,
#define KERNEL(NAME, T, TA, BODY) \
__global__ void __launch_bounds__(256) k_##NAME(T* out, int iters, TA a, TA b) \
{ \
	T x0 = (T)(threadIdx.x + 1), x1 = x0 + (T)1, x2 = x0 + (T)2, x3 = x0 + (T)3, x4 = x0 + (T)4, x5 = x0 + (T)5, x6 = x0 + (T)6, x7 = x0 + (T)7; \
	for (int i = 0; i < iters; ++i) { _Pragma("unroll") for (int k = 0; k < PER_ITER / 8; ++k) { BODY } } \
	out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7; \
}

KERNEL(fma_f32, float, float, STREAM8("v_fma_f32 %0, %0, %1, %2", "v"(a) COMMA "v"(b)))
KERNEL(mul_f32, float, float, STREAM8("v_mul_f32 %0, %0, %1", "v"(a)))
KERNEL(add_f32, float, float, STREAM8("v_add_f32 %0, %0, %1", "v"(a)))
KERNEL(pk_fma_f32, double, double, STREAM8("v_pk_fma_f32 %0, %0, %1, %2", "v"(a) COMMA "v"(b)))
KERNEL(fma_f64, double, double, STREAM8("v_fma_f64 %0, %0, %1, %2", "v"(a) COMMA "v"(b)))
KERNEL(mul_f64, double, double, STREAM8("v_mul_f64 %0, %0, %1", "v"(a)))
KERNEL(add_f64, double, double, STREAM8("v_add_f64 %0, %0, %1", "v"(a)))
KERNEL(rcp_f32, float, float, STREAM8("v_rcp_f32 %0, %0", "v"(a)))
KERNEL(sqrt_f32, float, float, STREAM8("v_sqrt_f32 %0, %0", "v"(a)))
KERNEL(rsq_f32, float, float, STREAM8("v_rsq_f32 %0, %0", "v"(a)))
KERNEL(exp_f32, float, float, STREAM8("v_exp_f32 %0, %0", "v"(a)))
KERNEL(log_f32, float, float, STREAM8("v_log_f32 %0, %0", "v"(a)))
KERNEL(rcp_f64, double, double, STREAM8("v_rcp_f64 %0, %0", "v"(a)))
KERNEL(sqrt_f64, double, double, STREAM8("v_sqrt_f64 %0, %0", "v"(a)))
KERNEL(div_scale_f32, float, float, STREAM8_VCC("v_div_scale_f32 %0, vcc, %0, %1, %0", "v"(a)))
KERNEL(div_fmas_f32, float, float, STREAM8_VCC("v_div_fmas_f32 %0, %0, %1, %2", "v"(a) COMMA "v"(b)))
KERNEL(div_fixup_f32, float, float, STREAM8("v_div_fixup_f32 %0, %0, %1, %2", "v"(a) COMMA "v"(b)))
KERNEL(div_scale_f64, double, double, STREAM8_VCC("v_div_scale_f64 %0, vcc, %0, %1, %0", "v"(a)))
KERNEL(div_fmas_f64, double, double, STREAM8_VCC("v_div_fmas_f64 %0, %0, %1, %2", "v"(a) COMMA "v"(b)))
KERNEL(div_fixup_f64, double, double, STREAM8("v_div_fixup_f64 %0, %0, %1, %2", "v"(a) COMMA "v"(b)))
KERNEL(add_u32, unsigned, unsigned, STREAM8("v_add_u32 %0, %0, %1", "v"(a)))
KERNEL(and_b32, unsigned, unsigned, STREAM8("v_and_b32 %0, %0, %1", "v"(a)))
KERNEL(cndmask_b32, unsigned, unsigned, STREAM8_VCC("v_cndmask_b32 %0, %0, %1, vcc", "v"(a)))
KERNEL(mul_lo_u32, unsigned, unsigned, STREAM8("v_mul_lo_u32 %0, %0, %1", "v"(a)))
KERNEL(mul_hi_u32, unsigned, unsigned, STREAM8("v_mul_hi_u32 %0, %0, %1", "v"(a)))
KERNEL(mad_u64_u32, unsigned long long, unsigned, STREAM8_VCC("v_mad_u64_u32 %0, vcc, %1, %2, %0", "v"(a) COMMA "v"(b)))
KERNEL(lshlrev_b64, unsigned long long, unsigned, STREAM8("v_lshlrev_b64 %0, 1, %0", "v"(a)))
KERNEL(lshrrev_b32, unsigned, unsigned, STREAM8("v_lshrrev_b32 %0, 1, %0", "v"(a)))
KERNEL(cvt_f32_u32, unsigned, unsigned, STREAM8("v_cvt_f32_u32 %0, %0", "v"(a)))
KERNEL(cvt_f32_ubyte0, unsigned, unsigned, STREAM8("v_cvt_f32_ubyte0 %0, %0", "v"(a)))
KERNEL(cvt_f64_f32, double, float, STREAM8("v_cvt_f64_f32 %0, %1", "v"(a)))
KERNEL(cvt_f32_f64, float, double, STREAM8("v_cvt_f32_f64 %0, %1", "v"(a)))
KERNEL(cvt_i32_f64, int, double, STREAM8("v_cvt_i32_f64 %0, %1", "v"(a)))
KERNEL(max_f32, float, float, STREAM8("v_max_f32 %0, %0, %1", "v"(a)))
KERNEL(mov_dpp, unsigned, unsigned, STREAM8("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v"(a)))
KERNEL(cmp_f32, unsigned, float, STREAM8_VCC("v_cmp_lt_f32 vcc, %1, %2\n\tv_nop", "v"(a) COMMA "v"(b)))   /* two instructions per count: halved below */
// round 3: the rest of what the megakernels' disassembly holds in numbers
#define STREAM8_RO(ASM, CON) STREAM8(ASM, CON)
#define STREAM8_S20(ASM, CON) \
	asm volatile(ASM : "+v"(x0) : CON : "s20", "s21"); asm volatile(ASM : "+v"(x1) : CON : "s20", "s21"); asm volatile(ASM : "+v"(x2) : CON : "s20", "s21"); asm volatile(ASM : "+v"(x3) : CON : "s20", "s21"); \
	asm volatile(ASM : "+v"(x4) : CON : "s20", "s21"); asm volatile(ASM : "+v"(x5) : CON : "s20", "s21"); asm volatile(ASM : "+v"(x6) : CON : "s20", "s21"); asm volatile(ASM : "+v"(x7) : CON : "s20", "s21");
KERNEL(cndmask_vcc, unsigned, unsigned, STREAM8_RO("v_cndmask_b32 %0, %0, %1, vcc", "v"(a)))   /* reads vcc only */
KERNEL(sub_f32, float, float, STREAM8("v_sub_f32 %0, %0, %1", "v"(a)))
KERNEL(fmac_f32, float, float, STREAM8("v_fmac_f32 %0, %1, %2", "v"(a) COMMA "v"(b)))
KERNEL(add_f32_mod, float, float, STREAM8("v_add_f32_e64 %0, |%0|, -%1", "v"(a)))
KERNEL(mul_f32_sgpr, float, float, STREAM8("v_mul_f32 %0, %1, %0", "s"(a)))
KERNEL(min_f32, float, float, STREAM8("v_min_f32 %0, %0, %1", "v"(a)))
KERNEL(max3_f32, float, float, STREAM8("v_max3_f32 %0, %0, %1, %2", "v"(a) COMMA "v"(b)))
KERNEL(min3_f32, float, float, STREAM8("v_min3_f32 %0, %0, %1, %2", "v"(a) COMMA "v"(b)))
KERNEL(med3_f32, float, float, STREAM8("v_med3_f32 %0, %0, %1, %2", "v"(a) COMMA "v"(b)))
KERNEL(mov_b32, unsigned, unsigned, STREAM8("v_mov_b32 %0, %1", "v"(a)))
KERNEL(xor_b32, unsigned, unsigned, STREAM8("v_xor_b32 %0, %0, %1", "v"(a)))
KERNEL(or_b32, unsigned, unsigned, STREAM8("v_or_b32 %0, %0, %1", "v"(a)))
KERNEL(lshlrev_b32, unsigned, unsigned, STREAM8("v_lshlrev_b32 %0, 1, %0", "v"(a)))
KERNEL(sub_u32, unsigned, unsigned, STREAM8("v_sub_u32 %0, %0, %1", "v"(a)))
KERNEL(lshl_add_u32, unsigned, unsigned, STREAM8("v_lshl_add_u32 %0, %0, 2, %1", "v"(a)))
KERNEL(add3_u32, unsigned, unsigned, STREAM8("v_add3_u32 %0, %0, %1, %2", "v"(a) COMMA "v"(b)))
KERNEL(and_or_b32, unsigned, unsigned, STREAM8("v_and_or_b32 %0, %0, %1, %2", "v"(a) COMMA "v"(b)))
KERNEL(bfe_u32, unsigned, unsigned, STREAM8("v_bfe_u32 %0, %0, 4, 8", "v"(a)))
KERNEL(bfi_b32, unsigned, unsigned, STREAM8("v_bfi_b32 %0, %1, %0, %2", "v"(a) COMMA "v"(b)))
KERNEL(perm_b32, unsigned, unsigned, STREAM8("v_perm_b32 %0, %0, %1, %2", "v"(a) COMMA "v"(b)))
KERNEL(alignbit_b32, unsigned, unsigned, STREAM8("v_alignbit_b32 %0, %0, %1, 7", "v"(a)))
KERNEL(pk_fma_f16, unsigned, unsigned, STREAM8("v_pk_fma_f16 %0, %0, %1, %2", "v"(a) COMMA "v"(b)))
KERNEL(pk_max_f16, unsigned, unsigned, STREAM8("v_pk_max_f16 %0, %0, %1", "v"(a)))
KERNEL(pk_min_f16, unsigned, unsigned, STREAM8("v_pk_min_f16 %0, %0, %1", "v"(a)))
KERNEL(pk_maximum3_f16, unsigned, unsigned, STREAM8("v_pk_maximum3_f16 %0, %0, %1, %2", "v"(a) COMMA "v"(b)))
KERNEL(pk_add_f16, unsigned, unsigned, STREAM8("v_pk_add_f16 %0, %0, %1", "v"(a)))
KERNEL(pk_mul_f16, unsigned, unsigned, STREAM8("v_pk_mul_f16 %0, %0, %1", "v"(a)))
KERNEL(pk_max_i16, unsigned, unsigned, STREAM8("v_pk_max_i16 %0, %0, %1", "v"(a)))
KERNEL(pk_mad_i16, unsigned, unsigned, STREAM8("v_pk_mad_i16 %0, %0, %1, %2", "v"(a) COMMA "v"(b)))
KERNEL(add_f32_sdwa, float, float, STREAM8("v_add_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1", "v"(a)))
KERNEL(cvt_f32_u32_sdwa, unsigned, unsigned, STREAM8("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2", "v"(a)))
KERNEL(cvt_pk_f32_fp8, double, unsigned, STREAM8("v_cvt_pk_f32_fp8 %0, %1", "v"(a)))
KERNEL(mul_u32_u24, unsigned, unsigned, STREAM8("v_mul_u32_u24 %0, %0, %1", "v"(a)))
KERNEL(mad_u32_u24, unsigned, unsigned, STREAM8("v_mad_u32_u24 %0, %0, %1, %2", "v"(a) COMMA "v"(b)))
KERNEL(add_co_u32, unsigned, unsigned, STREAM8_VCC("v_add_co_u32 %0, vcc, %0, %1", "v"(a)))
KERNEL(addc_co_u32, unsigned, unsigned, STREAM8_VCC("v_addc_co_u32 %0, vcc, %0, %1, vcc", "v"(a)))
KERNEL(min_u32, unsigned, unsigned, STREAM8("v_min_u32 %0, %0, %1", "v"(a)))
KERNEL(cvt_u32_f32, unsigned, unsigned, STREAM8("v_cvt_u32_f32 %0, %0", "v"(a)))
KERNEL(cvt_f32_i32, unsigned, unsigned, STREAM8("v_cvt_f32_i32 %0, %0", "v"(a)))
KERNEL(floor_f32, float, float, STREAM8("v_floor_f32 %0, %0", "v"(a)))
KERNEL(fract_f32, float, float, STREAM8("v_fract_f32 %0, %0", "v"(a)))
KERNEL(rndne_f32, float, float, STREAM8("v_rndne_f32 %0, %0", "v"(a)))
KERNEL(ldexp_f32, float, int, STREAM8("v_ldexp_f32 %0, %0, %1", "v"(a)))
KERNEL(frexp_mant_f32, float, float, STREAM8("v_frexp_mant_f32 %0, %0", "v"(a)))
KERNEL(sin_f32, float, float, STREAM8("v_sin_f32 %0, %0", "v"(a)))
KERNEL(ldexp_f64, double, int, STREAM8("v_ldexp_f64 %0, %0, %1", "v"(a)))
KERNEL(rndne_f64, double, double, STREAM8("v_rndne_f64 %0, %0", "v"(a)))
KERNEL(cvt_f64_i32, double, int, STREAM8("v_cvt_f64_i32 %0, %1", "v"(a)))
KERNEL(pk_mul_f32, double, double, STREAM8("v_pk_mul_f32 %0, %0, %1", "v"(a)))
KERNEL(pk_add_f32, double, double, STREAM8("v_pk_add_f32 %0, %0, %1", "v"(a)))
KERNEL(lshrrev_b64, unsigned long long, unsigned, STREAM8("v_lshrrev_b64 %0, 1, %0", "v"(a)))
KERNEL(cmp_only_f32, unsigned, float, STREAM8_VCC("v_cmp_lt_f32 vcc, %1, %2", "v"(a) COMMA "v"(b)))
KERNEL(cmp_sgpr_f32, unsigned, float, STREAM8_S20("v_cmp_lt_f32 s[20:21], %1, %2", "v"(a) COMMA "v"(b)))
KERNEL(cmp_u32, unsigned, unsigned, STREAM8_VCC("v_cmp_lt_u32 vcc, %1, %2", "v"(a) COMMA "v"(b)))
KERNEL(cmp_class_f32, unsigned, float, STREAM8_VCC("v_cmp_class_f32 vcc, %1, %2", "v"(a) COMMA "v"(b)))
KERNEL(readlane, unsigned, unsigned, STREAM8_S20("v_readlane_b32 s20, %0, 3", "v"(a)))
KERNEL(readfirstlane, unsigned, unsigned, STREAM8_S20("v_readfirstlane_b32 s20, %0", "v"(a)))
// operand forms: an SGPR, an inline constant, a 32-bit literal as a source of a "fast" opcode
KERNEL(fma_f32_sgpr, float, float, STREAM8("v_fma_f32 %0, %0, %1, %0", "s"(a)))
KERNEL(add_f32_sgpr, float, float, STREAM8("v_add_f32 %0, %1, %0", "s"(a)))
KERNEL(add_u32_sgpr, unsigned, unsigned, STREAM8("v_add_u32 %0, %1, %0", "s"(a)))
KERNEL(and_b32_sgpr, unsigned, unsigned, STREAM8("v_and_b32 %0, %1, %0", "s"(a)))
KERNEL(mov_b32_sgpr, unsigned, unsigned, STREAM8("v_mov_b32 %0, %1", "s"(a)))
KERNEL(mul_f32_inline, float, float, STREAM8("v_mul_f32 %0, 2.0, %0", "v"(a)))
KERNEL(mul_f32_literal, float, float, STREAM8("v_mul_f32 %0, 0x3f800054, %0", "v"(a)))
KERNEL(add_u32_inline, unsigned, unsigned, STREAM8("v_add_u32 %0, 7, %0", "v"(a)))
KERNEL(add_u32_literal, unsigned, unsigned, STREAM8("v_add_u32 %0, 0x12345, %0", "v"(a)))
KERNEL(and_b32_literal, unsigned, unsigned, STREAM8("v_and_b32 %0, 0xffffffe0, %0", "v"(a)))
KERNEL(fmaak_f32, float, float, STREAM8("v_fmaak_f32 %0, %0, %1, 0x3f800054", "v"(a)))
KERNEL(mul_f32_e64, float, float, STREAM8("v_mul_f32_e64 %0, %0, %1", "v"(a)))
KERNEL(ashrrev_i32, unsigned, unsigned, STREAM8("v_ashrrev_i32 %0, 1, %0", "v"(a)))
KERNEL(lshlrev_b32_v, unsigned, unsigned, STREAM8("v_lshlrev_b32 %0, %1, %0", "v"(a)))
KERNEL(lshrrev_b32_v, unsigned, unsigned, STREAM8("v_lshrrev_b32 %0, %1, %0", "v"(a)))
KERNEL(not_b32, unsigned, unsigned, STREAM8("v_not_b32 %0, %0", "v"(a)))
KERNEL(mbcnt_lo, unsigned, unsigned, STREAM8("v_mbcnt_lo_u32_b32 %0, %1, %0", "v"(a)))
KERNEL(mbcnt_hi, unsigned, unsigned, STREAM8("v_mbcnt_hi_u32_b32 %0, %1, %0", "v"(a)))
KERNEL(or3_b32, unsigned, unsigned, STREAM8("v_or3_b32 %0, %0, %1, %2", "v"(a) COMMA "v"(b)))
KERNEL(xad_u32, unsigned, unsigned, STREAM8("v_xad_u32 %0, %0, %1, %2", "v"(a) COMMA "v"(b)))
KERNEL(add_lshl_u32, unsigned, unsigned, STREAM8("v_add_lshl_u32 %0, %0, %1, 2", "v"(a)))
// a compare and the select that reads its mask (two instructions per count)
KERNEL(cmp_cndmask, float, float, STREAM8_VCC("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc", "v"(a) COMMA "v"(b)))
KERNEL(cmp_cndmask_e64, float, float, STREAM8_S20("v_cmp_lt_f32 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]", "v"(a) COMMA "v"(b)))
KERNEL(cndmask_e64, unsigned, unsigned, STREAM8_S20("v_cndmask_b32_e64 %0, %0, %1, s[20:21]", "v"(a)))
KERNEL(cmp_x2_cndmask, float, float, STREAM8_VCC("v_cmp_lt_f32 vcc, %0, %1\n\tv_fma_f32 %0, %0, %1, %2\n\tv_cndmask_b32 %0, %0, %2, vcc", "v"(a) COMMA "v"(b)))
KERNEL(writelane, unsigned, unsigned, STREAM8("v_writelane_b32 %0, s20, 3", "v"(a)))
KERNEL(cmpx_f32, unsigned, float, STREAM8("v_cmpx_le_f32 exec, %1, %1", "v"(a)))
KERNEL(bpermute, unsigned, unsigned, STREAM8("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)", "v"(a)))
// round 5: what a box plane could enter an f32 fma through without v_cvt_f32_ubyte, and gfx950's three-operand bit / min / max forms
KERNEL(fma_mix_f32_lo, float, unsigned, STREAM8("v_fma_mix_f32 %0, %1, %0, %0 op_sel_hi:[1,0,0]", "v"(a)))                          /* src0 = low f16 half of a dword */
KERNEL(fma_mix_f32_hi, float, unsigned, STREAM8("v_fma_mix_f32 %0, %1, %0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]", "v"(a)))           /* src0 = high f16 half */
KERNEL(fma_mix_f32_f32, float, float, STREAM8("v_fma_mix_f32 %0, %0, %1, %2", "v"(a) COMMA "v"(b)))                                   /* all three sources f32 */
KERNEL(bitop3_b32, unsigned, unsigned, STREAM8("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96", "v"(a) COMMA "v"(b)))
KERNEL(maximum3_f32, float, float, STREAM8("v_maximum3_f32 %0, %0, %1, %2", "v"(a) COMMA "v"(b)))
KERNEL(minimum3_f32, float, float, STREAM8("v_minimum3_f32 %0, %0, %1, %2", "v"(a) COMMA "v"(b)))
KERNEL(cvt_f32_f16, unsigned, unsigned, STREAM8("v_cvt_f32_f16 %0, %1", "v"(a)))
KERNEL(cvt_f32_f16_sdwa, unsigned, unsigned, STREAM8("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1", "v"(a)))
KERNEL(dot2_f32_f16, float, unsigned, STREAM8("v_dot2_f32_f16 %0, %1, %2, %0", "v"(a) COMMA "v"(b)))
KERNEL(mul_f32_sdwa_byte, float, unsigned, STREAM8("v_mul_f32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD", "v"(a)))
KERNEL(lshl_or_b32, unsigned, unsigned, STREAM8("v_lshl_or_b32 %0, %0, 8, %1", "v"(a)))
KERNEL(fmac_f32_dpp, float, float, STREAM8("v_fmac_f32_dpp %0, %1, %2 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf", "v"(a) COMMA "v"(b)))
KERNEL(cvt_f32_ubyte3, unsigned, unsigned, STREAM8("v_cvt_f32_ubyte3 %0, %1", "v"(a)))
KERNEL(ffbh_u32, unsigned, unsigned, STREAM8("v_ffbh_u32 %0, %0", "v"(a)))
KERNEL(bcnt_u32, unsigned, unsigned, STREAM8("v_bcnt_u32_b32 %0, %0, %1", "v"(a)))

struct Case { const char* name; const char* counterClass; double perCount; double ms[3]; };

template <typename T, typename TA>
static double Run(void (*k)(T*, int, TA, TA), int blocks, int iters, TA a, TA b)
{
	T* out; hipMalloc(&out, (size_t)blocks * 256 * sizeof(T));
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 50, a, b);
	double best = 1e30;
	for (int rep = 0; rep < 3; ++rep) {
		hipEventRecord(e0);
		hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, a, b);
		hipEventRecord(e1); hipDeviceSynchronize();
		float ms; hipEventElapsedTime(&ms, e0, e1);
		best = std::min(best, (double)ms);
	}
	hipFree(out); hipEventDestroy(e0); hipEventDestroy(e1);
	return best;
}

int main(int argc, char** argv)
{
	hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount, iters = 4000;
	printf("device %s, %d CUs, clockRate %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
	const int wps[3] = { 1, 4, 8 };
	std::vector<Case> cases;
	#define CASE(NAME, CLS, PER, T, TA, A, B) { Case c; c.name = #NAME; c.counterClass = CLS; c.perCount = PER; \
		for (int w = 0; w < 3; ++w) c.ms[w] = Run<T, TA>(k_##NAME, cus * wps[w], iters, (TA)(A), (TA)(B)); cases.push_back(c); }
	// counterClass: the SQ_INSTS_VALU_* counter of rocprofv3 the opcode is counted under (tools/pmc_traffic.py); "" = only in SQ_INSTS_VALU
	CASE(fma_f32, "FMA_F32", 1, float, float, 1.0001f, 0.5f)
	CASE(mul_f32, "MUL_F32", 1, float, float, 1.0001f, 0)
	CASE(add_f32, "ADD_F32", 1, float, float, 0.5f, 0)
	CASE(pk_fma_f32, "", 1, double, double, 1.0, 0.5)
	CASE(fma_f64, "FMA_F64", 1, double, double, 1.0001, 0.5)
	CASE(mul_f64, "MUL_F64", 1, double, double, 1.0001, 0)
	CASE(add_f64, "ADD_F64", 1, double, double, 0.5, 0)
	CASE(rcp_f32, "TRANS_F32", 1, float, float, 0, 0)
	CASE(sqrt_f32, "TRANS_F32", 1, float, float, 0, 0)
	CASE(rsq_f32, "TRANS_F32", 1, float, float, 0, 0)
	CASE(exp_f32, "TRANS_F32", 1, float, float, 0, 0)
	CASE(log_f32, "TRANS_F32", 1, float, float, 0, 0)
	CASE(rcp_f64, "TRANS_F64", 1, double, double, 0, 0)
	CASE(sqrt_f64, "TRANS_F64", 1, double, double, 0, 0)
	CASE(div_scale_f32, "", 1, float, float, 3.0f, 0)
	CASE(div_fmas_f32, "", 1, float, float, 1.0001f, 0.5f)
	CASE(div_fixup_f32, "", 1, float, float, 3.0f, 2.0f)
	CASE(div_scale_f64, "", 1, double, double, 3.0, 0)
	CASE(div_fmas_f64, "", 1, double, double, 1.0001, 0.5)
	CASE(div_fixup_f64, "", 1, double, double, 3.0, 2.0)
	CASE(add_u32, "INT32", 1, unsigned, unsigned, 3u, 0)
	CASE(and_b32, "INT32", 1, unsigned, unsigned, 0xffffffu, 0)
	CASE(cndmask_b32, "", 1, unsigned, unsigned, 3u, 0)
	CASE(mul_lo_u32, "INT32", 1, unsigned, unsigned, 3u, 0)
	CASE(mul_hi_u32, "INT32", 1, unsigned, unsigned, 3u, 0)
	CASE(mad_u64_u32, "INT64", 1, unsigned long long, unsigned, 3u, 5u)
	CASE(lshlrev_b64, "INT64", 1, unsigned long long, unsigned, 0, 0)
	CASE(lshrrev_b32, "INT32", 1, unsigned, unsigned, 0, 0)
	CASE(cvt_f32_u32, "CVT", 1, unsigned, unsigned, 0, 0)
	CASE(cvt_f32_ubyte0, "CVT", 1, unsigned, unsigned, 0, 0)
	CASE(cvt_f64_f32, "CVT", 1, double, float, 1.5f, 0)
	CASE(cvt_f32_f64, "CVT", 1, float, double, 1.5, 0)
	CASE(cvt_i32_f64, "CVT", 1, int, double, 1.5, 0)
	CASE(max_f32, "", 1, float, float, 0.5f, 0)
	CASE(mov_dpp, "", 1, unsigned, unsigned, 0, 0)
	CASE(cmp_f32, "", 2, unsigned, float, 0.5f, 1.5f)
	CASE(cndmask_vcc, "", 1, unsigned, unsigned, 3u, 0)
	CASE(sub_f32, "ADD_F32", 1, float, float, 0.5f, 0)
	CASE(fmac_f32, "FMA_F32", 1, float, float, 1.0001f, 0.5f)
	CASE(add_f32_mod, "ADD_F32", 1, float, float, 0.5f, 0)
	CASE(mul_f32_sgpr, "MUL_F32", 1, float, float, 1.0001f, 0)
	CASE(min_f32, "", 1, float, float, 0.5f, 0)
	CASE(max3_f32, "", 1, float, float, 0.5f, 0.25f)
	CASE(min3_f32, "", 1, float, float, 0.5f, 0.25f)
	CASE(med3_f32, "", 1, float, float, 0.5f, 0.25f)
	CASE(mov_b32, "", 1, unsigned, unsigned, 3u, 0)
	CASE(xor_b32, "INT32", 1, unsigned, unsigned, 3u, 0)
	CASE(or_b32, "INT32", 1, unsigned, unsigned, 3u, 0)
	CASE(lshlrev_b32, "INT32", 1, unsigned, unsigned, 0, 0)
	CASE(sub_u32, "INT32", 1, unsigned, unsigned, 3u, 0)
	CASE(lshl_add_u32, "INT32", 1, unsigned, unsigned, 3u, 0)
	CASE(add3_u32, "INT32", 1, unsigned, unsigned, 3u, 5u)
	CASE(and_or_b32, "INT32", 1, unsigned, unsigned, 3u, 5u)
	CASE(bfe_u32, "INT32", 1, unsigned, unsigned, 0, 0)
	CASE(bfi_b32, "INT32", 1, unsigned, unsigned, 3u, 5u)
	CASE(perm_b32, "", 1, unsigned, unsigned, 3u, 0x03020100u)
	CASE(alignbit_b32, "", 1, unsigned, unsigned, 3u, 0)
	CASE(pk_fma_f16, "", 1, unsigned, unsigned, 0x3c003c00u, 0x38003800u)
	CASE(pk_max_f16, "", 1, unsigned, unsigned, 0x3c003c00u, 0)
	CASE(pk_min_f16, "", 1, unsigned, unsigned, 0x3c003c00u, 0)
	CASE(pk_maximum3_f16, "", 1, unsigned, unsigned, 0x3c003c00u, 0x38003800u)
	CASE(pk_add_f16, "", 1, unsigned, unsigned, 0x3c003c00u, 0)
	CASE(pk_mul_f16, "", 1, unsigned, unsigned, 0x3c003c00u, 0)
	CASE(pk_max_i16, "", 1, unsigned, unsigned, 0x3c003c00u, 0)
	CASE(pk_mad_i16, "", 1, unsigned, unsigned, 0x00030003u, 0x00050005u)
	CASE(add_f32_sdwa, "", 1, float, float, 0.5f, 0)
	CASE(cvt_f32_u32_sdwa, "", 1, unsigned, unsigned, 0x11223344u, 0)
	CASE(cvt_pk_f32_fp8, "", 1, double, unsigned, 0x38383838u, 0)
	CASE(mul_u32_u24, "INT32", 1, unsigned, unsigned, 3u, 0)
	CASE(mad_u32_u24, "INT32", 1, unsigned, unsigned, 3u, 5u)
	CASE(add_co_u32, "INT32", 1, unsigned, unsigned, 3u, 0)
	CASE(addc_co_u32, "INT32", 1, unsigned, unsigned, 3u, 0)
	CASE(min_u32, "INT32", 1, unsigned, unsigned, 3u, 0)
	CASE(cvt_u32_f32, "CVT", 1, unsigned, unsigned, 0, 0)
	CASE(cvt_f32_i32, "CVT", 1, unsigned, unsigned, 0, 0)
	CASE(floor_f32, "", 1, float, float, 0, 0)
	CASE(fract_f32, "", 1, float, float, 0, 0)
	CASE(rndne_f32, "", 1, float, float, 0, 0)
	CASE(ldexp_f32, "", 1, float, int, 1, 0)
	CASE(frexp_mant_f32, "", 1, float, float, 0, 0)
	CASE(sin_f32, "TRANS_F32", 1, float, float, 0, 0)
	CASE(ldexp_f64, "", 1, double, int, 1, 0)
	CASE(rndne_f64, "", 1, double, double, 0, 0)
	CASE(cvt_f64_i32, "CVT", 1, double, int, 3, 0)
	CASE(pk_mul_f32, "", 1, double, double, 1.0, 0)
	CASE(pk_add_f32, "", 1, double, double, 1.0, 0)
	CASE(lshrrev_b64, "INT64", 1, unsigned long long, unsigned, 0, 0)
	CASE(cmp_only_f32, "", 1, unsigned, float, 0.5f, 1.5f)
	CASE(cmp_sgpr_f32, "", 1, unsigned, float, 0.5f, 1.5f)
	CASE(cmp_u32, "", 1, unsigned, unsigned, 3u, 5u)
	CASE(cmp_class_f32, "", 1, unsigned, float, 0.5f, 1.5f)
	CASE(readlane, "", 1, unsigned, unsigned, 0, 0)
	CASE(readfirstlane, "", 1, unsigned, unsigned, 0, 0)
	CASE(fma_f32_sgpr, "FMA_F32", 1, float, float, 1.0001f, 0)
	CASE(add_f32_sgpr, "ADD_F32", 1, float, float, 0.5f, 0)
	CASE(add_u32_sgpr, "INT32", 1, unsigned, unsigned, 3u, 0)
	CASE(and_b32_sgpr, "INT32", 1, unsigned, unsigned, 0xffffffu, 0)
	CASE(mov_b32_sgpr, "", 1, unsigned, unsigned, 3u, 0)
	CASE(mul_f32_inline, "MUL_F32", 1, float, float, 0, 0)
	CASE(mul_f32_literal, "MUL_F32", 1, float, float, 0, 0)
	CASE(add_u32_inline, "INT32", 1, unsigned, unsigned, 0, 0)
	CASE(add_u32_literal, "INT32", 1, unsigned, unsigned, 0, 0)
	CASE(and_b32_literal, "INT32", 1, unsigned, unsigned, 0, 0)
	CASE(fmaak_f32, "FMA_F32", 1, float, float, 1.0001f, 0)
	CASE(mul_f32_e64, "MUL_F32", 1, float, float, 1.0001f, 0)
	CASE(ashrrev_i32, "INT32", 1, unsigned, unsigned, 0, 0)
	CASE(lshlrev_b32_v, "INT32", 1, unsigned, unsigned, 1u, 0)
	CASE(lshrrev_b32_v, "INT32", 1, unsigned, unsigned, 1u, 0)
	CASE(not_b32, "INT32", 1, unsigned, unsigned, 0, 0)
	CASE(mbcnt_lo, "", 1, unsigned, unsigned, 0xffffu, 0)
	CASE(mbcnt_hi, "", 1, unsigned, unsigned, 0xffffu, 0)
	CASE(or3_b32, "INT32", 1, unsigned, unsigned, 3u, 5u)
	CASE(xad_u32, "INT32", 1, unsigned, unsigned, 3u, 5u)
	CASE(add_lshl_u32, "INT32", 1, unsigned, unsigned, 3u, 0)
	CASE(cmp_cndmask, "", 2, float, float, 0.5f, 1.5f)
	CASE(cmp_cndmask_e64, "", 2, float, float, 0.5f, 1.5f)
	CASE(cndmask_e64, "", 1, unsigned, unsigned, 3u, 0)
	CASE(cmp_x2_cndmask, "", 3, float, float, 0.5f, 1.5f)
	CASE(writelane, "", 1, unsigned, unsigned, 3u, 0)
	CASE(bpermute, "", 1, unsigned, unsigned, 4u, 0)
	CASE(fma_mix_f32_lo, "", 1, float, unsigned, 0x3c003c00u, 0)
	CASE(fma_mix_f32_hi, "", 1, float, unsigned, 0x3c003c00u, 0)
	CASE(fma_mix_f32_f32, "", 1, float, float, 1.0001f, 0.5f)
	CASE(bitop3_b32, "INT32", 1, unsigned, unsigned, 3u, 5u)
	CASE(maximum3_f32, "", 1, float, float, 0.5f, 0.25f)
	CASE(minimum3_f32, "", 1, float, float, 0.5f, 0.25f)
	CASE(cvt_f32_f16, "CVT", 1, unsigned, unsigned, 0x3c003c00u, 0)
	CASE(cvt_f32_f16_sdwa, "CVT", 1, unsigned, unsigned, 0x3c003c00u, 0)
	CASE(dot2_f32_f16, "", 1, float, unsigned, 0x3c003c00u, 0x38003800u)
	CASE(mul_f32_sdwa_byte, "", 1, float, unsigned, 0x11223344u, 0)
	CASE(lshl_or_b32, "INT32", 1, unsigned, unsigned, 3u, 0)
	CASE(fmac_f32_dpp, "", 1, float, float, 1.0001f, 0.5f)
	CASE(cvt_f32_ubyte3, "CVT", 1, unsigned, unsigned, 0x11223344u, 0)
	CASE(ffbh_u32, "INT32", 1, unsigned, unsigned, 0, 0)
	CASE(bcnt_u32, "INT32", 1, unsigned, unsigned, 3u, 0)
	const double* base = cases[0].ms;
	printf("%-16s %-10s | ms at 1 / 4 / 8 waves per SIMD | issue cycles per wave64 instruction (v_fma_f32 = 2) at 1 / 4 / 8\n", "opcode", "class");
	std::string json = "{\n \"note\": \"issue cycles per wave64 VALU instruction at 4 waves per SIMD = 2 x (time of an independent stream of the opcode / time of the same stream of v_fma_f32), tools/valu_calib.hip on the bench box; v_fma_f32 = 2 cycles per MI355X_MICROARCH.md\",\n \"device\": \"";
	json += prop.gcnArchName; json += "\",\n \"cycles\": {\n";
	for (size_t i = 0; i < cases.size(); ++i) {
		const Case& c = cases[i];
		double cyc[3]; for (int w = 0; w < 3; ++w) cyc[w] = 2.0 * (c.ms[w] / c.perCount) / base[w];
		printf("%-16s %-10s | %8.3f %8.3f %8.3f | %6.2f %6.2f %6.2f\n", c.name, c.counterClass, c.ms[0], c.ms[1], c.ms[2], cyc[0], cyc[1], cyc[2]);
		char buf[256]; snprintf(buf, sizeof(buf), "  \"%s\": {\"class\": \"%s\", \"w1\": %.3f, \"w4\": %.3f, \"w8\": %.3f}%s\n", c.name, c.counterClass, cyc[0], cyc[1], cyc[2], i + 1 < cases.size() ? "," : "");
		json += buf;
	}
	{
		// absolute rate of the reference stream: wave-level v_fma_f32 per second per SIMD at 4 waves per SIMD (x 2 cycles = the clock the chip held)
		const double insts = (double)iters * PER_ITER * 4;
		char buf[256]; snprintf(buf, sizeof(buf), " },\n \"fma_f32_wave_insts_per_s_per_simd_w4\": %.4g,\n \"implied_clock_ghz_w4\": %.4f\n}\n", insts / (base[1] * 1e-3), 2.0 * insts / (base[1] * 1e-3) / 1e9);
		json += buf;
		printf("v_fma_f32 at 4 waves/SIMD: %.1f M wave-instructions/s per SIMD -> clock held ~%.3f GHz if 2 cycles each\n", insts / (base[1] * 1e-3) / 1e6, 2.0 * insts / (base[1] * 1e-3) / 1e9);
	}
	if (argc > 1) { FILE* f = fopen(argv[1], "w"); if (f) { fputs(json.c_str(), f); fclose(f); } }
	return 0;
}
