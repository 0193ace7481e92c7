// Bit-exact restatement, for host AND device, of the glibc 2.35 (x86-64) single-precision
// math routines that the reference renderer's build calls on its hot path:
//   sinf cosf expf logf powf   -- glibc's double-precision-core routines (ARM optimized
//                                 routines), in the *_fma ifunc variants x86-64 hosts with
//                                 FMA select: FMA placement is written out explicitly
//   acosf asinf atanf atan2f tanf -- glibc's fdlibm-derived float routines (no FMA variant)
// The reference's results are defined by these (its g++ build links them); a GPU path
// that wants the reference's pixels needs the same roundings, not merely < 1 ulp.
//
// Tables are glibc's published constants (__exp2f_data, __logf_data, __powf_log2_data,
// __sincosf_table, __inv_pio4).  tools/check_glibc_math.cc verifies every function here
// against the host libm over the full float domain (or the stated sub-domain).
//
// Everything is plain C arithmetic on IEEE binary32/binary64 + fma: identical on x86-64
// and gfx950 (both have correctly rounded fp64 add/mul/fma, fp32 add/mul/div/sqrt).
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RLM_FN __host__ __device__ static inline
#else
#define RLM_FN static inline
#endif

namespace rlm {

RLM_FN uint32_t asuint(float f) { union { float f; uint32_t i; } u; u.f = f; return u.i; }
RLM_FN float asfloat(uint32_t i) { union { uint32_t i; float f; } u; u.i = i; return u.f; }
RLM_FN uint64_t asuint64(double f) { union { double f; uint64_t i; } u; u.f = f; return u.i; }
RLM_FN double asdouble(uint64_t i) { union { uint64_t i; double f; } u; u.i = i; return u.f; }
RLM_FN double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
// A double constant that has to sit in a VGPR pair (gfx9 VALU instructions take one scalar operand; an fma with two constant operands
// needs the other in vector registers): materialised where it is used.  Left to itself the compiler hoists the two v_mov out of the
// megakernel's bounce loop, finds no registers for a pair that lives that long, and spills it to scratch -- one scratch reload, a trip
// through the vector memory pipeline, per inlined call.
#if defined(__HIP_DEVICE_COMPILE__)
#define RLM_LOCAL_CONST(name, value) \
	uint32_t name##_lo = (uint32_t)__builtin_bit_cast(uint64_t, (double)(value)), name##_hi = (uint32_t)(__builtin_bit_cast(uint64_t, (double)(value)) >> 32); \
	asm volatile("" : "+v"(name##_lo), "+v"(name##_hi)); /* two 32-bit immediates: trivially rematerialisable, never spilled */ \
	const double name = __builtin_bit_cast(double, ((uint64_t)name##_hi << 32) | name##_lo)
#else
#define RLM_LOCAL_CONST(name, value) const double name = (value)
#endif
// sqrtf(x) and 1.0f / x: on the device the short exact sequences described in rl_math.h (RL_EXACT_FAST_RCP_SQRT=0: the compiler's expansions)
#ifndef RL_EXACT_FAST_RCP_SQRT
#define RL_EXACT_FAST_RCP_SQRT 1
#endif
#if defined(__HIP_DEVICE_COMPILE__) && RL_EXACT_FAST_RCP_SQRT
// (the out-of-range path inline: as a call -- __noinline__ -- every one of the ~45 sites constrains the register allocation around it, and the Cornell
// frame took 14.75 ms instead of 14.00; measured, round 3)
#ifndef RL_EXACT_SLOW_ATTR
#define RL_EXACT_SLOW_ATTR __forceinline__
#endif
__device__ RL_EXACT_SLOW_ATTR static float rcp1_slow_(float x) { return 1.0f / x; }
__device__ RL_EXACT_SLOW_ATTR static float sqrt_slow_(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ static float rcp1_(float x)
{
	if (((asuint(x) & 0x7fffffffu) - 0x00800000u) < (0x7e800000u - 0x00800000u)) {   // 2^-126 <= |x| < 2^126
		const float r0 = __builtin_amdgcn_rcpf(x);
		return __builtin_fmaf(__builtin_fmaf(-x, r0, 1.0f), r0, r0);
	}
	return rcp1_slow_(x);
}
__device__ __forceinline__ static float sqrtf_(float x)
{
	if ((asuint(x) - 0x0d000000u) < (0x7f800000u - 0x0d000000u)) {   // 2^-101 <= x <= FLT_MAX
		const float y = __builtin_amdgcn_rsqf(x);
		const float g = x * y, h = 0.5f * y;
		return __builtin_fmaf(__builtin_fmaf(-g, g, x), h, g);
	}
	return sqrt_slow_(x);
}
#else
RLM_FN float rcp1_(float x) { return 1.0f / x; }
RLM_FN float sqrtf_(float x) { return __builtin_sqrtf(x); }
#endif
RLM_FN float fabsf_(float x) { return __builtin_fabsf(x); }

// ---------------------------------------------------------------------------------------
// tables (function-local statics would not work on device; kept as macros expanding to
// constant arrays inside each function so both compilers place them in constant memory)
#define RLM_EXP2F_TAB { \
	0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, \
	0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, \
	0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull, \
	0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, \
	0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull, \
	0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, \
	0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, \
	0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull }

#define RLM_LOG_INVC { \
	0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0, 0x1.3c995b0b80385p+0, \
	0x1.30d190c8864a5p+0, 0x1.25e227b0b8eap+0, 0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0, \
	0x1.0953f419900a7p+0, 0x1p+0, 0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1, \
	0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1 }
#define RLM_LOGF_LOGC { \
	-0x1.57bf7808caadep-2, -0x1.2bef0a7c06ddbp-2, -0x1.01eae7f513a67p-2, -0x1.b31d8a68224e9p-3, \
	-0x1.6574f0ac07758p-3, -0x1.1aa2bc79c81p-3, -0x1.a4e76ce8c0e5ep-4, -0x1.1973c5a611cccp-4, \
	-0x1.252f438e10c1ep-5, 0x0p+0, 0x1.aa5aa5df25984p-5, 0x1.c5e53aa362eb4p-4, \
	0x1.526e57720db08p-3, 0x1.bc2860d22477p-3, 0x1.1058bc8a07ee1p-2, 0x1.4043057b6ee09p-2 }
#define RLM_POWF_LOGC { \
	-0x1.efec65b963019p-2, -0x1.b0b6832d4fca4p-2, -0x1.7418b0a1fb77bp-2, -0x1.39de91a6dcf7bp-2, \
	-0x1.01d9bf3f2b631p-2, -0x1.97c1d1b3b7afp-3, -0x1.2f9e393af3c9fp-3, -0x1.960cbbf788d5cp-4, \
	-0x1.a6f9db6475fcep-5, 0x0p+0, 0x1.338ca9f24f53dp-4, 0x1.476a9543891bap-3, \
	0x1.e840b4ac4e4d2p-3, 0x1.40645f0c6651cp-2, 0x1.88e9c2c1b9ff8p-2, 0x1.ce0a44eb17bccp-2 }

// Where the per-lane table look-ups go.  Host, and device builds without RLM_LDS_TABLES: constant arrays local to each function
// (constant memory on the device: a 64-address gather through the vector memory pipeline per call).  Device builds that define
// RLM_LDS_TABLES before including this header provide `__shared__ double rlm_lds_tab[80]` (INVC, logf LOGC, powf LOGC, exp2f T as
// bit patterns) filled by rlm_fill_lds_tables() at the start of every kernel that reaches these functions: an LDS read instead.
#if defined(RLM_LDS_TABLES) && defined(__HIP_DEVICE_COMPILE__)
#define RLM_DECL_LOGF_TABLES
#define RLM_DECL_POWF_TABLES
#define RLM_DECL_EXP2F_TABLE
#define RLM_INVC(i) rlm_lds_tab[(i)]
#define RLM_LOGF_LOGC_AT(i) rlm_lds_tab[16 + (i)]
#define RLM_POWF_LOGC_AT(i) rlm_lds_tab[32 + (i)]
#define RLM_EXP2F_AT(i) asuint64(rlm_lds_tab[48 + (i)])
#else
#define RLM_DECL_LOGF_TABLES const double INVC[16] = RLM_LOG_INVC, LOGC[16] = RLM_LOGF_LOGC;
#define RLM_DECL_POWF_TABLES const double INVC[16] = RLM_LOG_INVC, LOGC[16] = RLM_POWF_LOGC;
#define RLM_DECL_EXP2F_TABLE const uint64_t T[32] = RLM_EXP2F_TAB;
#define RLM_INVC(i) INVC[(i)]
#define RLM_LOGF_LOGC_AT(i) LOGC[(i)]
#define RLM_POWF_LOGC_AT(i) LOGC[(i)]
#define RLM_EXP2F_AT(i) T[(i)]
#endif

// ---------------------------------------------------------------------------------------
// expf  (glibc sysdeps/ieee754/flt-32/e_expf.c, non-TOINT path, FMA-contracted)
RLM_FN float expf_(float x)
{
	RLM_DECL_EXP2F_TABLE
	const double SHIFT = 0x1.8p+52, InvLn2N = 0x1.71547652b82fep+5;
	const double C0 = 0x1.c6af84b912394p-20, C2 = 0x1.62e42ff0c52d6p-6;
	RLM_LOCAL_CONST(C1, 0x1.ebfce50fac4f3p-13);
	double xd = (double)x;
	uint32_t abstop = (asuint(x) >> 20) & 0x7ff;
	if (abstop >= ((asuint(88.0f) >> 20) & 0x7ff)) {
		if (asuint(x) == 0xff800000u) return 0.0f;
		if (abstop >= 0x7f8) return x + x;
		if (x > 0x1.62e42ep6f) return asfloat(0x7f800000u);          // overflow -> +inf
		if (x < -0x1.9fe368p6f) return 0.0f;                          // underflow -> +0
	}
	// z = InvLn2N * xd has two uses, both additions, so the FMA build fuses the product into both
	double kd = fma_(InvLn2N, xd, SHIFT);
	uint64_t ki = asuint64(kd);
	kd -= SHIFT;
	double r = fma_(InvLn2N, xd, -kd);
	double z;
	uint64_t t = RLM_EXP2F_AT(ki % 32);
	t += ki << (52 - 5);
	double s = asdouble(t);
	z = fma_(C0, r, C1);
	double r2 = r * r;
	double y = fma_(C2, r, 1.0);
	y = fma_(z, r2, y);
	y = y * s;
	return (float)y;
}

// logf  (glibc sysdeps/ieee754/flt-32/e_logf.c, FMA-contracted)
RLM_FN float logf_(float x)
{
	RLM_DECL_LOGF_TABLES
	const double Ln2 = 0x1.62e42fefa39efp-1;
	const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2;
	RLM_LOCAL_CONST(A2, -0x1.ffffef20a4123p-2);
	uint32_t ix = asuint(x);
	if (ix == 0x3f800000u) return 0.0f;
	if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
		if (ix * 2 == 0) return asfloat(0xff800000u);                 // log(0) = -inf
		if (ix == 0x7f800000u) return x;
		if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return asfloat(0x7fc00000u) ; // invalid -> NaN (sign handled by caller tests)
		ix = asuint(x * 0x1p23f);
		ix -= 23u << 23;
	}
	uint32_t tmp = ix - 0x3f330000u;
	int i = (tmp >> (23 - 4)) % 16;
	int k = (int32_t)tmp >> 23;
	uint32_t iz = ix - (tmp & (0x1ffu << 23));
	double invc = RLM_INVC(i), logc = RLM_LOGF_LOGC_AT(i);
	double z = (double)asfloat(iz);
	double r = fma_(z, invc, -1.0);
	double y0 = fma_((double)k, Ln2, logc);
	double r2 = r * r;
	double y = fma_(A1, r, A2);
	y = fma_(A0, r2, y);
	y = fma_(y, r2, y0 + r);
	return (float)y;
}

// powf  (glibc sysdeps/ieee754/flt-32/e_powf.c, non-TOINT path, FMA-contracted)
RLM_FN int powf_checkint(uint32_t iy)
{
	int e = iy >> 23 & 0xff;
	if (e < 0x7f) return 0;
	if (e > 0x7f + 23) return 2;
	if (iy & ((1u << (0x7f + 23 - e)) - 1)) return 0;
	if (iy & (1u << (0x7f + 23 - e))) return 1;
	return 2;
}
RLM_FN int powf_zeroinfnan(uint32_t ix) { return 2 * ix - 1 >= 2u * 0x7f800000u - 1; }

RLM_FN float powf_(float x, float y)
{
	RLM_DECL_POWF_TABLES
	RLM_DECL_EXP2F_TABLE
	const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2,
	             A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp+0;
	const double C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;
	const double SHIFT = 0x1.8p+47;
	uint32_t sign_bias = 0;
	uint32_t ix = asuint(x), iy = asuint(y);
	if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || powf_zeroinfnan(iy)) {
		if (powf_zeroinfnan(iy)) {
			if (2 * iy == 0) return 1.0f;
			if (ix == 0x3f800000u) return 1.0f;
			if (2 * ix > 2u * 0x7f800000u || 2 * iy > 2u * 0x7f800000u) return x + y;
			if (2 * ix == 2 * 0x3f800000u) return 1.0f;
			if ((2 * ix < 2 * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;
			return y * y;
		}
		if (powf_zeroinfnan(ix)) {
			float x2 = x * x;
			if ((ix & 0x80000000u) && powf_checkint(iy) == 1) { x2 = -x2; sign_bias = 1; }
			return (iy & 0x80000000u) ? 1 / x2 : x2;
		}
		if (ix & 0x80000000u) {
			int yint = powf_checkint(iy);
			if (yint == 0) return asfloat(0x7fc00000u);                // invalid
			if (yint == 1) sign_bias = 1u << (5 + 11);
			ix &= 0x7fffffffu;
		}
		if (ix < 0x00800000u) {
			ix = asuint(x * 0x1p23f);
			ix &= 0x7fffffffu;
			ix -= 23u << 23;
		}
	}
	// log2_inline
	uint32_t tmp = ix - 0x3f330000u;
	int i = (tmp >> (23 - 4)) % 16;
	uint32_t top = tmp & 0xff800000u;
	uint32_t iz = ix - top;
	int k = (int32_t)top >> 23;
	double invc = RLM_INVC(i), logc = RLM_POWF_LOGC_AT(i);
	double z = (double)asfloat(iz);
	double r = fma_(z, invc, -1.0);
	double y0 = logc + (double)k;
	double r2 = r * r;
	double yy = fma_(A0, r, A1);
	double p = fma_(A2, r, A3);
	double r4 = r2 * r2;
	double q = fma_(A4, r, y0);
	q = fma_(p, r2, q);
	yy = fma_(yy, r4, q);
	double logx = yy;
	double ylogx = (double)y * logx;
	if ((asuint64(ylogx) >> 47 & 0xffff) >= (asuint64(126.0) >> 47)) {
		if (ylogx > 0x1.fffffffd1d571p+6) return sign_bias ? asfloat(0xff800000u) : asfloat(0x7f800000u);
		if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;
	}
	// exp2_inline
	double kd = ylogx + SHIFT;
	uint64_t ki = asuint64(kd);
	kd -= SHIFT;
	r = ylogx - kd;
	uint64_t t = RLM_EXP2F_AT(ki % 32);
	uint64_t ski = ki + sign_bias;
	t += ski << (52 - 5);
	double s = asdouble(t);
	z = fma_(C0, r, C1);
	r2 = r * r;
	double e = fma_(C2, r, 1.0);
	e = fma_(z, r2, e);
	e = e * s;
	return (float)e;
}

// ---------------------------------------------------------------------------------------
// sinf / cosf  (glibc sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, s_sincosf.h; non-TOINT, FMA)
// __sincosf_table[0]; table [1] is the same with the cosine coefficients negated, which in
// round-to-nearest is exactly "negate the cosine-branch result" (negation commutes with fma and
// with the final rounding), so one set of constants serves both.
#define RLM_SC_C0 0x1p+0
#define RLM_SC_C1 -0x1.ffffffd0c621cp-2
#define RLM_SC_S1 -0x1.555545995a603p-3
#define RLM_SC_C2 0x1.55553e1068f19p-5
#define RLM_SC_S2 0x1.1107605230bc4p-7
#define RLM_SC_C3 -0x1.6c087e89a359dp-10
#define RLM_SC_S3 -0x1.994eb3774cf24p-13
#define RLM_SC_C4 0x1.99343027bf8c3p-16

// negcos: use __sincosf_table[1] (chosen by the caller when quadrant & 2)
RLM_FN float sincosf_poly(double x, double x2, bool negcos, int n)
{
	if ((n & 1) == 0) {
		double x3 = x * x2;
		double s1 = fma_(x2, RLM_SC_S3, RLM_SC_S2);
		double x7 = x3 * x2;
		double s = fma_(x3, RLM_SC_S1, x);
		return (float)fma_(x7, s1, s);
	} else {
		double x4 = x2 * x2;
		double c2 = fma_(x2, RLM_SC_C4, RLM_SC_C3);
		double c1 = fma_(x2, RLM_SC_C1, RLM_SC_C0);
		double x6 = x4 * x2;
		double c = fma_(x4, RLM_SC_C2, c1);
		float r = (float)fma_(x6, c2, c);
		return negcos ? -r : r;
	}
}

RLM_FN double sincosf_reduce_large(uint32_t xi, int* np)
{
	const uint32_t inv_pio4[24] = {
		0xa2, 0xa2f9, 0xa2f983, 0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529,
		0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1, 0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0,
		0x34ddc0db, 0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041 };
	const uint32_t* arr = &inv_pio4[(xi >> 26) & 15];
	int shift = (xi >> 23) & 7;
	uint64_t n, res0, res1, res2;
	xi = (xi & 0xffffff) | 0x800000;
	xi <<= shift;
	res0 = xi * arr[0];
	res1 = (uint64_t)xi * arr[4];
	res2 = (uint64_t)xi * arr[8];
	res0 = (res2 >> 32) | (res0 << 32);
	res0 += res1;
	n = (res0 + (1ULL << 61)) >> 62;
	res0 -= n << 62;
	double x = (double)(int64_t)res0;
	*np = (int)n;
	return x * 0x1.921FB54442D18p-62;
}

RLM_FN double sincosf_sign(int q) { q &= 3; return (q == 1 || q == 2) ? -1.0 : 1.0; }   // sign[] = {1,-1,-1,1}

// |y| < 120 (the only range the renderer produces): one path for both glibc branches.  For |y| < pi/4
// glibc skips the reduction; running it anyway yields n = 0, x - 0*hpi = x and sign 1, i.e. the same
// operands, so the results are identical and lanes on either side of pi/4 do not diverge.
template <int COS>
RLM_FN float sincosf_(float y)
{
	const double hpi_inv = 0x1.45f306dc9c883p+23, hpi = 0x1.921fb54442d18p+0;
	double x = (double)y;
	const uint32_t top = (asuint(y) >> 20) & 0x7ff;
	int n;
	if (top < ((asuint(120.0f) >> 20) & 0x7ff)) {
		if (top < ((asuint(0x1p-12f) >> 20) & 0x7ff)) return COS ? 1.0f : y;
		double r = x * hpi_inv;
		n = ((int32_t)r + 0x800000) >> 24;
		x = fma_(-(double)n, hpi, x);
		double s = sincosf_sign(n);
		return sincosf_poly(x * s, x * x, (n & 2) != 0, COS ? (n ^ 1) : n);
	} else if (top < 0x7f8) {
		uint32_t xi = asuint(y);
		int sign = xi >> 31;
		x = sincosf_reduce_large(xi, &n);
		double s = sincosf_sign(n + sign);
		return sincosf_poly(x * s, x * x, ((n + sign) & 2) != 0, COS ? (n ^ 1) : n);
	}
	return asfloat(0x7fc00000u);
}
RLM_FN float sinf_(float x) { return sincosf_<0>(x); }
RLM_FN float cosf_(float x) { return sincosf_<1>(x); }

// sinf(y) and cosf(y) together: one reduction, both polynomials (each identical to the separate calls).
RLM_FN void sincosf_both(float y, float* outSin, float* outCos)
{
	const double hpi_inv = 0x1.45f306dc9c883p+23, hpi = 0x1.921fb54442d18p+0;
	const uint32_t top = (asuint(y) >> 20) & 0x7ff;
	if (top < ((asuint(120.0f) >> 20) & 0x7ff) && top >= ((asuint(0x1p-12f) >> 20) & 0x7ff)) {
		double x = (double)y;
		double r = x * hpi_inv;
		int n = ((int32_t)r + 0x800000) >> 24;
		x = fma_(-(double)n, hpi, x);
		const double xs = x * sincosf_sign(n), x2 = x * x;
		const bool neg = (n & 2) != 0;
		*outSin = sincosf_poly(xs, x2, neg, n);
		*outCos = sincosf_poly(xs, x2, neg, n ^ 1);
		return;
	}
	*outSin = sincosf_<0>(y);
	*outCos = sincosf_<1>(y);
}

// Signs of sinf(y) and cosf(y) as glibc returns them (bit 31 of the results), without the polynomials: the quadrant n and
// the sign of the reduced argument decide them (an odd polynomial has its argument's sign, the cosine polynomial is positive
// on [-pi/4, pi/4], table [1] negates it).  Valid for 0 <= y < 120; used where only +-0 products of the values matter.
RLM_FN void sincosf_signs(float y, bool* sinNeg, bool* cosNeg)
{
	const double hpi_inv = 0x1.45f306dc9c883p+23, hpi = 0x1.921fb54442d18p+0;
	const uint32_t top = (asuint(y) >> 20) & 0x7ff;
	if (top < ((asuint(0x1p-12f) >> 20) & 0x7ff)) { *sinNeg = (asuint(y) >> 31) != 0; *cosNeg = false; return; }
	double x = (double)y;
	double r = x * hpi_inv;
	int n = ((int32_t)r + 0x800000) >> 24;
	x = fma_(-(double)n, hpi, x);
	const bool xsNeg = (asuint64(x * sincosf_sign(n)) >> 63) != 0;
	const bool tab1 = (n & 2) != 0;
	*sinNeg = (n & 1) ? tab1 : xsNeg;
	*cosNeg = (n & 1) ? xsNeg : tab1;
}

// ---------------------------------------------------------------------------------------
// fdlibm-derived float routines (glibc sysdeps/ieee754/flt-32/e_acosf.c, e_asinf.c,
// s_atanf.c, e_atan2f.c, s_tanf.c + k_tanf.c + e_rem_pio2f.c); float arithmetic, no FMA.
RLM_FN float acosf_(float x)
{
	// Same operations per input as glibc's three-way branch (|x|<0.5, x<-0.5, x>0.5), arranged so that a
	// wave whose lanes fall into different ranges evaluates the shared rational p(z)/q(z) once: only the
	// choice of z and the short tails differ.  Values computed for a range the lane is not in are discarded.
	const float one = 1.0000000000e+00f, pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f,
		pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f,
		pS4 = 7.9153501429e-04f, pS5 = 3.4793309169e-05f,
		qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f, qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
	const int32_t hx = (int32_t)asuint(x), ix = hx & 0x7fffffff;
	if (ix >= 0x3f800000) {
		if (ix == 0x3f800000) return (hx > 0) ? 0.0f : pi + 2.0f * pio2_lo;
		return (x - x) / (x - x);
	}
	const bool small = ix < 0x3f000000;
	if (small && ix <= 0x23000000) return pio2_hi + pio2_lo;
	const bool neg = hx < 0;
	const float z = small ? x * x : (neg ? (one + x) * 0.5f : (one - x) * 0.5f);
	const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
	const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
	const float r = p / q;
	const float s = sqrtf_(z);
	// |x| < 0.5
	const float rSmall = pio2_hi - (x - (pio2_lo - r * x));
	// x < -0.5
	const float wNeg = r * s - pio2_lo;
	const float rNeg = pi - 2.0f * (s + wNeg);
	// x > 0.5
	const float df = asfloat(asuint(s) & 0xfffff000u);
	const float c = (z - df * df) / (s + df);
	const float wPos = r * s + c;
	const float rPos = 2.0f * (df + wPos);
	return small ? rSmall : (neg ? rNeg : rPos);
}

// asinf (glibc e_asinf.c: glibc's own minimax p0..p4, not fdlibm's rational)
RLM_FN float asinf_(float x)
{
	const float one = 1.0f, huge = 1.000e+30f,
		pio2_hi = 1.57079637050628662109375f, pio2_lo = -4.37113900018624283e-8f, pio4_hi = 0.785398185253143310546875f,
		p0 = 1.666675248e-1f, p1 = 7.495297643e-2f, p2 = 4.547037598e-2f, p3 = 2.417951451e-2f, p4 = 4.216630880e-2f;
	float t, w, p, q, c, r, s;
	int32_t hx = (int32_t)asuint(x), ix = hx & 0x7fffffff;
	if (ix == 0x3f800000) {
		return x * pio2_hi + x * pio2_lo;
	} else if (ix > 0x3f800000) {
		return (x - x) / (x - x);
	} else if (ix < 0x3f000000) {
		if (ix < 0x32000000) {
			if (huge + x > one) return x;
		} else {
			t = x * x;
			w = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
			return x + x * w;
		}
	}
	w = one - fabsf_(x);
	t = w * 0.5f;
	p = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
	s = sqrtf_(t);
	if (ix >= 0x3F79999A) {
		t = pio2_hi - (2.0f * (s + s * p) - pio2_lo);
	} else {
		w = asfloat(asuint(s) & 0xfffff000u);
		c = (t - w * w) / (s + w);
		r = p;
		p = 2.0f * s * r - (pio2_lo - 2.0f * c);
		q = pio4_hi - 2.0f * w;
		t = pio4_hi - (p - q);
	}
	if (hx > 0) return t; else return -t;
}

// atanf (glibc s_atanf.c)
RLM_FN float atanf_(float x)
{
	const float atanhi[4] = { 4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f };
	const float atanlo[4] = { 5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f };
	const float aT[11] = { 3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f,
		-7.6918758452e-02f, 6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f };
	const float one = 1.0f, huge = 1.0e30f;
	float w, s1, s2, z;
	int32_t hx = (int32_t)asuint(x), ix = hx & 0x7fffffff, id;
	if (ix >= 0x4c000000) {
		if (ix > 0x7f800000) return x + x;
		if (hx > 0) return atanhi[3] + atanlo[3];
		else return -atanhi[3] - atanlo[3];
	}
	if (ix < 0x3ee00000) {
		if (ix < 0x31000000) {
			if (huge + x > one) return x;
		}
		id = -1;
	} else {
		x = fabsf_(x);
		if (ix < 0x3f980000) {
			if (ix < 0x3f300000) { id = 0; x = (2.0f * x - one) / (2.0f + x); }
			else { id = 1; x = (x - one) / (x + one); }
		} else {
			if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (one + 1.5f * x); }
			else { id = 3; x = -1.0f / x; }
		}
	}
	z = x * x;
	w = z * z;
	s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
	s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
	if (id < 0) return x - x * (s1 + s2);
	z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
	return (hx < 0) ? -z : z;
}

// atan2f (glibc e_atan2f.c)
RLM_FN float atan2f_(float y, float x)
{
	const float tiny = 1.0e-30f, zero = 0.0f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f,
		pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
	float z;
	int32_t hx = (int32_t)asuint(x), ix = hx & 0x7fffffff, hy = (int32_t)asuint(y), iy = hy & 0x7fffffff, k, m;
	if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
	if (hx == 0x3f800000) return atanf_(y);
	m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
	if (iy == 0) {
		switch (m) {
			case 0: case 1: return y;
			case 2: return pi + tiny;
			case 3: return -pi - tiny;
		}
	}
	if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
	if (ix == 0x7f800000) {
		if (iy == 0x7f800000) {
			switch (m) {
				case 0: return pi_o_4 + tiny;
				case 1: return -pi_o_4 - tiny;
				case 2: return 3.0f * pi_o_4 + tiny;
				case 3: return -3.0f * pi_o_4 - tiny;
			}
		} else {
			switch (m) {
				case 0: return zero;
				case 1: return -zero;
				case 2: return pi + tiny;
				case 3: return -pi - tiny;
			}
		}
	}
	if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
	k = (iy - ix) >> 23;
	if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
	else if (hx < 0 && k < -60) z = 0.0f;
	else z = atanf_(fabsf_(y / x));
	switch (m) {
		case 0: return z;
		case 1: return asfloat(asuint(z) ^ 0x80000000u);
		case 2: return pi - (z - pi_lo);
		default: return (z - pi_lo) - pi;
	}
}

// __kernel_tanf (glibc k_tanf.c)
RLM_FN float kernel_tanf(float x, float y, int iy)
{
	const float one = 1.0f, pio4 = 7.8539812565e-01f, pio4lo = 3.7748947079e-08f;
	const float T[13] = { 3.3333334327e-01f, 1.3333334029e-01f, 5.3968254477e-02f, 2.1869488060e-02f, 8.8632395491e-03f,
		3.5920790397e-03f, 1.4562094584e-03f, 5.8804126456e-04f, 2.4646313977e-04f, 7.8179444245e-05f,
		7.1407252108e-05f, -1.8558637748e-05f, 2.5907305826e-05f };
	float z, r, v, w, s;
	int32_t hx = (int32_t)asuint(x), ix = hx & 0x7fffffff;
	if (ix < 0x39000000) {
		if ((int)x == 0) {
			if ((ix | (iy + 1)) == 0) return one / fabsf_(x);
			else if (iy == 1) return x;
			else return -one / x;
		}
	}
	if (ix >= 0x3f2ca140) {
		if (hx < 0) { x = -x; y = -y; }
		z = pio4 - x;
		w = pio4lo - y;
		x = z + w; y = 0.0f;
		if (fabsf_(x) < 0x1p-13f) return (1 - ((hx >> 30) & 2)) * iy * (1.0f - 2 * iy * x);
	}
	z = x * x;
	w = z * z;
	r = T[1] + w * (T[3] + w * (T[5] + w * (T[7] + w * (T[9] + w * T[11]))));
	v = z * (T[2] + w * (T[4] + w * (T[6] + w * (T[8] + w * (T[10] + w * T[12])))));
	s = z * x;
	r = y + z * (s * (r + v) + y);
	r += T[0] * s;
	w = x + r;
	if (ix >= 0x3f2ca140) {
		v = (float)iy;
		return (float)(1 - ((hx >> 30) & 2)) * (v - 2.0f * (x - (w * w / (w + v) - r)));
	}
	if (iy == 1) return w;
	{
		float a, t;
		z = asfloat(asuint(w) & 0xfffff000u);
		v = r - (z - x);
		t = a = -1.0f / w;
		t = asfloat(asuint(t) & 0xfffff000u);
		s = 1.0f + t * z;
		return t + a * (s + t * v);
	}
}

// tanf (glibc 2.35 s_tanf.c): range reduction shares sinf/cosf's reduce_fast / reduce_large
// (double arithmetic, NOT fused: tanf has no FMA ifunc variant), then fdlibm's __kernel_tanf
// on the float head/tail of the reduced argument.
RLM_FN float tanf_(float x)
{
	const double hpi_inv = 0x1.45f306dc9c883p+23, hpi = 0x1.921fb54442d18p+0;
	const uint32_t ux = asuint(x);
	const int32_t ix = (int32_t)(ux & 0x7fffffff);
	if (ix >= 0x7f800000) return x - x;
	double dx = (double)x;
	int n;
	if (((ux >> 20) & 0x7ff) < 0x42f) {
		// glibc calls __kernel_tanf(x, 0, 1) directly for |x| <= pi/4; the reduction gives n = 0,
		// dx = x - 0*hpi = x, head x and tail 0 there, i.e. the same call -- one path, no divergence.
		double r = dx * hpi_inv;
		n = ((int32_t)r + 0x800000) >> 24;
		double nh = (double)n * hpi;
		dx = dx - nh;
	} else {
		dx = sincosf_reduce_large(ux, &n);
		if (ux >> 31) dx = -dx;
	}
	float y0 = (float)dx;
	float y1 = (float)(dx - (double)y0);
	return kernel_tanf(y0, y1, 1 - ((n & 1) << 1));
}

#if defined(RLM_LDS_TABLES) && defined(__HIPCC__)
// First statement of every kernel that reaches expf_ / logf_ / powf_ (all threads of the block must call it: it ends in a barrier).
__device__ __forceinline__ void rlm_fill_lds_tables()
{
#if defined(__HIP_DEVICE_COMPILE__)
	const double INVC[16] = RLM_LOG_INVC, LOGF[16] = RLM_LOGF_LOGC, POWF[16] = RLM_POWF_LOGC;
	const uint64_t T[32] = RLM_EXP2F_TAB;
	const unsigned i = threadIdx.x;
	if (i < 16u) { rlm_lds_tab[i] = INVC[i]; rlm_lds_tab[16u + i] = LOGF[i]; rlm_lds_tab[32u + i] = POWF[i]; }
	if (i < 32u) rlm_lds_tab[48u + i] = asdouble(T[i]);
	__syncthreads();
#endif
}
#endif

} // namespace rlm
