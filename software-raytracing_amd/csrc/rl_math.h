// Device math for the megakernel.  Every transcendental the reference's hot path
// calls (reference render/material.cc, render/brdf.h, core/random.cc,
// render/texture.cc, render/renderer.cc:159-181) goes through one of these
// wrappers, so that the numerical policy lives in one place:
//   + - * / sqrt : IEEE binary32, correctly rounded (hipcc default for HIP), and the
//                  file is compiled with -ffp-contract=off (no FMA contraction), so
//                  these match the x86-64 reference build bit for bit;
//   sinf cosf tanf acosf asinf atan2f expf logf powf : the reference build gets these from
//                  glibc, whose results are NOT correctly rounded (up to ~0.8 ulp), so
//                  "a good libm" is not enough to reproduce its pixels: a last-bit
//                  difference moves a scattered direction, and a few such samples per
//                  million flip a hit/miss decision and change a pixel by 1e-2.  They are
//                  therefore restated operation by operation from glibc 2.35's algorithms
//                  (csrc/rl_glibc_math.h, verified exhaustively against the host libm by
//                  tools/check_glibc_math.cc) and run on the device's fp64/fp32 units.
//   fmodf(x, 1)  : exact by definition.
// RAYLIB_OCML_MATH=1 at build time selects the platform's OCML functions instead (faster,
// <= 1-2 ulp from the above; used to price the exact versions).
#pragma once

#include <hip/hip_runtime.h>
#include "rl_glibc_math.h"

namespace rl { namespace rtm {

#if defined(RAYLIB_OCML_MATH)
__device__ __forceinline__ float sin_(float x)  { return sinf(x); }
__device__ __forceinline__ float cos_(float x)  { return cosf(x); }
__device__ __forceinline__ float tan_(float x)  { return tanf(x); }
__device__ __forceinline__ float acos_(float x) { return acosf(x); }
__device__ __forceinline__ float asin_(float x) { return asinf(x); }
__device__ __forceinline__ float atan2_(float y, float x) { return atan2f(y, x); }
__device__ __forceinline__ float atan_(float x) { return atanf(x); }
__device__ __forceinline__ float exp_(float x)  { return expf(x); }
__device__ __forceinline__ float log_(float x)  { return logf(x); }
__device__ __forceinline__ float pow_(float x, float y) { return powf(x, y); }
__device__ __forceinline__ void sincos_(float x, float* s, float* c) { *s = sinf(x); *c = cosf(x); }
__device__ __forceinline__ void sincos_signs_(float x, bool* sn, bool* cn) { *sn = sinf(x) < 0.0f || (sinf(x) == 0.0f && signbit(sinf(x))); *cn = signbit(cosf(x)); }
#else
// Real calls, not inlined: the megakernel evaluates ~40 transcendentals per bounce at ~20 call sites;
// inlined, k_trace was 72 KB of code against a 64 KB instruction cache shared by two CUs.
// Except expf and logf (round 2): they are the short ones (~80 instructions with their table in LDS) and the frequent ones (14 of the
// ~20 transcendentals of a microfacet scattering event), and a call is not free -- the callee starts with s_waitcnt vmcnt(0) lgkmcnt(0)
// because it cannot know what the caller has in flight, plus the jump, the return and the argument moves.  Measured on the Cornell
// frame: exp + log inline 21.02 -> 20.05 ms; + acos 20.06; + pow 20.7; everything inline 20.8 (RL_MATH_OUTLINE_EXPLOG=1 restores the calls).
#ifndef RL_MATH_INLINE
#define RL_MATH_CALL __device__ __noinline__
#else
#define RL_MATH_CALL __device__ __forceinline__
#endif
RL_MATH_CALL float sin_(float x)  { return rlm::sinf_(x); }
RL_MATH_CALL float cos_(float x)  { return rlm::cosf_(x); }
#ifdef RL_MATH_INLINE_TAN
__device__ __forceinline__ float tan_(float x)  { return rlm::tanf_(x); }
#else
RL_MATH_CALL float tan_(float x)  { return rlm::tanf_(x); }
#endif
#ifndef RL_MATH_INLINE_ACOS
#define RL_MATH_INLINE_ACOS 1   /* round 3, with Erf / ErfInv: rl_render.hip */
#endif
#if RL_MATH_INLINE_ACOS
__device__ __forceinline__ float acos_(float x) { return rlm::acosf_(x); }
#else
RL_MATH_CALL float acos_(float x) { return rlm::acosf_(x); }
#endif
RL_MATH_CALL float asin_(float x) { return rlm::asinf_(x); }
RL_MATH_CALL float atan2_(float y, float x) { return rlm::atan2f_(y, x); }
RL_MATH_CALL float atan_(float x) { return rlm::atanf_(x); }
#ifndef RL_MATH_OUTLINE_EXPLOG
__device__ __forceinline__ float exp_(float x)  { return rlm::expf_(x); }
__device__ __forceinline__ float log_(float x)  { return rlm::logf_(x); }
#else
RL_MATH_CALL float exp_(float x)  { return rlm::expf_(x); }
RL_MATH_CALL float log_(float x)  { return rlm::logf_(x); }
#endif
#ifdef RL_MATH_INLINE_POW
__device__ __forceinline__ float pow_(float x, float y) { return rlm::powf_(x, y); }
#else
RL_MATH_CALL float pow_(float x, float y) { return rlm::powf_(x, y); }
#endif
// sinf(x) and cosf(x) of one argument share their range reduction (each result is the separate call's)
RL_MATH_CALL float2 sincos2_(float x) { float s, c; rlm::sincosf_both(x, &s, &c); return make_float2(s, c); }
__device__ __forceinline__ void sincos_(float x, float* s, float* c) { const float2 r = sincos2_(x); *s = r.x; *c = r.y; }
// sign bits of sinf(x), cosf(x) for 0 <= x < 120 without evaluating them
__device__ __forceinline__ void sincos_signs_(float x, bool* sn, bool* cn) { rlm::sincosf_signs(x, sn, cn); }
#endif
__device__ __forceinline__ float fmod1_(float x) { return fmodf(x, 1.0f); }

// ---- 1.0f / x and sqrtf(x), correctly rounded, in fewer issue cycles ---------------------------------------------------------------------
// The compiler's IEEE expansions are long: 1.0f / x = v_div_scale x 2, v_rcp, five fma / mul, v_div_fmas, v_div_fixup = 36 VALU issue cycles per
// wave64; sqrtf = v_sqrt, both neighbours tried with an fma residual each, compares, selects, scaling for tiny inputs = 57 (tools/valu_calib.hip's
// cost table; tools/static_profile.py finds 23 sqrtf sites = 8 % of the Cornell kernel's static issue cycles).  Both kernels sit on the VALU issue
// port, and most divisions of a scattering event are reciprocals: normalize, 1 / tan, 1 / (1 + ...).
//   rcp1_:  r0 = v_rcp_f32(x) (1 ulp); r = fma(fma(-x, r0, 1), r0, r0)                           -- for 2^-126 <= |x| < 2^126
//   sqrt_:  y = v_rsq_f32(x); g = x * y; h = y / 2; g = fma(fma(-g, g, x), h, g)                 -- for 2^-101 <= x <= FLT_MAX
// are the correctly rounded results for EVERY input of those ranges: tools/verify_fastmath.hip compares all 2^32 bit patterns with the compiler's
// expansions on the device (0 mismatches inside the ranges; outside -- zeros, denormals, the tiny inputs whose residual would underflow, infinities,
// NaN, negative radicands -- the expansion itself runs), and RaylibAMD_VerifyExactMath repeats that sweep inside the product library
// (tests/test_math_exact.py).  20 and 22 issue cycles instead of 36 and 57.  The sequences live in rl_glibc_math.h (its acosf / asinf take square roots too).
__device__ __forceinline__ float rcp1_(float x) { return rlm::rcp1_(x); }
__device__ __forceinline__ float sqrt_(float x) { return rlm::sqrtf_(x); }

// ---- a / b when y = RN(1 / b) is in hand (a divisor fixed per launch, per triangle ...) ---------------------------------------------------------
//   q0 = a * y;  q = fma(fma(-b, q0, a), y, q0)
// is the correctly rounded quotient for EVERY pair of significands: tools/verify_fastdiv.hip walks all 2^23 x 2^23 pairs on the device against the
// compiler's IEEE expansion (0 mismatches; with the raw v_rcp_f32 for y, or a y that is not the correctly rounded reciprocal, it is not).  6 issue
// cycles instead of 36.  Every step scales exactly with a power of two, so the significand proof carries to all operands that keep the intermediate
// values in the normal range -- the conditions v_div_scale_f32 tests for, and which the CALLER has to guarantee:
//   2^-126 <= |b| <= 2^126 (y normal),  |a| >= 2^-102 (the residual a - q0 b, a multiple of 2^(exponent(a) - 47), is exact: RaylibAMD_VerifyExactMath's sweep fails at 2^-103),  2^-126 <= |a / b| < 2^127.
// a = +0 gives +0 like the division (for positive b); other zeros, infinities and NaN are the caller's to keep away or to not care about.
__device__ __forceinline__ float div_by_(float a, float b, float y) { const float q0 = a * y; return __builtin_fmaf(__builtin_fmaf(-b, q0, a), y, q0); }

}} // namespace rl::rtm
