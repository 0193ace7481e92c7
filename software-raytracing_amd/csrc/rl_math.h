// Device math for the megakernel.  Every transcendental the reference's hot path
// calls (reference render/material.cc, render/brdf.h, core/random.cc,
// render/texture.cc, render/renderer.cc:159-181) goes through one of these
// wrappers, so that the numerical policy lives in one place:
//   + - * / sqrt : IEEE binary32, correctly rounded (hipcc default for HIP), and the
//                  file is compiled with -ffp-contract=off (no FMA contraction), so
//                  these match the x86-64 reference build bit for bit;
//   transcendentals: see each function.
#pragma once

#include <hip/hip_runtime.h>

namespace rl { namespace rtm {

__device__ __forceinline__ float rsqrt_exact_div(float x) { return 1.0f / sqrtf(x); }

__device__ __forceinline__ float sin_(float x)  { return sinf(x); }
__device__ __forceinline__ float cos_(float x)  { return cosf(x); }
__device__ __forceinline__ float tan_(float x)  { return tanf(x); }
__device__ __forceinline__ float acos_(float x) { return acosf(x); }
__device__ __forceinline__ float asin_(float x) { return asinf(x); }
__device__ __forceinline__ float atan2_(float y, float x) { return atan2f(y, x); }
__device__ __forceinline__ float exp_(float x)  { return expf(x); }
__device__ __forceinline__ float log_(float x)  { return logf(x); }
__device__ __forceinline__ float pow_(float x, float y) { return powf(x, y); }
__device__ __forceinline__ float fmod1_(float x) { return fmodf(x, 1.0f); }   // exact in any correct implementation

}} // namespace rl::rtm
