/*
 * raylib_amd_rng.h -- the deterministic random-number contract of the MI355X raylib.
 *
 * The reference renderer has NO seed: every RNG table is filled from
 * std::random_device per thread (reference raylib/core/random.h:17-29,45) and
 * consumed in thread-scheduling order, so two runs never agree.  A parity
 * target needs a stream that is a pure function of (seed, pixel, sample), so
 * this header defines one.  It is shared, verbatim, by
 *   - the HIP megakernel            (software-raytracing_amd/csrc/)
 *   - the CPU oracle restatement    (oracle/oracle.cc)
 *   - the RNG overlay that is put in front of the reference's own sources
 *     when oracle/_ref is built     (oracle/ref_shim/core/random.h)
 *
 * Contract
 *   One PCG32 (XSH-RR 64/32) stream per camera sample, keyed by
 *   (seed, pixelIndex = y*W + x, sampleIndex).  Every random number the
 *   reference would draw for that sample -- from ANY of its four per-thread
 *   tables (renderer.cc:211 "randomsAA", random.cc:5,37,44) -- is the next
 *   value of that one stream, in program order:
 *     [2 jitter draws if sampleIndex >= 1]  (renderer.cc:236-237)
 *     2 lens-disk draws, 1 shutter-time draw (camera.h:46-48, always)
 *     then per bounce: Lambertian 2, Metal 2, Dielectric 1, Microfacet 2,
 *     Mirror / DiffuseLight 0           (material.cc:195-431, material.h:149)
 *   A draw is a float in [0,1): the top 24 bits of the PCG output * 2^-24.
 *   BVH-build axis draws (bvh.cc:43) use the stream RAYLIB_RNG_BUILD_PIXEL.
 */
#ifndef RAYLIB_AMD_RNG_H
#define RAYLIB_AMD_RNG_H

#include <stdint.h>

#if defined(__HIPCC__)
#define RAYLIB_RNG_FN __host__ __device__ static inline
#else
#define RAYLIB_RNG_FN static inline
#endif

#define RAYLIB_RNG_BUILD_PIXEL 0xFFFFFFFFu

typedef struct RaylibRngStream { uint64_t state; } RaylibRngStream;

RAYLIB_RNG_FN uint64_t raylib_rng_mix64(uint64_t z)
{
	/* splitmix64 finaliser */
	z += 0x9E3779B97F4A7C15ull;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

/* mixedSeed = raylib_rng_mix64(seed): the same for every stream of a render, so callers may hoist it. */
RAYLIB_RNG_FN RaylibRngStream raylib_rng_begin_mixed(uint64_t mixedSeed, uint32_t pixelIndex, uint32_t sampleIndex)
{
	RaylibRngStream s;
	uint64_t key = ((uint64_t)pixelIndex << 32) | (uint64_t)sampleIndex;
	s.state = raylib_rng_mix64(mixedSeed ^ key);
	return s;
}

RAYLIB_RNG_FN RaylibRngStream raylib_rng_begin(uint64_t seed, uint32_t pixelIndex, uint32_t sampleIndex)
{
	return raylib_rng_begin_mixed(raylib_rng_mix64(seed), pixelIndex, sampleIndex);
}

RAYLIB_RNG_FN uint32_t raylib_rng_next_u32(RaylibRngStream* s)
{
	uint64_t old = s->state;
#if defined(__HIP_DEVICE_COMPILE__)
	/* same arithmetic; the 64-bit increment is handed to the compiler as two 32-bit immediates it cannot merge, so that it is
	 * rebuilt where it is used (two moves) instead of being kept in a register pair across the megakernel's loops -- where the
	 * pool schedule had no registers left for it and reloaded it from scratch memory at every draw */
	uint32_t inc_lo = 0xF767814Fu, inc_hi = 0x14057B7Eu;
	asm volatile("" : "+v"(inc_lo), "+v"(inc_hi));
	s->state = old * 6364136223846793005ull + (((uint64_t)inc_hi << 32) | inc_lo);
#else
	s->state = old * 6364136223846793005ull + 1442695040888963407ull;
#endif
	uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
	uint32_t rot = (uint32_t)(old >> 59u);
	return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
}

RAYLIB_RNG_FN float raylib_rng_next_float(RaylibRngStream* s)
{
	return (float)(raylib_rng_next_u32(s) >> 8) * (1.0f / 16777216.0f);
}

#endif /* RAYLIB_AMD_RNG_H */
