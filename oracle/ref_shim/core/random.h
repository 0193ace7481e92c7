// TEST INFRASTRUCTURE -- determinism overlay used only when oracle/_ref is built.
//
// This header is placed on the include path IN FRONT OF /root/reference/raylib,
// so that every reference translation unit that says #include "core/random.h"
// (random.cc, bvh.cc, camera.h, material.h, renderer.cc) gets this file instead
// of the reference's raylib/core/random.h.  Nothing else of the reference is
// replaced: its random.cc (sphere / disk / hemisphere samplers, Random()) is
// compiled unmodified against this header.
//
// Why it exists: the reference's `class RNG` fills a table from
// std::random_device (reference raylib/core/random.h:17-29) and has no seed,
// so the unmodified reference never renders the same image twice.
// Here `RNG::Peek()` returns the next value of the per-sample stream defined
// in include/raylib_amd_rng.h; the four thread_local tables of the reference
// (renderer.cc:211, random.cc:5,37,44) therefore all read ONE stream in
// program order.  The stream is selected by the driver (oracle/ref_glue.cc)
// with RefRngSelect() before each camera sample / before a BVH build.
#pragma once

#include "raylib_types.h"
#include "core/int_types.h"
#include "core/vec3.h"
#include "raylib_amd_rng.h"

#include <random>
#include <vector>
#include <algorithm>

// One stream per thread; the driver re-keys it per (pixel, sample).
extern thread_local RaylibRngStream g_refRngStream;
extern thread_local uint64_t g_refRngDraws; // how many values were consumed (test introspection)

inline void RefRngSelect(uint64_t seed, uint32_t pixelIndex, uint32_t sampleIndex)
{
	g_refRngStream = raylib_rng_begin(seed, pixelIndex, sampleIndex);
	g_refRngDraws = 0;
}

// Same public surface as the reference class (ctor(uint32), Seek, Peek).
class RNG
{
public:
	explicit RNG(uint32 /*nSamples: table size, not observable in results*/) {}
	inline void Seek(int32) {}
	inline float Peek()
	{
		++g_refRngDraws;
		return raylib_rng_next_float(&g_refRngStream);
	}
};

// Declarations the reference's random.cc defines (reference core/random.h:68-73).
RAYLIB_API float Random();
RAYLIB_API vec3 RandomInUnitSphere();
vec3 RandomInHemisphere(const vec3& axis);
vec3 RandomInUnitDisk();
vec3 RandomInCosineHemisphere();
