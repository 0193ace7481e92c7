// Exhaustive comparison, on the device, of cheap exact-candidate sequences for 1.0f / x and sqrtf(x) with the IEEE results the compiler's own
// expansions give (v_div_scale / v_rcp / fma chain / v_div_fmas / v_div_fixup: 36 issue cycles; v_sqrt + two-sided correction + scaling: 57).
// Every one of the 2^32 bit patterns is tried; per candidate the program prints the number of mismatching inputs and the smallest / largest
// mismatching |x| (as bits), from which the guard of csrc/rl_math.h rcp1_ / sqrt_ is read off.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/verify_fastmath.hip -o tools/verify_fastmath && tools/verify_fastmath
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <string.h>

__device__ __forceinline__ float rcpA(float x) { float r0 = __builtin_amdgcn_rcpf(x); float e = __builtin_fmaf(-x, r0, 1.0f); return __builtin_fmaf(e, r0, r0); }
__device__ __forceinline__ float rcpB(float x) { float r = rcpA(x); float e = __builtin_fmaf(-x, r, 1.0f); return __builtin_fmaf(e, r, r); }
__device__ __forceinline__ float sqrtS1(float x)
{
	float y = __builtin_amdgcn_rsqf(x);
	float g = x * y, h = 0.5f * y;
	float r = __builtin_fmaf(-h, g, 0.5f);
	g = __builtin_fmaf(g, r, g); h = __builtin_fmaf(h, r, h);
	float d = __builtin_fmaf(-g, g, x);
	return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float sqrtS3(float x)
{
	float y = __builtin_amdgcn_rsqf(x);
	float g = x * y, h = 0.5f * y;
	float d = __builtin_fmaf(-g, g, x);
	return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float sqrtS4(float x)   // S3 + one more residual step
{
	float y = __builtin_amdgcn_rsqf(x);
	float g = x * y, h = 0.5f * y;
	float d = __builtin_fmaf(-g, g, x);
	g = __builtin_fmaf(d, h, g);
	d = __builtin_fmaf(-g, g, x);
	return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float sqrtS5(float x)   // hardware sqrt (1 ulp) + one residual step with h from rsq
{
	float g = __builtin_amdgcn_sqrtf(x);
	float h = 0.5f * __builtin_amdgcn_rsqf(x);
	float d = __builtin_fmaf(-g, g, x);
	return __builtin_fmaf(d, h, g);
}

struct Result { unsigned long long bad; unsigned int lo, hi, first; unsigned int n; unsigned int ex[16][3]; unsigned int byExp[256]; };

template <int WHICH>
__global__ void __launch_bounds__(256) k_sweep(Result* res)
{
	const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x, n = (unsigned long long)gridDim.x * blockDim.x;
	unsigned long long bad = 0; unsigned int lo = 0xffffffffu, hi = 0u;
	for (unsigned long long b = tid; b < (1ull << 32); b += n) {
		const unsigned int bits = (unsigned int)b;
		const float x = __uint_as_float(bits);
		float want, got;
		if (WHICH < 2) { want = 1.0f / x; got = WHICH == 0 ? rcpA(x) : rcpB(x); }
		else { want = __builtin_sqrtf(x); got = WHICH == 2 ? sqrtS1(x) : WHICH == 3 ? sqrtS3(x) : WHICH == 4 ? sqrtS4(x) : sqrtS5(x); }
		const bool same = __float_as_uint(want) == __float_as_uint(got) || (want != want && got != got);
		if (!same) { ++bad; const unsigned int a = bits & 0x7fffffffu; lo = min(lo, a); hi = max(hi, a); }
	}
	if (bad) { atomicAdd(&res->bad, bad); atomicMin(&res->lo, lo); atomicMax(&res->hi, hi); }
}

// the same, restricted to |x| in [loBits, hiBits] (the guard's range): must print 0
template <int WHICH>
__global__ void __launch_bounds__(256) k_range(Result* res, unsigned int loBits, unsigned int hiBits)
{
	const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x, n = (unsigned long long)gridDim.x * blockDim.x;
	unsigned long long bad = 0; unsigned int first = 0xffffffffu;
	for (unsigned long long b = tid; b < (1ull << 32); b += n) {
		const unsigned int bits = (unsigned int)b, a = bits & 0x7fffffffu;
		if (a < loBits || a > hiBits) continue;
		if (WHICH >= 2 && (bits >> 31)) continue;
		const float x = __uint_as_float(bits);
		float want, got;
		if (WHICH < 2) { want = 1.0f / x; got = WHICH == 0 ? rcpA(x) : rcpB(x); }
		else { want = __builtin_sqrtf(x); got = WHICH == 2 ? sqrtS1(x) : WHICH == 3 ? sqrtS3(x) : WHICH == 4 ? sqrtS4(x) : sqrtS5(x); }
		if (__float_as_uint(want) != __float_as_uint(got)) {
			++bad; first = min(first, bits);
			atomicAdd(&res->byExp[(bits >> 23) & 255u], 1u);
			if ((bits & 0x3ffu) == 0x155u) { const unsigned int k = atomicAdd(&res->n, 1u); if (k < 16u) { res->ex[k][0] = bits; res->ex[k][1] = __float_as_uint(want); res->ex[k][2] = __float_as_uint(got); } }
		}
	}
	if (bad) { atomicAdd(&res->bad, bad); atomicMin(&res->first, first); }
}

template <int WHICH> static void Run(const char* name, unsigned int loBits, unsigned int hiBits)
{
	Result* d; hipMalloc(&d, sizeof(Result));
	Result h; memset(&h, 0, sizeof(h)); h.lo = 0xffffffffu; h.first = 0xffffffffu;
	hipMemcpy(d, &h, sizeof(h), hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k_sweep<WHICH>, dim3(4096), dim3(256), 0, 0, d);
	hipDeviceSynchronize();
	hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
	printf("%-8s all 2^32 inputs: %llu mismatches", name, h.bad);
	if (h.bad) { float a, b; memcpy(&a, &h.lo, 4); memcpy(&b, &h.hi, 4); printf(", |x| from 0x%08x (%g) to 0x%08x (%g)", h.lo, a, h.hi, b); }
	Result g; memset(&g, 0, sizeof(g)); g.lo = 0xffffffffu; g.first = 0xffffffffu;
	hipMemcpy(d, &g, sizeof(g), hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k_range<WHICH>, dim3(4096), dim3(256), 0, 0, d, loBits, hiBits);
	hipDeviceSynchronize();
	hipMemcpy(&g, d, sizeof(g), hipMemcpyDeviceToHost);
	printf(" | inside the guard [0x%08x, 0x%08x]: %llu mismatches", loBits, hiBits, g.bad);
	if (g.bad) printf(" (first 0x%08x)", g.first);
	printf("\n");
	if (g.bad) {
		printf("   by exponent:"); for (int e = 0; e < 256; ++e) if (g.byExp[e]) printf(" %d:%u", e, g.byExp[e]); printf("\n");
		for (unsigned k = 0; k < (g.n < 16u ? g.n : 16u); ++k) { float x, w, o; memcpy(&x, &g.ex[k][0], 4); memcpy(&w, &g.ex[k][1], 4); memcpy(&o, &g.ex[k][2], 4); printf("   x 0x%08x (%.9g): want 0x%08x got 0x%08x (%+d ulp)\n", g.ex[k][0], x, g.ex[k][1], g.ex[k][2], (int)(g.ex[k][2] - g.ex[k][1])); }
	}
	hipFree(d);
}

int main()
{
	// guards: reciprocal |x| in [2^-126, 2^126); square root x in [2^-126, FLT_MAX]
	Run<0>("rcpA", 0x00800000u, 0x7e7fffffu);
	Run<1>("rcpB", 0x00800000u, 0x7e7fffffu);
	Run<2>("sqrtS1", 0x00800000u, 0x7f7fffffu);
	Run<3>("sqrtS3", 0x00800000u, 0x7f7fffffu);
	Run<4>("sqrtS4", 0x00800000u, 0x7f7fffffu);
	Run<5>("sqrtS5", 0x00800000u, 0x7f7fffffu);
	// ... and the pair the product uses, with the product's guards (csrc/rl_glibc_math.h rcp1_ / sqrtf_): both lines must end in "0 mismatches"
	Run<0>("rcp1_", 0x00800000u, 0x7e7fffffu);
	Run<3>("sqrtf_", 0x0d000000u, 0x7f7fffffu);
	return 0;
}
