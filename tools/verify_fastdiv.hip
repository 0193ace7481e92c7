// Exhaustive check, on the device, of short exact-candidate sequences for a / b given the correctly rounded reciprocal y = RN(1 / b), against
// the compiler's IEEE expansion of a / b (v_div_scale x 2, v_rcp, fma chain, v_div_fmas, v_div_fixup: 36 issue cycles).
//   V5:  q0 = a * y; r0 = fma(-b, q0, a); q1 = fma(r0, y, q0); r1 = fma(-b, q1, a); q2 = fma(r1, y, q1)        (10 issue cycles)
//   V3:  q0, r0, q1 only                                                                                        (6 issue cycles)
//   V5r: V5 with the raw v_rcp_f32 (1 ulp) in place of y
// As long as nothing under- or overflows the result depends on the two SIGNIFICANDS only (every step scales exactly with a power of two), so all
// 2^23 x 2^23 significand pairs of a, b in [1, 2) are the whole proof for operands whose exponents keep a, b, y, a / b and the residuals
// a - q b (multiples of 2^(exponent(a) - 47)) normal.  One thread owns one divisor and walks all numerators; the grid is cut into launches of
// BATCH divisors so that no launch runs for more than about a second.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/verify_fastdiv.hip -o tools/verify_fastdiv && tools/verify_fastdiv [first] [count]
// prints, per variant, the number of mismatching pairs and of divisors with at least one mismatch (and the first few pairs).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

struct Result { unsigned long long bad[3]; unsigned long long badDivisors[3]; unsigned int n; unsigned int ex[16][4]; };

__device__ __forceinline__ float rcpRN(float x) { float r0 = __builtin_amdgcn_rcpf(x); float e = __builtin_fmaf(-x, r0, 1.0f); return __builtin_fmaf(e, r0, r0); }

__global__ void __launch_bounds__(256) k_pairs(Result* res, unsigned int firstDivisor)
{
	const unsigned int mb = firstDivisor + blockIdx.x * blockDim.x + threadIdx.x;
	const float b = __uint_as_float(0x3f800000u | mb);
	const float y = rcpRN(b), yr = __builtin_amdgcn_rcpf(b);
	if (__float_as_uint(y) != __float_as_uint(1.0f / b)) { atomicAdd(&res->bad[0], 1ull << 40); }   // the reciprocal itself (tools/verify_fastmath.hip): never
	unsigned long long bad5 = 0, bad3 = 0, bad5r = 0;
	for (unsigned int ma = 0; ma < (1u << 23); ++ma) {
		const float a = __uint_as_float(0x3f800000u | ma);
		const float want = a / b;
		const float q0 = a * y, r0 = __builtin_fmaf(-b, q0, a), q1 = __builtin_fmaf(r0, y, q0), r1 = __builtin_fmaf(-b, q1, a), q2 = __builtin_fmaf(r1, y, q1);
		const float p0 = a * yr, s0 = __builtin_fmaf(-b, p0, a), p1 = __builtin_fmaf(s0, yr, p0), s1 = __builtin_fmaf(-b, p1, a), p2 = __builtin_fmaf(s1, yr, p1);
		const unsigned int w = __float_as_uint(want);
		if (__float_as_uint(q2) != w) {
			++bad5;
			const unsigned int k = atomicAdd(&res->n, 1u);
			if (k < 16u) { res->ex[k][0] = __float_as_uint(a); res->ex[k][1] = __float_as_uint(b); res->ex[k][2] = w; res->ex[k][3] = __float_as_uint(q2); }
		}
		bad3 += __float_as_uint(q1) != w;
		bad5r += __float_as_uint(p2) != w;
	}
	if (bad5) { atomicAdd(&res->bad[0], bad5); atomicAdd(&res->badDivisors[0], 1ull); }
	if (bad3) { atomicAdd(&res->bad[1], bad3); atomicAdd(&res->badDivisors[1], 1ull); }
	if (bad5r) { atomicAdd(&res->bad[2], bad5r); atomicAdd(&res->badDivisors[2], 1ull); }
}

int main(int argc, char** argv)
{
	const unsigned long long first = argc > 1 ? strtoull(argv[1], 0, 0) : 0ull, count = argc > 2 ? strtoull(argv[2], 0, 0) : (1ull << 23);
	const unsigned int BATCH = 1u << 18;   // 4 waves per SIMD
	Result* d; hipMalloc(&d, sizeof(Result));
	Result h; memset(&h, 0, sizeof(h));
	hipMemcpy(d, &h, sizeof(h), hipMemcpyHostToDevice);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0, 0);
	for (unsigned long long at = first; at < first + count; at += BATCH) {
		hipLaunchKernelGGL(k_pairs, dim3(BATCH / 256), dim3(256), 0, 0, d, (unsigned int)at);
		if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed at divisor %llu\n", at); return 1; }
		if (((at - first) / BATCH) % 4 == 3) { hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost); printf("  ... divisors [%llu, %llu): V5 %llu  V3 %llu  V5r %llu mismatches so far\n", first, at + BATCH, h.bad[0], h.bad[1], h.bad[2]); fflush(stdout); }
	}
	hipEventRecord(e1, 0); hipEventSynchronize(e1);
	float ms = 0; hipEventElapsedTime(&ms, e0, e1);
	hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
	const char* names[3] = { "V5  (y = RN(1/b), two corrections)", "V3  (y = RN(1/b), one correction)", "V5r (raw v_rcp_f32, two corrections)" };
	printf("significand pairs: divisors [%llu, %llu) x 2^23 numerators = %.4g pairs in %.1f s\n", first, first + count, (double)count * 8388608.0, ms * 1e-3);
	for (int v = 0; v < 3; ++v) printf("%-40s %llu mismatching pairs, %llu divisors with a mismatch\n", names[v], h.bad[v], h.badDivisors[v]);
	for (unsigned k = 0; k < (h.n < 16u ? h.n : 16u); ++k) printf("   V5: a 0x%08x b 0x%08x want 0x%08x got 0x%08x\n", h.ex[k][0], h.ex[k][1], h.ex[k][2], h.ex[k][3]);
	return h.bad[0] ? 2 : 0;
}
