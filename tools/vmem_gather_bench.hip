// What does a wave pay for fetching one 64-byte record PER LANE from scattered addresses (the pool kernel's node fetch)?
//   A  "own":   every lane issues 4 x global_load_dwordx4 on its own record (what k_trace_pool does: 4 L1 accesses per lane and record)
//   B  "quad":  the 4 lanes of a quad fetch the 4 chunks of ONE record with ONE instruction (64 contiguous bytes per quad), 4 instructions for the
//               quad's 4 records, and the chunks reach their owner through LDS (4 x ds_write_b128 + 4 x ds_read_b128)
// Dependent chains as in a tree walk: the next record's index comes out of the record just read.  Prints ns per record and wave, records/s chip-wide.
// build: hipcc --offload-arch=gfx950 -O3 tools/vmem_gather_bench.hip -o tools/vmem_gather_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct alignas(64) Rec { uint32_t w[16]; };

template <int MODE>
__global__ void __launch_bounds__(256, 4) k_gather(const Rec* __restrict__ table, uint32_t mask, int iters, uint32_t* __restrict__ out)
{
	__shared__ uint4 s_x[4][4][64];   // [wave][k][lane]: MODE 1 only (16 KB per workgroup)
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u & mask;
	uint32_t acc = 0;
	for (int it = 0; it < iters; ++it) {
		uint4 a, b, c, d;
		if (MODE == 0) {
			const uint4* p = (const uint4*)(table + idx);
			a = p[0]; b = p[1]; c = p[2]; d = p[3];
		} else {
			uint4 r[4];
			#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const uint32_t ik = (uint32_t)__shfl((int)idx, (int)((lane & ~3u) + k));
				r[k] = ((const uint4*)(table + ik))[lane & 3u];
			}
			#pragma unroll
			for (int k = 0; k < 4; ++k) s_x[wave][k][lane] = r[k];
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
			const uint4* q = &s_x[wave][lane & 3u][lane & ~3u];
			a = q[0]; b = q[1]; c = q[2]; d = q[3];
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
		}
		acc += a.x ^ b.y ^ c.z ^ d.w;
		idx = (a.x + b.x + c.x + d.x + acc) & mask;   // the next record depends on this one
	}
	out[blockIdx.x * 256u + threadIdx.x] = acc + idx;
}

int main(int argc, char** argv)
{
	const int iters = 2000;
	for (uint32_t n : { 256u, 4096u, 65536u, 1u << 20 }) {   // 16 KB (L1), 256 KB, 4 MB (L2), 64 MB (Infinity Cache)
		std::vector<Rec> h(n);
		uint32_t s = 12345u;
		for (Rec& r : h) for (uint32_t& w : r.w) { s = s * 1664525u + 1013904223u; w = s >> 7; }
		Rec* d; uint32_t* out;
		CHECK(hipMalloc(&d, n * sizeof(Rec))); CHECK(hipMalloc(&out, 1024 * 256 * 4));
		CHECK(hipMemcpy(d, h.data(), n * sizeof(Rec), hipMemcpyHostToDevice));
		float ms[2] = { 0, 0 };
		uint32_t sum[2] = { 0, 0 };
		for (int mode = 0; mode < 2; ++mode) {
			hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
			for (int rep = 0; rep < 2; ++rep) {
				CHECK(hipEventRecord(e0));
				if (mode == 0) hipLaunchKernelGGL(k_gather<0>, dim3(1024), dim3(256), 0, 0, d, n - 1, iters, out);
				else hipLaunchKernelGGL(k_gather<1>, dim3(1024), dim3(256), 0, 0, d, n - 1, iters, out);
				CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
				CHECK(hipEventElapsedTime(&ms[mode], e0, e1));
			}
			std::vector<uint32_t> o(1024 * 256);
			CHECK(hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost));
			for (uint32_t v : o) sum[mode] += v;
		}
		const double recs = 1024.0 * 256 * iters;
		printf("%8u records (%6.0f KB): own 4 x dwordx4 %.2f ms = %.1f G records/s | quad-cooperative + LDS %.2f ms = %.1f G records/s | x%.2f  (checksums %s)\n",
		       n, n * 64 / 1024.0, ms[0], recs / ms[0] / 1e6, ms[1], recs / ms[1] / 1e6, ms[0] / ms[1], sum[0] == sum[1] ? "equal" : "DIFFER");
		CHECK(hipFree(d)); CHECK(hipFree(out));
	}
	return 0;
}
