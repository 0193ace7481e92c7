// What does the vector memory path charge a wave for one record PER LANE from scattered addresses, by the SHAPE of the fetch: how many instructions, how wide
// (tools/vmem_gather_bench.hip found one 16-byte lane-load per clock and CU whatever the table size; round 5 asks whether that is per byte or per instruction:
// the 8-wide node of 128 bytes = eight dwordx4 loads made the tree walk no faster although it issues a quarter fewer VALU cycles per step).
// Dependent chains as in a tree walk (the next record's index comes out of the record just read), 16 waves per CU, table of 4 MB (L2-resident).
// build: hipcc --offload-arch=gfx950 -O3 tools/vmem_width_bench.hip -o tools/vmem_width_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// BYTES per record fetched as BYTES / W loads of W bytes each (W = 4, 8, 16); records are 128-byte aligned slots
template <int BYTES, int W>
__global__ void __launch_bounds__(256, 4) k_fetch(const char* __restrict__ table, uint32_t mask, int iters, uint32_t* __restrict__ out)
{
	uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u & mask;
	uint32_t acc = 0;
	for (int it = 0; it < iters; ++it) {
		const char* p = table + (size_t)idx * 128u;
		uint32_t x = 0;
		// (inline assembly: the compiler's load vectoriser would merge adjacent narrow loads back into dwordx4)
		if constexpr (W == 16) { _Pragma("unroll") for (int k = 0; k < BYTES / 16; ++k) { const uint4 v = ((const uint4*)p)[k]; x += v.x ^ v.y ^ v.z ^ v.w; } }
		else if constexpr (W == 8) {
			uint2 v[BYTES / 8];
			_Pragma("unroll") for (int k = 0; k < BYTES / 8; ++k) asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=v"(v[k]) : "v"(p), "n"(k * 8));
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			_Pragma("unroll") for (int k = 0; k < BYTES / 8; ++k) { asm volatile("" : "+v"(v[k])); x += v[k].x ^ v[k].y; }
		} else {
			uint32_t v[BYTES / 4];
			_Pragma("unroll") for (int k = 0; k < BYTES / 4; ++k) asm volatile("global_load_dword %0, %1, off offset:%2" : "=v"(v[k]) : "v"(p), "n"(k * 4));
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			_Pragma("unroll") for (int k = 0; k < BYTES / 4; ++k) { asm volatile("" : "+v"(v[k])); x += v[k]; }
		}
		acc += x;
		idx = (x + acc) & mask;
	}
	out[blockIdx.x * 256u + threadIdx.x] = acc + idx;
}

// the same fetch (dwordx4 loads) with `SHARE` neighbouring lanes on ONE record: what lanes that walk the same node (the top of a tree) cost
template <int BYTES, int SHARE>
__global__ void __launch_bounds__(256, 4) k_shared(const char* __restrict__ table, uint32_t mask, int iters, uint32_t* __restrict__ out)
{
	uint32_t idx = ((blockIdx.x * 256u + threadIdx.x) / SHARE) * 2654435761u & mask;
	uint32_t acc = 0;
	for (int it = 0; it < iters; ++it) {
		const char* p = table + (size_t)idx * 128u;
		uint32_t x = 0;
		_Pragma("unroll") for (int k = 0; k < BYTES / 16; ++k) { const uint4 v = ((const uint4*)p)[k]; x += v.x ^ v.y ^ v.z ^ v.w; }
		acc += x;
		idx = (x + acc) & mask;   // (lanes that start on one record stay together: same data, same chain)
	}
	out[blockIdx.x * 256u + threadIdx.x] = acc + idx;
}
template <int BYTES, int SHARE>
static void run_shared(const char* d, uint32_t n, uint32_t* out)
{
	const int iters = 2000;
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	float ms = 0;
	for (int rep = 0; rep < 2; ++rep) {
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL((k_shared<BYTES, SHARE>), dim3(1024), dim3(256), 0, 0, d, n - 1, iters, out);
		CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
		CHECK(hipEventElapsedTime(&ms, e0, e1));
	}
	const double recs = 1024.0 * 256 * iters;
	hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
	printf("%3d bytes as %2d x 16-byte loads, %2d lanes per record: %7.2f ms  %5.2f clocks per lane-record and CU (at 2.1 GHz)  %5.2f per lane-load\n", BYTES, BYTES / 16, SHARE, ms,
	       ms * 1e-3 * 2.1e9 * prop.multiProcessorCount / recs, ms * 1e-3 * 2.1e9 * prop.multiProcessorCount / recs / (BYTES / 16));
}

template <int BYTES, int W>
static void run(const char* d, uint32_t n, uint32_t* out)
{
	const int iters = 2000;
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	float ms = 0;
	for (int rep = 0; rep < 2; ++rep) {
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL((k_fetch<BYTES, W>), dim3(1024), dim3(256), 0, 0, d, n - 1, iters, out);
		CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
		CHECK(hipEventElapsedTime(&ms, e0, e1));
	}
	const double recs = 1024.0 * 256 * iters;
	hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
	const double clk = 2.1e9;   // nominal sustained clock: the figure is for comparing shapes, not an absolute
	printf("%3d bytes as %2d x %2d-byte loads: %7.2f ms  %6.1f G records/s  %5.2f clocks per record and CU (at 2.1 GHz)  %5.2f per load\n", BYTES, BYTES / W, W, ms, recs / ms / 1e6,
	       ms * 1e-3 * clk * prop.multiProcessorCount / recs, ms * 1e-3 * clk * prop.multiProcessorCount / recs / (BYTES / W));
}

int main()
{
	const uint32_t n = 32768;   // x 128 bytes = 4 MB
	std::vector<uint32_t> h((size_t)n * 32);
	uint32_t s = 12345u;
	for (uint32_t& w : h) { s = s * 1664525u + 1013904223u; w = s >> 7; }
	char* d; uint32_t* out;
	CHECK(hipMalloc(&d, (size_t)n * 128)); CHECK(hipMalloc(&out, 1024 * 256 * 4));
	CHECK(hipMemcpy(d, h.data(), (size_t)n * 128, hipMemcpyHostToDevice));
	run<16, 16>(d, n, out); run<32, 16>(d, n, out); run<48, 16>(d, n, out); run<64, 16>(d, n, out); run<80, 16>(d, n, out); run<96, 16>(d, n, out); run<128, 16>(d, n, out);
	run<16, 8>(d, n, out); run<48, 8>(d, n, out); run<64, 8>(d, n, out); run<128, 8>(d, n, out);
	run<16, 4>(d, n, out); run<32, 4>(d, n, out); run<64, 4>(d, n, out);
	run_shared<128, 1>(d, n, out); run_shared<128, 2>(d, n, out); run_shared<128, 4>(d, n, out); run_shared<128, 16>(d, n, out); run_shared<128, 64>(d, n, out);
	return 0;
}
