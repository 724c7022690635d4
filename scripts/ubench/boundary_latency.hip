// Memory latency seen by the first loads of a kernel, right after a kernel boundary (MI355X, gfx950).
//   hipcc -O3 --offload-arch=gfx950 boundary_latency.hip -o boundary_latency && ./boundary_latency
// A writer kernel fills a buffer; the reader kernel (launched behind it on the same stream, like consecutive PCG
// kernels) times, per wave, (1) a load of data the writer just produced, (2) a load of data that was last written
// long ago (static, e.g. matrix blocks), (3) a second load of the line fetched in (1) (L2 / TCP hit), (4) a dependent
// load whose address comes from (1).  100 MHz s_memrealtime ticks -> 10 ns resolution.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void writer(double* fresh, int n, double v)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) fresh[i] = v + i;
}

__device__ __forceinline__ unsigned long long now()
{
	unsigned long long t;
	asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
	return t;
}

__global__ void reader(const double* fresh, const double* stat, const int* chase, unsigned long long* out, double* sink, int n)
{
	const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
	const int i = (wave * 64 + lane) % n;
	const unsigned long long t0 = now();
	const double a = fresh[i];
	const unsigned long long t1 = now();
	const double b = stat[i];
	const unsigned long long t2 = now();
	const double c = fresh[i ^ 1];
	const unsigned long long t3 = now();
	const int j = chase[(int)a % n];
	const unsigned long long t4 = now();
	const double d = stat[(j + 4096) % n];
	const unsigned long long t5 = now();
	if (a + b + c + d == -1.0) sink[0] = a;
	if (lane == 0)
	{
		out[wave * 6 + 0] = t0; out[wave * 6 + 1] = t1; out[wave * 6 + 2] = t2; out[wave * 6 + 3] = t3; out[wave * 6 + 4] = t4; out[wave * 6 + 5] = t5;
	}
}

int main()
{
	const int n = 1 << 20;
	double *fresh, *stat, *sink; int* chase; unsigned long long* out;
	CHECK(hipMalloc(&fresh, n * 8)); CHECK(hipMalloc(&stat, n * 8)); CHECK(hipMalloc(&sink, 8)); CHECK(hipMalloc(&chase, n * 4));
	std::vector<int> hc(n);
	for (int i = 0; i < n; i++) hc[i] = (int)(((long long)i * 7919 + 12345) % n);
	CHECK(hipMemcpy(chase, hc.data(), n * 4, hipMemcpyHostToDevice));
	CHECK(hipMemset(stat, 0, n * 8));
	for (int waves : { 64, 1024, 4096 })
	{
		const int blocks = waves / 4;
		CHECK(hipMalloc(&out, (size_t)waves * 6 * 8));
		std::vector<unsigned long long> h((size_t)waves * 6);
		for (int rep = 0; rep < 3; rep++)
		{
			hipLaunchKernelGGL(writer, dim3(n / 256), dim3(256), 0, 0, fresh, n, (double)rep);
			hipLaunchKernelGGL(reader, dim3(blocks), dim3(256), 0, 0, fresh, stat, chase, out, sink, n);
			CHECK(hipDeviceSynchronize());
		}
		CHECK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
		const char* names[5] = { "fresh (written by previous kernel)", "static", "same line again", "dependent (chase[], static)", "dependent 2 (static)" };
		printf("%d waves:\n", waves);
		for (int s = 0; s < 5; s++)
		{
			std::vector<long long> d(waves);
			for (int w = 0; w < waves; w++) d[w] = (long long)(h[w * 6 + s + 1] - h[w * 6 + s]) * 10;
			std::sort(d.begin(), d.end());
			printf("  %-36s min %5lld  p50 %5lld  p90 %5lld  max %5lld ns\n", names[s], d[0], d[waves / 2], d[waves * 9 / 10], d[waves - 1]);
		}
		CHECK(hipFree(out));
	}
	return 0;
}
