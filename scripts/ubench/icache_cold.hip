// Cost of executing straight-line code for the first time in a kernel (cold instruction cache) on MI355X.
//   hipcc -O3 --offload-arch=gfx950 icache_cold.hip -o icache_cold && ./icache_cold
// Each wave runs the same 4 KB / 16 KB block of scalar adds twice and timestamps both passes (100 MHz clock):
// pass 2 - pass 1 = instruction fetch cost of the block.  Launched repeatedly behind another kernel, like the
// alternating PCG kernels.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__global__ void other(double* p) { p[threadIdx.x] += 1.0; }

__device__ __forceinline__ unsigned long long now()
{
	unsigned long long t;
	asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
	return t;
}

template <int KB>
__global__ void code_block(unsigned long long* out)
{
	const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	unsigned long long t[3];
	int acc = 0;
#pragma unroll 1
	for (int pass = 0; pass < 2; pass++)
	{
		t[pass] = now();
		if (KB == 4) asm volatile(".rept 1024\n\ts_add_u32 %0, %0, 1\n\t.endr" : "+s"(acc));
		else asm volatile(".rept 4096\n\ts_add_u32 %0, %0, 1\n\t.endr" : "+s"(acc));
	}
	t[2] = now();
	if ((threadIdx.x & 63) == 0) { out[wave * 4] = t[0]; out[wave * 4 + 1] = t[1]; out[wave * 4 + 2] = t[2]; out[wave * 4 + 3] = acc; }
}

template <int KB>
void run(int waves)
{
	unsigned long long* out; double* p;
	hipMalloc(&out, (size_t)waves * 32); hipMalloc(&p, 4096);
	std::vector<unsigned long long> h((size_t)waves * 4);
	for (int rep = 0; rep < 4; rep++)
	{
		hipLaunchKernelGGL(other, dim3(1), dim3(64), 0, 0, p);
		hipLaunchKernelGGL(code_block<KB>, dim3(waves / 4), dim3(256), 0, 0, out);
		hipDeviceSynchronize();
	}
	hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
	std::vector<long long> a(waves), b(waves);
	for (int w = 0; w < waves; w++) { a[w] = (long long)(h[w * 4 + 1] - h[w * 4]) * 10; b[w] = (long long)(h[w * 4 + 2] - h[w * 4 + 1]) * 10; }
	std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
	printf("%2d KB block, %4d waves: first pass p50 %5lld p90 %5lld ns | second pass p50 %5lld p90 %5lld ns\n", KB, waves, a[waves / 2], a[waves * 9 / 10], b[waves / 2], b[waves * 9 / 10]);
	hipFree(out); hipFree(p);
}

int main()
{
	for (int waves : { 256, 1024, 2560 }) { run<4>(waves); run<16>(waves); }
	return 0;
}
