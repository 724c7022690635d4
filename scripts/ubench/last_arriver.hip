// Price and correctness of a "last arriver reduces" tail (for the next round: pre-reducing the SpMV's row-sum partials
// per aggregate inside the SpMV kernel, so that the two-level kernel stops re-adding them in every workgroup).
//   hipcc -O3 --offload-arch=gfx950 last_arriver.hip -o last_arriver && ./last_arriver
// G groups of PER workgroups; every workgroup writes 12 doubles, releases, bumps its group's counter; the workgroup that
// sees the count complete adds the PER partials in index order (device-scope loads), stores the sums and resets the
// counter.  The host checks every launch's sums exactly and compares the kernel time with the same kernel without the tail.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int PER = 12, VAL = 12;

template <bool TAIL>
__global__ __launch_bounds__(256) void producer(double* part, double* sum, int* counter, int iter)
{
	__shared__ int last;
	const int wg = blockIdx.x, grp = wg / PER;
	if (threadIdx.x < VAL) part[(size_t)wg * VAL + threadIdx.x] = 1e-3 * iter + wg + 0.125 * threadIdx.x;
	if (!TAIL) return;
	__threadfence();                       // release the partial to the device
	__syncthreads();
	if (threadIdx.x == 0) last = __hip_atomic_fetch_add(&counter[grp], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == PER - 1;
	__syncthreads();
	if (!last) return;
	if (threadIdx.x < VAL)
	{
		double s = 0;
		for (int m = 0; m < PER; m++)
			s += __hip_atomic_load(&part[(size_t)(grp * PER + m) * VAL + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		sum[(size_t)grp * VAL + threadIdx.x] = s;
	}
	if (threadIdx.x == 0) __hip_atomic_store(&counter[grp], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void consumer(const double* sum, double* out, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = sum[i]; }

int main()
{
	const int G = 56, WG = G * PER, launches = 2000;
	double *part, *sum, *out; int* counter;
	CHECK(hipMalloc(&part, sizeof(double) * WG * VAL)); CHECK(hipMalloc(&sum, sizeof(double) * G * VAL)); CHECK(hipMalloc(&out, sizeof(double) * G * VAL));
	CHECK(hipMalloc(&counter, sizeof(int) * G)); CHECK(hipMemset(counter, 0, sizeof(int) * G));
	std::vector<double> h(G * VAL);
	long bad = 0;
	for (int it = 0; it < launches; it++)
	{
		hipLaunchKernelGGL(producer<true>, dim3(WG), dim3(256), 0, 0, part, sum, counter, it);
		hipLaunchKernelGGL(consumer, dim3((G * VAL + 255) / 256), dim3(256), 0, 0, sum, out, G * VAL);
		if (it % 50 == 49)
		{
			CHECK(hipMemcpy(h.data(), out, sizeof(double) * G * VAL, hipMemcpyDeviceToHost));
			for (int g = 0; g < G; g++)
				for (int v = 0; v < VAL; v++)
				{
					double s = 0;
					for (int m = 0; m < PER; m++) s += 1e-3 * it + (g * PER + m) + 0.125 * v;
					bad += h[g * VAL + v] != s;
				}
		}
	}
	CHECK(hipDeviceSynchronize());
	printf("%d launches, %ld wrong sums\n", launches, bad);
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	for (int variant = 0; variant < 2; variant++)
	{
		CHECK(hipEventRecord(e0, 0));
		for (int it = 0; it < 1000; it++)
		{
			if (variant) hipLaunchKernelGGL(producer<true>, dim3(WG), dim3(256), 0, 0, part, sum, counter, it);
			else hipLaunchKernelGGL(producer<false>, dim3(WG), dim3(256), 0, 0, part, sum, counter, it);
		}
		CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
		float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
		printf("%s: %.2f us per launch (back-to-back eager launches)\n", variant ? "with the last-arriver tail" : "partials only            ", ms);
	}
	return 0;
}
