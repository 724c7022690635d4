// Micro-benchmark: cost of a dependent kernel in a hipGraph on MI355X as a function of kernarg size, grid size and
// of reading a value that the previous kernel wrote with device-scope atomics.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Big { double* p[60]; };
__global__ void k_small(double* p) { if (p == nullptr) p[0] = 1; }
__global__ void k_big(Big b) { if (b.p[0] == nullptr) b.p[1][0] = 1; }
__global__ void k_chain(double* slots, double* out)
{
	// read 16 slots written by the previous launch's atomics, wave-reduce, one atomic back
	const int lane = threadIdx.x & 63;
	double v = lane < 16 ? slots[lane] : 0;
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
	if (threadIdx.x == 0) atomicAdd(&slots[blockIdx.x % 16], v * 1e-30);
	if (v == 12345.678) out[0] = v;
}
template <class F> double timeit(hipStream_t s, int reps, F fn)
{
	hipGraph_t g; hipGraphExec_t e; hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
	for (int i = 0; i < reps; i++) fn();
	hipStreamEndCapture(s, &g); hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
	hipGraphLaunch(e, s); hipEventRecord(a, s); hipGraphLaunch(e, s); hipEventRecord(b, s); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b); return ms * 1e3 / reps;
}
int main()
{
	hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	double *d, *slots; CK(hipMalloc(&d, 1 << 20)); CK(hipMalloc(&slots, 1024)); CK(hipMemset(slots, 0, 1024));
	Big big; for (auto& p : big.p) p = d;
	for (int grid : { 1, 84, 333, 2048 })
	{
		printf("grid %4d x 256: small-arg %.2f us  big-arg(480B) %.2f us  atomic-chain %.2f us   | eager small %.2f us\n", grid,
			timeit(s, 200, [&] { hipLaunchKernelGGL(k_small, dim3(grid), dim3(256), 0, s, d); }),
			timeit(s, 200, [&] { hipLaunchKernelGGL(k_big, dim3(grid), dim3(256), 0, s, big); }),
			timeit(s, 200, [&] { hipLaunchKernelGGL(k_chain, dim3(grid), dim3(256), 0, s, slots, d); }),
			[&] { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, s);
			      for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_small, dim3(grid), dim3(256), 0, s, d);
			      hipEventRecord(b, s); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); return ms * 1e3 / 200; }());
	}
	return 0;
}
