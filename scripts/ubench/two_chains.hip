// Micro-benchmark (round 5, verdict item 4): do dependent-kernel CHAINS on different streams of one process overlap on MI355X?
// T host threads, one stream each, every thread replays a chain of `len` dependent ~5 us kernels (64 workgroups: a latency-bound PCG
// iteration in miniature) `reps` times -- as hipGraph launches or as eager launches -- and waits for it; wall time for T = 1, 2, 4.
// Perfect overlap: wall(T) == wall(1).  Full serialisation: wall(T) == T x wall(1).
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_spin(double* p, int ticks)
{
	const unsigned long long t0 = wall_clock64();
	while (wall_clock64() - t0 < (unsigned long long)ticks) {}
	if (p == nullptr) p[0] = 1;
}

static double run(int T, int len, int reps, bool graph, int grid, int ticks, bool lowprio_mix)
{
	std::vector<hipStream_t> st(T); std::vector<hipGraphExec_t> ex(T);
	std::vector<double*> buf(T);
	int lo = 0, hi = 0; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
	for (int t = 0; t < T; t++)
	{
		CK(hipStreamCreateWithPriority(&st[t], hipStreamNonBlocking, (lowprio_mix && (t & 1)) ? lo : 0));
		CK(hipMalloc(&buf[t], 4096));
		hipGraph_t g; CK(hipGraphCreate(&g, 0));
		hipGraphNode_t prev = nullptr;
		for (int i = 0; i < len; i++)
		{
			void* args[2] = { &buf[t], &ticks };
			hipKernelNodeParams p = {}; p.func = (void*)k_spin; p.gridDim = dim3(grid); p.blockDim = dim3(256); p.kernelParams = args;
			hipGraphNode_t n; CK(hipGraphAddKernelNode(&n, g, prev ? &prev : nullptr, prev ? 1 : 0, &p)); prev = n;
		}
		CK(hipGraphInstantiate(&ex[t], g, nullptr, nullptr, 0)); CK(hipGraphDestroy(g));
		CK(hipGraphLaunch(ex[t], st[t])); CK(hipStreamSynchronize(st[t]));
	}
	std::atomic<int> ready{ 0 }; std::atomic<bool> go{ false };
	std::vector<std::thread> th;
	for (int t = 0; t < T; t++) th.emplace_back([&, t] {
		CK(hipSetDevice(0));
		ready++; while (!go.load()) {}
		for (int r = 0; r < reps; r++)
		{
			if (graph) CK(hipGraphLaunch(ex[t], st[t]));
			else for (int i = 0; i < len; i++) hipLaunchKernelGGL(k_spin, dim3(grid), dim3(256), 0, st[t], buf[t], ticks);
			CK(hipStreamSynchronize(st[t]));
		}
	});
	while (ready.load() < T) {}
	const auto t0 = std::chrono::steady_clock::now();
	go = true;
	for (auto& x : th) x.join();
	const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	for (int t = 0; t < T; t++) { hipGraphExecDestroy(ex[t]); hipStreamDestroy(st[t]); hipFree(buf[t]); }
	return dt * 1e6 / (reps * len);     // us per kernel of ONE chain
}

int main(int argc, char** argv)
{
	const int len = 64, reps = 40;
	for (int grid : { 64, 666 })
		for (int ticks : { 100, 500 })       // wall_clock64 runs at 100 MHz: 1 us, 5 us
			for (int graph = 1; graph >= 0; graph--)
			{
				printf("grid %3d, %d us kernels, %s:", grid, ticks / 100, graph ? "hipGraph of 64" : "eager launches");
				for (int T : { 1, 2, 4, 8 }) printf("  T=%d %.2f us/kernel", T, run(T, len, reps, graph, grid, ticks, false));
				printf("\n"); fflush(stdout);
			}
	printf("mixed priorities (odd streams low), grid 64, 5 us, graphs:");
	for (int T : { 2, 4 }) printf("  T=%d %.2f us/kernel", T, run(T, len, reps, true, 64, 500, true));
	printf("\n");
	return 0;
}
