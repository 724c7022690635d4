// ba_solver.hip -- host orchestration of one bundle-adjustment problem on one MI355X + the C ABI
// (include/cuba_hip.h).  Behavioural counterpart of class CudaBlockSolver and of the LM loop in
// CudaBundleAdjustmentImpl::optimize (/root/reference/src/cuda_bundle_adjustment.cpp:73-673, 793-857),
// re-organised around landmark-sorted edges and fused kernels (ba_edge.hip, ba_linearize.hip, ba_pcg.hip, ba_coarse.hip).
//
// There is deliberately no CPU fallback: every entry point fails with CUBA_HIP_ERR_NO_DEVICE /
// CUBA_HIP_ERR_RUNTIME when no gfx950 device is usable.

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <atomic>
#include <functional>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/cuba_hip.h"
#include "ba_kernels.hpp"
#include "host_pool.hpp"
#include "ba_structure.hpp"

using namespace cubahip;

namespace
{

struct HipError { hipError_t code; const char* what; const char* file; int line; };

#define HIP_TRY(expr)                                                    \
	do {                                                                 \
		hipError_t err__ = (expr);                                       \
		if (err__ != hipSuccess) throw HipError{ err__, #expr, __FILE__, __LINE__ }; \
	} while (0)

struct StateError { std::string msg; };
struct ArgError { std::string msg; };

template <typename T>
class DevBuf
{
public:
	DevBuf() = default;
	DevBuf(const DevBuf&) = delete;
	DevBuf& operator=(const DevBuf&) = delete;
	~DevBuf() { release(); }
	void release()
	{
		if (ptr_) (void)hipFree(ptr_);
		ptr_ = nullptr; size_ = cap_ = 0;
	}
	void resize(size_t n)
	{
		if (n > cap_)
		{
			release();
			if (n) HIP_TRY(hipMalloc((void**)&ptr_, n * sizeof(T)));
			cap_ = n;
		}
		size_ = n;
	}
	void upload(const std::vector<T>& h, hipStream_t s)
	{
		resize(h.size());
		if (!h.empty()) HIP_TRY(hipMemcpyAsync(ptr_, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
	}
	void uploadRaw(const T* h, size_t n, hipStream_t s)
	{
		resize(n);
		if (n) HIP_TRY(hipMemcpyAsync(ptr_, h, n * sizeof(T), hipMemcpyHostToDevice, s));
	}
	void zero(hipStream_t s) { if (size_) HIP_TRY(hipMemsetAsync(ptr_, 0, size_ * sizeof(T), s)); }
	T* data() const { return ptr_; }
	size_t size() const { return size_; }
private:
	T* ptr_ = nullptr;
	size_t size_ = 0, cap_ = 0;
};

using Clock = std::chrono::steady_clock;

}  // namespace

struct cuba_hip_solver
{
	int device = 0;
	hipStream_t stream = nullptr;
	bool ownStream = false;
	std::string lastError;

	// options
	double pcgTol = sizeof(Scalar) == 8 ? 1e-7 : 1e-4;       // relative M^-1-norm residual; the objective is second-order in the solve error
	int pcgMaxIter = 0;          // 0 = automatic
	int coarseLinear = 1;        // 1: constant + linear coarse functions per aggregate (12 unknowns), 0: constant only (6)
	int pcgAggregate = -1;       // poses per coarse aggregate: -1 automatic, 0 = block-Jacobi only
	// device-side set-up (ba_structure.hip): the edge sort and the whole symbolic structure are built on the GPU; the host
	// pipeline below stays as the independent cross-check ("device_setup" = 0) and for graphs without edges
	bool deviceSetup = true;
	bool devTopology = false;    // the sorted edge arrays / permutation exist on the device only (host copies are stale)
	bool hostTopoValid = false;  // perm / h_lmptr / h_epose / h_spose / h_slm describe the current graph
	static int bitsFor(long long n) { int b = 1; while ((1LL << b) < n + 1) b++; return b; }
	DevBuf<int> d_rawEp, d_rawEl, d_counters, d_tmpI0, d_tmpI1, d_adjRow, d_lowerPtr, d_chunk;
	DevBuf<uint8_t> d_rawDim;
	DevBuf<double> d_rawMeas, d_rawOmega, d_chiCaller;
	DevBuf<uint32_t> d_perm, d_k32a, d_k32b, d_v32a, d_v32b;
	DevBuf<uint64_t> d_k64a, d_k64b, d_v64a, d_v64b;
	DevBuf<unsigned char> d_topoTemp;
	DevBuf<long long> d_pairCount, d_freeCount, d_freeScan;
	// Internal pose order.  The aggregates of the two-level preconditioner are runs of consecutive pose indices and must be
	// pieces of the trajectory (strongly coupled poses): with arbitrary vertex ids (the caller's solver order follows the ids)
	// they are not, and the PCG needs 20 x the iterations (KITTI-00 shape, shuffled ids: 1403 instead of 61 in the last LM
	// iteration).  When most blocks of the caller-order pattern lie far off the diagonal, the poses are renumbered internally
	// by a strongest-neighbour walk over the co-visibility counts (= Schur products per block), which recovers the
	// trajectory; every entry point keeps speaking the caller's order.
	bool poseReorder = true;
	bool reorderActive = false, reorderTried = false;
	std::vector<int> poseNewOfOld, poseOldOfNew;     // free poses only; identity unless reorderActive
	DevBuf<int> d_rawEpCaller, d_poseMap;
	bool mixedPrecision = false; // fp64 library: records + per-edge arithmetic of the pose / block passes in fp32 (sums, reduced system, PCG in fp64)
	bool profile = false;

	// host copy of the problem (solver order) and of the sort permutation
	int Pt = 0, Pf = 0, Lt = 0, Lf = 0, E = 0;
	bool haveGraph = false, haveStructure = false;
	int partLo = 0, partHi = -1; // landmark range [partLo, partHi) this handle evaluates (-1 = all): multi-GPU partition
	std::vector<int> perm;       // sorted position -> caller edge index
	std::vector<int> h_lmptr, h_epose;   // sorted, e_pose without the stereo bit
	RobustKernel rk[2] = { { 0, 0 }, { 0, 0 } };

	// device: state [q | t | Xw] contiguous (push/pop = one copy), edges, structure, system
	DevBuf<Scalar> d_state, d_backup, d_cam;
	// caller-controlled copies of the estimates (cuba_hip_snapshot_state[_slot]): slot -> [q | t | Xw] in the INTERNAL pose order that
	// was in force when the copy was made -- dropped whenever that order changes (applyPoseOrder / resetPoseOrder) or a new graph arrives
	std::map<int, DevBuf<Scalar>> d_snapshots;
	void dropSnapshots() { d_snapshots.clear(); }
	DevBuf<int> d_epose, d_elm, d_lmptr;
	DevBuf<Scalar> d_mu, d_mv, d_mr, d_w, d_perEdge;
	DevBuf<int> d_waveLm, d_bigLm, d_rowptr, d_colind, d_lmNfree, d_adjPtr, d_adjBlk, d_adjCol;
	DevBuf<int2> d_ell;
	DevBuf<long long> d_bigOfs, d_lmPairBase;
	DevBuf<Scalar> d_bigHpl;
	DevBuf<Scalar> d_red;        // [hsc | bsc | bp]
	DevBuf<Scalar> d_parts, d_lmSys, d_lmInv, d_xp, d_xl, d_minv, d_r, d_z, d_p0, d_p1, d_ap, d_rz, d_pq;
	DevBuf<unsigned long long> d_maxdiag;
	DevBuf<int> d_fail, d_iters, d_kbase, d_done, d_ticket;
	DevBuf<Scalar> d_eval;       // {chi2, landmark scale part, pose scale part} of cuba_hip_evaluate_device
	DevBuf<Scalar> d_coarse[3], d_gjPivots, d_rc, d_r2, d_qpart, d_hrow;   // coarse: two work buffers of the inversion + the inverse in use
	DevBuf<float> d_coarse32[2];  // option precond_fp32 (fp64 library): the inverse in use in fp32 [0] + the staging copy an overlapped inversion leaves [1]
	// The inverse the FIRST solve of the previous LM run was given (same damping regime: lambda_0 = tau * max diagonal): it serves the first
	// solve of the next run on this structure, so that no run waits for an in-line inversion -- the fresh one runs on the second stream
	// under that solve like every other.  A preconditioner only changes iteration counts; results stay a deterministic function of the
	// call sequence (and identical when a run is repeated from the same estimate: the cached inverse IS the fresh one then).
	DevBuf<Scalar> d_firstInv; DevBuf<float> d_firstInv32;
	bool firstInvValid = false, firstInvPending = false, coarseFirstReuse = true;
	bool precondFp32 = sizeof(Scalar) == 8;
	bool fp32Inverse() const { return precondFp32 && sizeof(Scalar) == 8; }
	size_t inv32Count() const { const size_t n = (size_t)6 * sys.cl * sys.nc; return n * ((n + 3) & ~(size_t)3); }
	DevBuf<int> d_blkrow, d_odBlocks, d_prodPtr, d_prodEa, d_prodEb, d_prodLm, d_pePtr, d_peEdge;
	DevBuf<int> d_prodBeg, d_prodEnd, d_peBeg, d_peEnd;     // landmark partition built on the device: the sub-ranges of the global lists it walks
	bool localRanges = false;
	DevBuf<Scalar> d_erec;
	DevBuf<int> d_cbI, d_cbJ, d_cbPtr, d_cbBlk;
	DevBuf<Scalar> d_cbWi, d_cbWj;
	std::vector<int> h_rowptr, h_colind;
	// Pinned, device-mapped host block: [0, 1024) the 4*NSLOT result slots the reduction kernels write DIRECTLY (the host
	// reads them after a stream synchronisation: no copy kernel, no copy latency), [1024, 2048) PCG flags written by the
	// last node of every iteration graph, [2048, 4096) staging for the few remaining explicit read-backs.
	Scalar* h_pinned = nullptr;
	Scalar* slotsDev = nullptr;   // device-side address of h_pinned
	int* flagsDev = nullptr;
	Scalar* hostStage() const { return (Scalar*)((char*)h_pinned + 2048); }
	Scalar slot(int i) const { return ((const volatile Scalar*)h_pinned)[i]; }   // device-written: never cached in a register across a wait

	DeviceGraph g;
	DeviceStructure st;
	DeviceSystem sys;

	// one hipGraph = `chunk` PCG iterations (kernel arguments are chunk-local, the device-side
	// kbase counter supplies the offset): replaying it costs one host call instead of 2-3 launches per iteration
	// (graphs are kept per chunk length -- 4, 8, ..., 256 and the exact batch lengths that come back)
	std::map<std::pair<int, const Scalar*>, hipGraphExec_t> pcgGraphs;   // key: chunk length, coarse inverse the kernels read
	bool useGraph = true;
	hipStream_t captureStream = nullptr;   // private stream used only by time_kernels to record timing graphs (the work stream may be the
	                                       // legacy default stream, which cannot be captured)
	hipStream_t capStream()
	{
		if (!captureStream) HIP_TRY(hipStreamCreateWithFlags(&captureStream, hipStreamNonBlocking));
		return captureStream;
	}

	// Graphs are instantiated on a helper thread (~2 us per node: 0.8 ms for the lengths 4 ... 64, which a NEW topology used to pay inside
	// its first solve, plus 0.25 ms per exact batch length): the solve never waits for one -- a batch whose graph is not there yet is
	// enqueued as plain launches of the same kernels with the same arguments (bit-identical results, ~0.7 us more per launch boundary).
	struct GraphJob { int chunk; const Scalar* acinv; DeviceGraph g; DeviceStructure st; DeviceSystem sys; int maxIter; Scalar tol2; uint64_t gen; };
	struct GraphBuilder
	{
		std::thread th;
		std::mutex m;
		std::condition_variable cv;
		std::deque<GraphJob> jobs;
		std::map<std::pair<int, const Scalar*>, hipGraphExec_t> ready;     // finished graphs of generation `gen`
		std::map<std::pair<int, const Scalar*>, int> requested;           // keys queued or being built (generation `gen`)
		std::vector<hipGraphExec_t> trash;                                 // graphs of a structure that is gone: destroyed off the critical path
		uint64_t gen = 0;
		bool stop = false, busy = false;
		std::atomic<int64_t> builds{ 0 };
		std::atomic<double> seconds{ 0.0 };
	} gb;
	void graphWorker()
	{
		(void)hipSetDevice(device);
		std::unique_lock<std::mutex> lk(gb.m);
		for (;;)
		{
			gb.cv.wait(lk, [&] { return gb.stop || !gb.jobs.empty() || !gb.trash.empty(); });
			if (gb.stop) return;
			if (gb.jobs.empty())
			{
				std::vector<hipGraphExec_t> t; t.swap(gb.trash);
				gb.busy = true;
				lk.unlock();
				for (hipGraphExec_t e : t) (void)hipGraphExecDestroy(e);          // (~0.1 ms each)
				lk.lock();
				gb.busy = false;
				gb.cv.notify_all();
				continue;
			}
			GraphJob j = gb.jobs.front(); gb.jobs.pop_front();
			gb.busy = true;
			lk.unlock();
			const auto t0 = Clock::now();
			hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
			bool ok = hipGraphCreate(&graph, 0) == hipSuccess && graph_add_pcg_chunk(graph, j.g, j.st, j.sys, j.chunk, j.maxIter, j.tol2, 1) == hipSuccess &&
				hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
			if (graph) (void)hipGraphDestroy(graph);
			if (!ok) (void)hipGetLastError();
			const double dt = std::chrono::duration<double>(Clock::now() - t0).count();
			lk.lock();
			gb.busy = false;
			if (ok && j.gen == gb.gen) { gb.ready[std::make_pair(j.chunk, j.acinv)] = exec; gb.builds++; gb.seconds.store(gb.seconds.load() + dt); }
			else if (exec) (void)hipGraphExecDestroy(exec);              // (the structure moved on meanwhile, or the build failed: plain launches serve)
			gb.cv.notify_all();
		}
	}
	// the graph of `chunk` iterations if it exists; otherwise it is ordered (once) and nullptr returned
	hipGraphExec_t pcgGraphIfReady(int chunk, int maxIter, Scalar tol2)
	{
		if (pcgGraphTol2 != tol2 || pcgGraphMaxIter != maxIter) { dropPcgGraph(); pcgGraphTol2 = tol2; pcgGraphMaxIter = maxIter; }   // baked-in arguments
		const auto key = std::make_pair(chunk, (const Scalar*)sys.acinv);
		auto it = pcgGraphs.find(key);
		if (it != pcgGraphs.end()) return it->second;
		std::lock_guard<std::mutex> lk(gb.m);
		for (auto& kv : gb.ready) pcgGraphs[kv.first] = kv.second;        // (everything that has finished moves to the solver's own map)
		gb.ready.clear();
		it = pcgGraphs.find(key);
		if (it != pcgGraphs.end()) return it->second;
		if (!gb.requested.count(key))
		{
			gb.requested[key] = 1;
			gb.jobs.push_back(GraphJob{ chunk, (const Scalar*)sys.acinv, g, st, sys, maxIter, tol2, gb.gen });
			if (!gb.th.joinable()) gb.th = std::thread([this] { graphWorker(); });
			gb.cv.notify_all();
		}
		return nullptr;
	}
	// (tests, time_kernels: wait until every ordered graph exists)
	void waitForGraphs()
	{
		std::unique_lock<std::mutex> lk(gb.m);
		gb.cv.wait(lk, [&] { return gb.jobs.empty() && !gb.busy; });
	}
	void dropPcgGraph()
	{
		{
			std::lock_guard<std::mutex> lk(gb.m);
			gb.gen++;                                                       // a build in flight is discarded when it lands
			gb.jobs.clear(); gb.requested.clear();
			// destroying a dozen instantiated graphs costs ~1.5 ms: with a helper thread alive that is its job, not the caller's
			const bool offload = gb.th.joinable() && !gb.stop;
			for (auto& kv : gb.ready) { if (offload) gb.trash.push_back(kv.second); else (void)hipGraphExecDestroy(kv.second); }
			gb.ready.clear();
			for (auto& kv : pcgGraphs) { if (offload) gb.trash.push_back(kv.second); else (void)hipGraphExecDestroy(kv.second); }
			if (offload && !gb.trash.empty()) gb.cv.notify_all();
		}
		pcgGraphs.clear();
		batchRequests.clear();
		graphsOrderedFor = nullptr;
	}
	std::map<int, int> batchRequests;      // how often a batch of this length was asked for since the graphs were dropped
	bool exactBatchGraphs = true;          // option "pcg_exact_batch_graphs"
	bool repeatPrediction = true;          // option "pcg_repeat_prediction"

	void enqueuePcgIteration(int k, int maxIter, Scalar tol2, hipStream_t s)
	{
		if (sys.agg > 0)
		{
			launch_pcg_spmv(g, st, sys, k, maxIter, tol2, s);
			launch_pcg2_fused(g, sys, k, k + 1, maxIter, tol2, 1, s);
		}
		else launch_pcg_iteration(g, st, sys, k, maxIter, tol2, s);
	}

	Scalar pcgGraphTol2 = 0; int pcgGraphMaxIter = 0;
	const void* graphsOrderedFor = nullptr;

	bool coarseValid = false;
	// overlapped refresh: while the PCG of trial k runs (with the inverse built from trial k-1's matrix), a second stream
	// assembles and inverts trial k's coarse matrix for trial k+1
	// Pays since the sweep became light (look-ahead pivot inversion: one workgroup runs the 16-step chain, the others ~2 us of
	// tile products): 9.43 -> 9.09 ms at KITTI-00 with a refresh under every trial; at S2M ten 0.85 ms sweeps per run cost the
	// latency-bound PCG kernels more than they save (28.7 vs 28.4 ms), one under every third trial does pay (27.1 ms; G4M 67.6 -> 65.3).
	int sideAge = 0;
	int overlapPeriod() const
	{
		// KITTI-07 (Nc 372): 4.20 / 4.79 ms with period 1 / 2; KITTI-00 (672): 8.55 / 8.33 / 8.41 ms with 1 / 2 / 3; S2M (1500): 28.7 / 27.7 / 27.1 / 27.1 with 1 / 2 / 3 / 4
		const int Nc = 6 * sys.cl * sys.nc;
		return Nc <= 512 ? 1 : Nc <= 1024 ? 2 : 3;
	}
	hipStream_t gjStream = nullptr;
	hipEvent_t evSetup = nullptr, evAssembled = nullptr, evInverse = nullptr, evFirstInv = nullptr;
	int liveInv = 0, pendingInv = -1;   // buffer with the inverse in use / buffer the running inversion will leave its result in
	bool assemblePending = false;       // the other stream may still be reading hsc
	void ensureOverlapObjects()
	{
		if (gjStream) return;
		int prioLow = 0, prioHigh = 0;
		HIP_TRY(hipDeviceGetStreamPriorityRange(&prioLow, &prioHigh));
		// (low priority: the sweep fills the gaps of the latency-bound PCG kernels it runs under; confining it to every n-th CU instead
		// was measured at <= 1 %, profiles/r03*)
		HIP_TRY(hipStreamCreateWithPriority(&gjStream, hipStreamNonBlocking, prioLow));
		HIP_TRY(hipEventCreateWithFlags(&evSetup, hipEventDisableTiming));
		HIP_TRY(hipEventCreateWithFlags(&evAssembled, hipEventDisableTiming));
		HIP_TRY(hipEventCreateWithFlags(&evInverse, hipEventDisableTiming));
		HIP_TRY(hipEventCreateWithFlags(&evFirstInv, hipEventDisableTiming));
		HIP_TRY(hipEventRecord(evFirstInv, gjStream));
	}
	// the work stream must not touch what a running inversion still uses
	void waitAssembled() { if (assemblePending) { HIP_TRY(hipStreamWaitEvent(stream, evAssembled, 0)); assemblePending = false; } }
	void drainInversion()
	{
		if (pendingInv >= 0) { HIP_TRY(hipStreamWaitEvent(stream, evInverse, 0)); pendingInv = -1; }
		assemblePending = false;
	}
	struct PatternEntry { uint64_t key; int ea, eb; };   // (column << 32 | product id + 1), the product's two sorted-edge ids
	std::vector<PatternEntry> h_ent; std::vector<int> h_work[6];   // work arrays of build_structure
	std::vector<double> h_chiSorted;         // per-edge chi2 in sorted order (staging of chi_squares)
	std::vector<Scalar> h_stage[6];          // host staging of set_graph (sorted measurements, state, cameras)
	std::vector<int> h_inEp, h_inEl; std::vector<uint8_t> h_inDim;   // the caller's index arrays of the last set_graph
	std::vector<int> h_spose[2], h_slm[2];   // sorted edge->pose (with the stereo bit) / edge->landmark of this and the previous set_graph
	int topoSlot = 0;
	std::vector<int> runIters;   // PCG iterations of the solves of the current LM run (sizes the next batch of launches)
	std::vector<int> prevRunIters;   // ... and of the previous run on this structure: a run that repeats it solve for solve is sized from it
	void startRunHistory() { if (!runIters.empty()) prevRunIters.swap(runIters); runIters.clear(); }
	int firstSolveIters = 0;     // ... and of the first solve of the previous run

	double lambda = 0;
	int maxIterAlloc = 0;
	long long nmul = 0;
	int64_t cntPcgIters = 0, cntTrials = 0, cntCoarseRefresh = 0, cntPcgLooks = 0, cntPcgEnqueued = 0, cntPcgUnconverged = 0;
	int64_t cntCoarseInline = 0;      // coarse inversions that ran on the WORK stream (in front of a solve), a subset of cntCoarseRefresh
	int64_t cntFp32Fallbacks = 0;     // solves repeated with the fp64 coarse inverse after the fp32-stored one broke the PCG down
	int64_t cntUploads = 0;           // successful cuba_hip_set_graph calls on this handle (never reset: identifies what the device holds)
	bool acceptUnconverged = false;   // true: a solve that hits max_iter hands back its best iterate as a success (inexact LM step)
	std::vector<int> pcgHistory;      // PCG iterations of every reduced solve since set_graph (negative = stopped at max_iter)
	double prof[CUBA_HIP_PROFILE_ITEMS] = { 0, 0, 0, 0, 0, 0, 0, 0 };

	~cuba_hip_solver()
	{
		{ std::lock_guard<std::mutex> lk(gb.m); gb.stop = true; gb.jobs.clear(); }
		gb.cv.notify_all();
		if (gb.th.joinable()) gb.th.join();
		for (hipGraphExec_t e : gb.trash) (void)hipGraphExecDestroy(e);
		gb.trash.clear();
		dropPcgGraph();
		if (gjStream) { (void)hipStreamSynchronize(gjStream); (void)hipStreamDestroy(gjStream); (void)hipEventDestroy(evSetup); (void)hipEventDestroy(evAssembled); (void)hipEventDestroy(evInverse); (void)hipEventDestroy(evFirstInv); }
		if (captureStream) (void)hipStreamDestroy(captureStream);
		if (upStream) { (void)hipStreamSynchronize(upStream); (void)hipStreamDestroy(upStream); (void)hipEventDestroy(evValues); }
		if (h_tileStage) (void)hipHostFree(h_tileStage);
		if (evTileInputs) (void)hipEventDestroy(evTileInputs);
		if (h_pinned) (void)hipHostFree(h_pinned);
		if (ownStream && stream) (void)hipStreamDestroy(stream);
	}

	void sync() { HIP_TRY(hipStreamSynchronize(stream)); }

	// Completion of the work enqueued so far, learnt from the ticket the last reporting kernel writes into the mapped host
	// block: a spin on host memory sees it ~1 us after the kernel, hipStreamSynchronize only after ~20 us.
	bool failDirty = true;       // the device-side failure flag of the PCG may be non-zero
	int expectedTicket = 0;
	bool hintSameEdges = false, hintSameValues = false;   // cuba_hip_hint_unchanged: promises about the next set_graph call
	bool fusedTail = true;       // optimize(): back-substitution, update and evaluation of a trial in one pass over the edges (option "fused_tail")
	void noteReport() { expectedTicket++; }
	void waitReport()
	{
		volatile int* flags = (volatile int*)((char*)h_pinned + 1024);
		{
			const auto t0 = Clock::now();
			for (long spins = 0; flags[3] != expectedTicket; spins++)
				if ((spins & 0xfff) == 0xfff && std::chrono::duration<double>(Clock::now() - t0).count() > 2.0) break;   // hung or failed launch: let the runtime say so
			if (flags[3] == expectedTicket)
			{
				// the results were written by kernels that precede the ticket write in stream order: order the (non-volatile)
				// reads of the result slots after the ticket read
				std::atomic_thread_fence(std::memory_order_acquire);
				return;
			}
		}
		sync();
		std::atomic_thread_fence(std::memory_order_acquire);
		expectedTicket = flags[3];
	}

	// host double <-> device Scalar transfers (plain copies in the fp64 build, staged conversion in the fp32 build)
	void downloadAsDouble(const Scalar* dsrc, double* hdst, size_t n)
	{
		if (!n) return;
		if (sizeof(Scalar) == sizeof(double))
		{
			HIP_TRY(hipMemcpyAsync(hdst, dsrc, n * sizeof(double), hipMemcpyDeviceToHost, stream));
			sync();
			return;
		}
		std::vector<Scalar> tmp(n);
		HIP_TRY(hipMemcpyAsync(tmp.data(), dsrc, n * sizeof(Scalar), hipMemcpyDeviceToHost, stream));
		sync();
		for (size_t i = 0; i < n; i++) hdst[i] = (double)tmp[i];
	}
	void uploadFromDouble(Scalar* ddst, const double* hsrc, size_t n)
	{
		if (!n) return;
		std::vector<Scalar> tmp(hsrc, hsrc + n);
		HIP_TRY(hipMemcpyAsync(ddst, tmp.data(), n * sizeof(Scalar), hipMemcpyHostToDevice, stream));
		sync();
	}

	// run fn(row) for all rows on a few host threads (persistent pool), rows split into contiguous chunks of similar weight
	template <class F>
	static void parallelRows(int nrows, const std::vector<long long>& start, F&& fn)
	{
		const long long total = nrows > 0 ? start[nrows] - start[0] : 0;
		const int T = (int)std::min<long long>(HostPool::instance().maxThreads(), total / 50000 + 1);
		if (T <= 1) { for (int i = 0; i < nrows; i++) fn(i); return; }
		std::vector<int> cut(T + 1, 0);
		for (int t = 1; t < T; t++)
		{
			const long long target = start[0] + total * t / T;
			cut[t] = std::max(cut[t - 1], (int)(std::upper_bound(start.begin(), start.begin() + nrows + 1, target) - start.begin()) - 1);
		}
		cut[T] = nrows;
		HostPool::instance().run(T, [&](int t) { for (int i = cut[t]; i < cut[t + 1]; i++) fn(i); });
	}

	// uniform version: fn(i) for i in [0, n)
	template <class F>
	static void parallelFor(int n, F&& fn)
	{
		const int T = (int)std::min<long long>(HostPool::instance().maxThreads(), n / 50000 + 1);
		if (T <= 1) { for (int i = 0; i < n; i++) fn(i); return; }
		HostPool::instance().run(T, [&](int t) {
			const int r0 = (int)((long long)n * t / T), r1 = (int)((long long)n * (t + 1) / T);
			for (int i = r0; i < r1; i++) fn(i);
		});
	}

	// set-up phase breakdown on stderr when CUBA_HIP_DEBUG is set
	Clock::time_point lapT;
	void lap(const char* what)
	{
		static const bool on = std::getenv("CUBA_HIP_DEBUG") != nullptr;
		if (!on) return;
		const auto now = Clock::now();
		if (what) std::fprintf(stderr, "[cuba_hip] %-34s %7.2f ms\n", what, 1e3 * std::chrono::duration<double>(now - lapT).count());
		lapT = now;
	}

	struct StageTimer
	{
		cuba_hip_solver* s; int item; Clock::time_point t0;
		StageTimer(cuba_hip_solver* s_, int item_) : s(s_), item(item_)
		{
			if (s->profile) { s->sync(); t0 = Clock::now(); }
		}
		~StageTimer()
		{
			if (s->profile)
			{
				(void)hipStreamSynchronize(s->stream);
				s->prof[item] += std::chrono::duration<double>(Clock::now() - t0).count();
			}
		}
	};

	// ---------------------------------------------------------------------------------------------
	// deferValues (cuba_hip_set_graph_begin): the caller keeps meas / omega valid until cuba_hip_set_graph_end, so their 32 bytes per
	// edge may still be crossing PCIe -- on a second stream -- while the structure analysis (which needs the index arrays only) runs
	hipStream_t upStream = nullptr; hipEvent_t evValues = nullptr;
	bool valuesPending = false, deferredUpload = false;
	int* h_tileStage = nullptr; size_t tileStageCap = 0; hipEvent_t evTileInputs = nullptr;     // page-locked staging of the tile-order inputs
	void finishValues()
	{
		if (!valuesPending) return;
		valuesPending = false;
		HIP_TRY(hipStreamWaitEvent(stream, evValues, 0));
		topo::launch_gather_edges(d_perm.data(), d_rawEp.data(), d_rawEl.data(), d_rawDim.data(), d_rawMeas.data(), d_rawOmega.data(), E,
			nullptr, nullptr, d_mu.data(), d_mv.data(), d_mr.data(), d_w.data(), stream);
	}
	void setGraph(int Pt_, int Pf_, int Lt_, int Lf_, const double* q, const double* t, const double* cam, const double* Xw,
		int E_, const int32_t* ep, const int32_t* el, const uint8_t* edim, const double* meas, const double* omega, bool deferValues = false)
	{
		if (valuesPending) { HIP_TRY(hipStreamSynchronize(upStream)); valuesPending = false; }      // (a begin without its end: the old upload must not outlive its arrays' replacement)
		deferredUpload = false;
		if (Pt_ < 0 || Lt_ < 0 || E_ < 0 || Pf_ < 0 || Pf_ > Pt_ || Lf_ < 0 || Lf_ > Lt_) throw ArgError{ "bad vertex counts" };
		if (Pt_ >= STEREO_BIT) throw ArgError{ "too many poses" };
		// the per-edge linearisation record carries 2*landmark+stereo in a Scalar slot: exact in fp32 only below 2^24
		if (sizeof(Scalar) == 4 && Lt_ >= (1 << 23)) throw ArgError{ "fp32 build: at most 2^23 - 1 landmarks" };
		if ((Pt_ && (!q || !t || !cam)) || (Lt_ && !Xw) || (E_ && (!ep || !el || !edim || !meas || !omega))) throw ArgError{ "null array" };
		const auto t0 = Clock::now();
		static const bool noCache = std::getenv("CUBA_HIP_NO_STRUCTURE_CACHE") != nullptr;   // A/B knob for set-up timings
		const bool sameCounts = !noCache && haveStructure && partHi < 0 && Pt == Pt_ && Pf == Pf_ && Lt == Lt_ && Lf == Lf_ && E == E_;
		// the very same index arrays as in the previous call (re-initialisation of an unchanged graph): the sort, the
		// permutation and the sorted index arrays on the host and on the device are all still valid
		bool sameInput = !noCache && haveGraph && Pt == Pt_ && Pf == Pf_ && Lt == Lt_ && Lf == Lf_ && E == E_ && (int)h_inEp.size() == E_;
		const bool promisedEdges = sameInput && hintSameEdges, promisedValues = promisedEdges && hintSameValues;
		hintSameEdges = hintSameValues = false;          // (a promise covers one call)
		if (sameInput && !promisedEdges)
		{
			std::atomic<int> diff{ 0 };
			parallelFor(E_, [&](int e) { if (h_inEp[e] != ep[e] || h_inEl[e] != el[e] || h_inDim[e] != edim[e]) diff.store(1, std::memory_order_relaxed); });
			sameInput = diff == 0;
		}
		// validate BEFORE any member changes: a rejected call must leave the previous graph fully usable
		if (!sameInput)
		{
			std::atomic<int> bad{ 0 };
			parallelFor(E_, [&](int e) {
				if (ep[e] < 0 || ep[e] >= Pt_ || el[e] < 0 || el[e] >= Lt_) bad.store(1, std::memory_order_relaxed);
				else if (edim[e] != 2 && edim[e] != 3) bad.store(2, std::memory_order_relaxed);
				else if (ep[e] >= Pf_ && el[e] >= Lf_) bad.store(3, std::memory_order_relaxed);
			});
			if (bad == 1) throw ArgError{ "edge index out of range" };
			if (bad == 2) throw ArgError{ "edge_dim must be 2 or 3" };
			if (bad == 3) throw ArgError{ "edge with both ends fixed (must be dropped by the caller)" };
		}
		// from here on the old graph is being replaced: a failure below (allocation, upload) leaves NO graph
		haveGraph = false;
		haveStructure = false;
		Pt = Pt_; Pf = Pf_; Lt = Lt_; Lf = Lf_; E = E_;
		lap(nullptr);
		// the estimates and cameras go up straight from the caller's arrays when they need neither a conversion (fp64 build) nor a
		// row permutation (internal pose order): no staging copy, and page-locked caller memory (cuba_hip_host_alloc) moves by DMA
		std::vector<Scalar>&state = h_stage[4], &camv = h_stage[5];
		auto stageState = [&] {
			state.resize((size_t)7 * Pt + (size_t)3 * Lt);
			for (size_t i = 0; i < (size_t)4 * Pt; i++) state[i] = (Scalar)q[i];
			for (size_t i = 0; i < (size_t)3 * Pt; i++) state[4 * (size_t)Pt + i] = (Scalar)t[i];
			for (size_t i = 0; i < (size_t)3 * Lt; i++) state[7 * (size_t)Pt + i] = (Scalar)Xw[i];
			camv.assign(cam, cam + 5 * (size_t)Pt);
		};
		const DeviceGraph gOld = g;
		bool sameTopology = false;
		const bool useDev = deviceSetup && E > 0;
		if (!useDev && (!hostTopoValid || reorderActive)) sameInput = false;      // the host-side sort of the previous call does not exist (device path)
		if (useDev)
		{
			// ---- device path: raw arrays go up as they are; sort, gather and the landmark pointers are kernels -------------
			// (an internal pose order found for this very topology is kept; otherwise it starts over as the identity)
			const bool keepOrder = reorderActive && sameInput && devTopology && sameCounts;
			const bool reuseSort = sameInput && devTopology && (keepOrder || !reorderActive);
			if (!keepOrder) resetPoseOrder();
			if (!reuseSort)
			{
				if (!sameInput) { h_inEp.assign(ep, ep + E); h_inEl.assign(el, el + E); h_inDim.assign(edim, edim + E); }
				d_rawEpCaller.uploadRaw(ep, E, stream); d_rawEl.uploadRaw(el, E, stream); d_rawDim.uploadRaw(edim, E, stream);
				d_rawEp.resize(E);
				HIP_TRY(hipMemcpyAsync(d_rawEp.data(), d_rawEpCaller.data(), sizeof(int) * (size_t)E, hipMemcpyDeviceToDevice, stream));
			}
			const bool keepValues = promisedValues && reuseSort && d_mu.size() == (size_t)E && d_w.size() == (size_t)E;   // (sorted measurement / information arrays of the previous call)
			const bool defer = deferValues && !keepValues && E > 0;
			deferredUpload = defer;          // (enqueued LAST, below: the copy engine serves its queue in order, and the small uploads of this call must not wait behind 18 MB)
			if (defer) { d_rawMeas.resize((size_t)3 * E); d_rawOmega.resize(E); }
			else if (!keepValues) { d_rawMeas.uploadRaw(meas, (size_t)3 * E, stream); d_rawOmega.uploadRaw(omega, E, stream); }
			lap("set_graph: raw uploads enqueued");
			if (!reuseSort) runDeviceEdgeSort(!defer);
			else if (!keepValues && !defer)
			{
				d_mu.resize(E); d_mv.resize(E); d_mr.resize(E); d_w.resize(E);
				topo::launch_gather_edges(d_perm.data(), d_rawEp.data(), d_rawEl.data(), d_rawDim.data(), d_rawMeas.data(), d_rawOmega.data(), E,
					nullptr, nullptr, d_mu.data(), d_mv.data(), d_mr.data(), d_w.data(), stream);
			}
			if (defer) { d_mu.resize(E); d_mv.resize(E); d_mr.resize(E); d_w.resize(E); valuesPending = true; }
			sameTopology = sameCounts && reuseSort;
			devTopology = true; hostTopoValid = false;
			if (reorderActive) { stageState(); permuteStateRows(state, camv); }          // the caller's rows -> the internal pose order kept from the last call
			lap("set_graph: device sort + gather enqueued");
		}
		else
		{
		resetPoseOrder();
		if (!sameInput)
		{
		h_inEp.assign(ep, ep + E); h_inEl.assign(el, el + E); h_inDim.assign(edim, edim + E);
		// sort edges by (landmark, pose, original index): counting sort on the landmark, small sorts inside
		h_lmptr.assign(Lt + 1, 0);
		for (int e = 0; e < E; e++) h_lmptr[el[e] + 1]++;
		for (int l = 0; l < Lt; l++) h_lmptr[l + 1] += h_lmptr[l];
		perm.assign(E, 0);
		{
			std::vector<int> cursor(h_lmptr.begin(), h_lmptr.end() - 1);
			for (int e = 0; e < E; e++) perm[cursor[el[e]]++] = e;
			std::vector<long long> byEdges(h_lmptr.begin(), h_lmptr.end());    // balance the small sorts by edge count
			parallelRows(Lt, byEdges, [&](int l) {
				std::sort(perm.begin() + h_lmptr[l], perm.begin() + h_lmptr[l + 1],
					[&](int a, int b) { return ep[a] != ep[b] ? ep[a] < ep[b] : a < b; });
			});
		}
		}
		lap("set_graph: validate + sort edges");
		std::vector<int>& sPose = h_spose[sameInput ? topoSlot : topoSlot ^ 1];   // the previous call's sorted index arrays stay in the other slot
		std::vector<int>& sLm = h_slm[sameInput ? topoSlot : topoSlot ^ 1];
		sPose.resize(E); sLm.resize(E);
		// staging buffers are members: a second set_graph of similar size touches no fresh pages
		std::vector<Scalar>&mu = h_stage[0], &mv = h_stage[1], &mr = h_stage[2], &w = h_stage[3];
		mu.resize(E); mv.resize(E); mr.resize(E); w.resize(E);
		h_epose.resize(E);
		{
			parallelFor(E, [&](int i) {       // random gather through the sort permutation
				const int e = perm[i];
				if (!sameInput)
				{
					h_epose[i] = ep[e];
					sPose[i] = ep[e] | (edim[e] == 3 ? STEREO_BIT : 0);
					sLm[i] = el[e];
				}
				mu[i] = meas[3 * (size_t)e]; mv[i] = meas[3 * (size_t)e + 1];
				mr[i] = edim[e] == 3 ? meas[3 * (size_t)e + 2] : 0.0;
				w[i] = omega[e];
			});
		}
		// Same vertices, same edges (in sorted order, same types) as last time: everything build_structure() derives from
		// the topology is still valid on the device -- only the values are new (the samples' warm-up + timed protocol,
		// repeated optimisation of one window).  Decided by comparing the sorted index arrays, 8 bytes per edge.
		sameTopology = sameCounts && hostTopoValid && (sameInput || (sPose == h_spose[topoSlot] && sLm == h_slm[topoSlot]));
		if (!sameInput) topoSlot ^= 1;
		lap("set_graph: gather sorted arrays");
		if (!sameInput || devTopology) { d_epose.upload(sPose, stream); d_elm.upload(sLm, stream); d_lmptr.upload(h_lmptr, stream); }
		d_mu.upload(mu, stream); d_mv.upload(mv, stream); d_mr.upload(mr, stream); d_w.upload(w, stream);
		devTopology = false; hostTopoValid = true;
		}
		const size_t nState = (size_t)7 * Pt + (size_t)3 * Lt;
		if (sizeof(Scalar) == sizeof(double) && !reorderActive)
		{
			d_state.resize(nState); d_cam.resize((size_t)5 * Pt);
			Scalar* ds = d_state.data();
			if (Pt)
			{
				HIP_TRY(hipMemcpyAsync(ds, q, sizeof(double) * 4 * (size_t)Pt, hipMemcpyHostToDevice, stream));
				HIP_TRY(hipMemcpyAsync(ds + 4 * (size_t)Pt, t, sizeof(double) * 3 * (size_t)Pt, hipMemcpyHostToDevice, stream));
				HIP_TRY(hipMemcpyAsync(d_cam.data(), cam, sizeof(double) * 5 * (size_t)Pt, hipMemcpyHostToDevice, stream));
			}
			if (Lt) HIP_TRY(hipMemcpyAsync(ds + 7 * (size_t)Pt, Xw, sizeof(double) * 3 * (size_t)Lt, hipMemcpyHostToDevice, stream));
		}
		else
		{
			if (!reorderActive) stageState();
			d_state.upload(state, stream);
			d_cam.upload(camv, stream);
		}
		d_backup.resize(nState);
		dropSnapshots();
		d_perEdge.resize(E);
		if (!h_pinned)
		{
			// Coherent (fine-grained) mapping: the spin-wait protocol below reads device-written results without a stream
			// synchronisation, which is only defined for coherent host memory (HIP_HOST_COHERENT defaults to 0).
			HIP_TRY(hipHostMalloc((void**)&h_pinned, 4096, hipHostMallocMapped | hipHostMallocCoherent));
			std::memset(h_pinned, 0, 4096);
			void* dev = nullptr;
			HIP_TRY(hipHostGetDevicePointer(&dev, h_pinned, 0));
			slotsDev = (Scalar*)dev; flagsDev = (int*)((char*)dev + 1024);
		}
		d_parts.resize(8192 + (size_t)E + 64); d_maxdiag.resize(64); d_fail.resize(1); d_iters.resize(1); d_kbase.resize(1); d_done.resize(1); d_ticket.resize(1);
		d_fail.zero(stream); d_iters.zero(stream); d_kbase.zero(stream); d_done.zero(stream); d_ticket.zero(stream);
		if (deferredUpload)
		{
			if (!upStream) { HIP_TRY(hipStreamCreateWithFlags(&upStream, hipStreamNonBlocking)); HIP_TRY(hipEventCreateWithFlags(&evValues, hipEventDisableTiming)); }
			// (the raw buffers may still be read by a gather of the previous call on the work stream)
			HIP_TRY(hipEventRecord(evValues, stream)); HIP_TRY(hipStreamWaitEvent(upStream, evValues, 0));
			HIP_TRY(hipMemcpyAsync(d_rawMeas.data(), meas, sizeof(double) * 3 * (size_t)E, hipMemcpyHostToDevice, upStream));
			HIP_TRY(hipMemcpyAsync(d_rawOmega.data(), omega, sizeof(double) * (size_t)E, hipMemcpyHostToDevice, upStream));
			HIP_TRY(hipEventRecord(evValues, upStream));
			deferredUpload = false;
		}
		sync(); expectedTicket = 0; ((volatile int*)((char*)h_pinned + 1024))[3] = 0;
		sync();   // host staging vectors go out of scope

		lap("set_graph: alloc + upload + sync");
		g = DeviceGraph();
		g.Pt = Pt; g.Pf = Pf; g.Lt = Lt; g.Lf = Lf; g.E = E;
		g.q = d_state.data(); g.t = d_state.data() + 4 * (size_t)Pt; g.Xw = d_state.data() + 7 * (size_t)Pt;
		g.cam = d_cam.data();
		g.e_pose = d_epose.data(); g.e_lm = d_elm.data(); g.lm_ptr = d_lmptr.data();
		g.e_mu = d_mu.data(); g.e_mv = d_mv.data(); g.e_mr = d_mr.data(); g.e_w = d_w.data();
		g.rk[0] = rk[0]; g.rk[1] = rk[1];
		g.e_begin = 0; g.e_end = E;
		partLo = 0; partHi = -1;
		if (sameTopology)
		{
			// the captured PCG graphs carry the DeviceGraph by value: they stay usable only if no buffer moved
			DeviceGraph a = gOld, b = g;
			a.rk[0] = a.rk[1] = b.rk[0] = b.rk[1] = RobustKernel();
			if (std::memcmp(&a, &b, sizeof(DeviceGraph)) != 0) dropPcgGraph();
			haveStructure = true;
		}
		coarseValid = false; startRunHistory();
		haveGraph = true;
		lambda = 0;
		for (double& v : prof) v = 0;
		cntPcgIters = cntTrials = cntCoarseRefresh = cntPcgLooks = cntPcgEnqueued = cntPcgUnconverged = 0;
		cntCoarseInline = cntFp32Fallbacks = 0;
		pcgHistory.clear();
		cntUploads++;
		prof[0] += std::chrono::duration<double>(Clock::now() - t0).count();
	}

	// ---------------------------------------------------------------------------------------------
	// Symbolic structure: Hsc pattern from landmark co-visibility (ref: HschurSparseBlockMatrix::
	// constructFromVertices, src/sparse_block_matrix.cpp:55-133 -- here sort/unique instead of a dense
	// P x P map, and every free pose always owns its diagonal block), destination block of every Schur
	// product (ref: findHschureMulBlockIndicesKernel, cuda_block_solver.cu:979-1000), symmetric adjacency
	// for the PCG, wave work list.
	// ---------------------------------------------------------------------------------------------
	void buildStructure()
	{
		if (!haveGraph) throw StateError{ "set_graph must be called first" };
		if (haveStructure) return;
		if (gjStream) { HIP_TRY(hipStreamSynchronize(gjStream)); pendingInv = -1; assemblePending = false; }   // an overlapped coarse inversion uses the old structure
		if (devTopology && deviceSetup) { buildStructureDevice(); return; }      // (landmark partitions included)
		localRanges = false;
		if (reorderActive) { std::vector<int> id(Pf); for (int i = 0; i < Pf; i++) id[i] = i; applyPoseOrder(id); }   // the host pipeline runs in the caller's order
		ensureHostTopology();
		const auto t0 = Clock::now();
		std::vector<int> nfree(Lf, 0);
		std::vector<long long> pairBase(Lf, 0);
		nmul = 0;
		long long npairs = 0;
		parallelFor(Lf, [&](int l) {
			int n = 0;
			for (int i = h_lmptr[l]; i < h_lmptr[l + 1]; i++) n += h_epose[i] < Pf;   // edges are sorted by pose: the free ones come first
			nfree[l] = n;
		});
		for (int l = 0; l < Lf; l++)
		{
			const int n = nfree[l];
			pairBase[l] = npairs;
			npairs += (long long)n * (n - 1) / 2;
			nmul += (long long)n * (n + 1) / 2;
		}
		if (npairs >= (1LL << 31)) throw ArgError{ "graph too dense: more than 2^31 Schur block products" };
		lap(nullptr);
		// Pattern of Hsc + product lists: bucket the (column, product id) pairs by block row, sort every row on its own
		// (cache resident, rows spread over host threads), then walk the sorted rows: a new column opens a new block,
		// and the products of a block are the consecutive entries with its column, already in landmark order (product
		// ids grow with the landmark index) -- the fixed summation order that makes the results reproducible.
		// per free pose: its edges (sorted-edge ids, ascending = landmark order), over the whole graph
		std::vector<int> peAllPtr(Pf + 1, 0);
		std::vector<int>& peAll = h_work[5];
		{
			// counting sort by pose, split over contiguous edge ranges: per-range histograms, offsets in (pose, range)
			// order, then every range scatters its own edges -- the lists stay in ascending edge order
			const int T = (int)std::min<long long>(HostPool::instance().maxThreads(), E / 50000 + 1);
			std::vector<int> hist((size_t)T * Pf, 0);
			auto range = [&](int t) { return std::make_pair((int)((long long)E * t / T), (int)((long long)E * (t + 1) / T)); };
			HostPool::instance().run(T, [&](int t) {
				int* h = hist.data() + (size_t)t * Pf;
				for (int i = range(t).first; i < range(t).second; i++) if (h_epose[i] < Pf) h[h_epose[i]]++;
			});
			int run = 0;
			for (int ps = 0; ps < Pf; ps++)
			{
				peAllPtr[ps] = run;
				for (int t = 0; t < T; t++) { const int c = hist[(size_t)t * Pf + ps]; hist[(size_t)t * Pf + ps] = run; run += c; }
			}
			peAllPtr[Pf] = run;
			peAll.resize(run);
			HostPool::instance().run(T, [&](int t) {
				int* cur = hist.data() + (size_t)t * Pf;
				for (int i = range(t).first; i < range(t).second; i++) if (h_epose[i] < Pf) peAll[cur[h_epose[i]]++] = i;
			});
		}
		lap("structure:   nfree + pose lists");
		const std::vector<int>& slm = h_slm[topoSlot];          // sorted edge -> landmark (set_graph)
		std::vector<long long> peWeight(peAllPtr.begin(), peAllPtr.end());
		// row i of the pattern collects, for every landmark pose i sees, the poses after it in that landmark's edge list:
		// each row is produced by one thread into its own segment (no atomics)
		std::vector<long long> rowStart(Pf + 1, 0);
		{
			std::vector<long long> cnt(Pf, 1);                   // the diagonal block always exists
			parallelRows(Pf, peWeight, [&](int i) {
				long long c = 1;
				for (int x = peAllPtr[i]; x < peAllPtr[i + 1]; x++)
				{
					const int e = peAll[x], l = slm[e];
					if (l < Lf) c += nfree[l] - 1 - (e - h_lmptr[l]);
				}
				cnt[i] = c;
			});
			for (int i = 0; i < Pf; i++) rowStart[i + 1] = rowStart[i] + cnt[i];
		}
		lap("structure:   count per row");
		// (the large work arrays are members: rebuilding for the next graph touches no fresh pages)
		// an entry carries its two (sorted) edges along: every pass below streams through memory, nothing is looked up
		// by product id (a product -> edges table is written once per landmark by several rows: cache-line ping-pong)
		std::vector<PatternEntry>& ent = h_ent; ent.resize((size_t)rowStart[Pf]);
		parallelRows(Pf, rowStart, [&](int i) {
			long long slot = rowStart[i];
			ent[slot++] = PatternEntry{ (uint64_t)i << 32, -1, -1 };                    // id 0 = diagonal seed, sorts first
			for (int x = peAllPtr[i]; x < peAllPtr[i + 1]; x++)
			{
				const int e = peAll[x], l = slm[e];
				if (l >= Lf) continue;
				const int b0 = h_lmptr[l], n = nfree[l], a2 = e - b0;
				long long idx = pairBase[l] + (long long)a2 * (n - 1) - (long long)a2 * (a2 - 1) / 2;   // id of product (a2, a2 + 1)
				for (int c = a2 + 1; c < n; c++, idx++, slot++)
				{
					ent[slot] = PatternEntry{ ((uint64_t)h_epose[b0 + c] << 32) | (uint64_t)(idx + 1), e, b0 + c };
				}
			}
		});
		lap("structure:   bucket fill");
		// landmark partition (multi-GPU): the PATTERN is global, the product lists cover the landmarks [lo, hi) only --
		// product ids are in landmark order, so that is an id range
		const int lo = std::max(0, partLo), hi = partHi < 0 ? Lt : std::min(Lt, partHi);
		const long long idLo = std::min(lo, Lf) < Lf ? pairBase[std::min(lo, Lf)] : npairs;
		const long long idHi = std::min(hi, Lf) < Lf ? pairBase[std::min(hi, Lf)] : npairs;
		std::vector<int> rowBlocks(Pf, 0);
		std::vector<long long> rowProducts(Pf + 1, 0);
		parallelRows(Pf, rowStart, [&](int i) {
			PatternEntry* e0 = ent.data() + rowStart[i]; PatternEntry* e1 = ent.data() + rowStart[i + 1];
			{
				// order by (column, product id).  The entries were produced in product-id order, so a STABLE counting sort
				// on the column does it in O(n + columns spanned); rows that span far more columns than they have
				// entries (loop closures) fall back to a comparison sort
				const size_t nEnt = (size_t)(e1 - e0);
				uint32_t cmax = (uint32_t)i;
				for (PatternEntry* e = e0; e < e1; e++) cmax = std::max(cmax, (uint32_t)(e->key >> 32));
				const size_t span = (size_t)cmax - (size_t)i + 1;          // columns of an upper-triangular row start at the row
				if (span <= 8 * nEnt + 64)
				{
					thread_local std::vector<PatternEntry> tmp;
					thread_local std::vector<int> cnt;
					tmp.assign(e0, e1);
					cnt.assign(span + 1, 0);
					for (const PatternEntry& x : tmp) cnt[(size_t)(x.key >> 32) - i + 1]++;
					for (size_t c = 0; c < span; c++) cnt[c + 1] += cnt[c];
					for (const PatternEntry& x : tmp) e0[cnt[(size_t)(x.key >> 32) - i]++] = x;
				}
				else std::sort(e0, e1, [](const PatternEntry& x, const PatternEntry& y) { return x.key < y.key; });
			}
			int u = 0; uint32_t last = 0xffffffffu; long long np = 0;
			for (PatternEntry* e = e0; e < e1; e++)
			{
				const uint32_t c = (uint32_t)(e->key >> 32); u += c != last; last = c;
				const long long id = (long long)(uint32_t)e->key - 1;
				np += id >= idLo && id < idHi;
			}
			rowBlocks[i] = u; rowProducts[i + 1] = np;
		});
		lap("structure:   row sorts");
		h_rowptr.assign(Pf + 1, 0);
		for (int i = 0; i < Pf; i++) { h_rowptr[i + 1] = h_rowptr[i] + rowBlocks[i]; rowProducts[i + 1] += rowProducts[i]; }
		const int nblk = h_rowptr[Pf];
		const long long nprodLocal = rowProducts[Pf];
		h_colind.assign(nblk, 0);
		std::vector<int> blkRow(nblk), prodPtr(nblk + 1, 0), odBlocks;
		std::vector<int>&prodEa = h_work[2], &prodEb = h_work[3];
		prodEa.resize((size_t)nprodLocal); prodEb.resize((size_t)nprodLocal);
		parallelRows(Pf, rowStart, [&](int i) {
			int k = h_rowptr[i] - 1; uint32_t last = 0xffffffffu;
			long long out = rowProducts[i];
			for (long long x = rowStart[i]; x < rowStart[i + 1]; x++)
			{
				const uint32_t c = (uint32_t)(ent[x].key >> 32); const uint32_t id = (uint32_t)ent[x].key;
				if (c != last) { k++; h_colind[k] = (int)c; blkRow[k] = i; prodPtr[k] = (int)out; last = c; }
				if (!id) continue;
				if ((long long)id - 1 >= idLo && (long long)id - 1 < idHi) { prodEa[out] = ent[x].ea; prodEb[out] = ent[x].eb; out++; }
			}
		});
		prodPtr[nblk] = (int)nprodLocal;
		lap("structure: Hsc pattern + product blocks");
		// symmetric adjacency over the upper storage
		std::vector<int> adjPtr(Pf + 1, 0);
		for (int i = 0; i < Pf; i++)
			for (int k = h_rowptr[i]; k < h_rowptr[i + 1]; k++)
			{
				adjPtr[i + 1]++;
				if (h_colind[k] != i) adjPtr[h_colind[k] + 1]++;
			}
		for (int i = 0; i < Pf; i++) adjPtr[i + 1] += adjPtr[i];
		std::vector<int> adjBlk(adjPtr[Pf]), adjCol(adjPtr[Pf]);
		{
			std::vector<int> cur(adjPtr.begin(), adjPtr.end() - 1);
			// lower part first (neighbours j < i arrive in increasing j), then the row's own upper part
			for (int i = 0; i < Pf; i++)
				for (int k = h_rowptr[i]; k < h_rowptr[i + 1]; k++)
				{
					const int j = h_colind[k];
					if (j != i) { adjBlk[cur[j]] = k | (int)0x80000000; adjCol[cur[j]] = i; cur[j]++; }
				}
			for (int i = 0; i < Pf; i++)
				for (int k = h_rowptr[i]; k < h_rowptr[i + 1]; k++) { adjBlk[cur[i]] = k; adjCol[cur[i]] = h_colind[k]; cur[i]++; }
		}
		lap("structure: adjacency");
		g.e_begin = h_lmptr[lo]; g.e_end = h_lmptr[hi];
		// blocks with products, longest lists first (the block pass takes them in this order)
		{
			int maxCnt = 0;
			for (int k = 0; k < nblk; k++) maxCnt = std::max(maxCnt, prodPtr[k + 1] - prodPtr[k]);
			std::vector<int> start(maxCnt + 2, 0);                 // stable counting sort by descending list length
			for (int k = 0; k < nblk; k++) { const int c = prodPtr[k + 1] - prodPtr[k]; if (c > 0) start[maxCnt - c + 1]++; }
			for (int c = 0; c <= maxCnt; c++) start[c + 1] += start[c];
			odBlocks.resize(start[maxCnt + 1]);
			for (int k = 0; k < nblk; k++) { const int c = prodPtr[k + 1] - prodPtr[k]; if (c > 0) odBlocks[start[maxCnt - c]++] = k; }
			// (plain row order was measured slower: 174 vs 135 us at KITTI-00 -- the long lists must start first; an XCD-aware order cut the
			// HBM-side fetch 2-3 x and bought nothing: the pass is latency-bound, profiles/r03j_block_order.txt)
			if (rowGroupedBlocks(nprodLocal))
			{
				std::vector<int> cntOf(nblk);
				for (int k = 0; k < nblk; k++) cntOf[k] = prodPtr[k + 1] - prodPtr[k];
				odBlocks = rowGroupedOrder(blkRow.data(), h_colind.data(), cntOf.data(), nblk);
			}
		}
		lap("structure: product lists");
		// per free pose: its edges inside this handle's landmark range (a contiguous run of the global list)
		std::vector<int> pePtr(Pf + 1, 0), peEdge;
		const bool wholeGraph = g.e_begin == 0 && g.e_end == E;      // then the global lists are the lists (no copy)
		if (!wholeGraph)
		{
			for (int i = g.e_begin; i < g.e_end; i++) if (h_epose[i] < Pf) pePtr[h_epose[i] + 1]++;
			for (int i = 0; i < Pf; i++) pePtr[i + 1] += pePtr[i];
			peEdge.resize(pePtr[Pf]);
			std::vector<int> cur(pePtr.begin(), pePtr.end() - 1);
			for (int i = g.e_begin; i < g.e_end; i++) if (h_epose[i] < Pf) peEdge[cur[h_epose[i]]++] = i;
		}
		lap("structure: pose edge lists");
		// wave work list: whole landmarks, at most 64 edges per wave; larger landmarks get a workgroup each
		std::vector<int> waveLm, bigLm;
		std::vector<long long> bigOfs;
		long long bigEdges = 0;
		{
			int start = -1, cnt = 0;
			auto flush = [&](int end) { if (start >= 0 && cnt > 0) { waveLm.push_back(start); waveLm.push_back(end); } start = -1; cnt = 0; };
			for (int l = lo; l < hi; l++)
			{
				// (the device pipeline packs chunks of WAVE_CHUNK landmarks independently: the same cuts here, so that both
				// pipelines produce the same waves -- the per-wave partial sums of the fused trial tail depend on them)
				if ((l - lo) % topo::WAVE_CHUNK == 0) flush(l);
				const int n = h_lmptr[l + 1] - h_lmptr[l];
				if (n > WAVE)
				{
					flush(l);
					bigLm.push_back(l); bigOfs.push_back(bigEdges); bigEdges += n;
					continue;
				}
				if (n == 0) continue;   // empty landmarks inside a run are harmless (no lanes)
				if (start >= 0 && cnt + n > WAVE) flush(l);
				if (start < 0) start = l;
				cnt += n;
			}
			flush(hi);
		}

		lap("structure: wave list");
		d_waveLm.upload(waveLm, stream); d_bigLm.upload(bigLm, stream); d_bigOfs.upload(bigOfs, stream);
		d_bigHpl.resize((size_t)bigEdges * 18);
		d_rowptr.upload(h_rowptr, stream); d_colind.upload(h_colind, stream);
		d_lmNfree.upload(nfree, stream);
		d_adjPtr.upload(adjPtr, stream); d_adjBlk.upload(adjBlk, stream); d_adjCol.upload(adjCol, stream);
		int ellM = 0, ellOver = 0;
		{
			int maxRow = 0;
			for (int i = 0; i < Pf; i++) maxRow = std::max(maxRow, adjPtr[i + 1] - adjPtr[i]);
			const int M = std::min(3, (maxRow + 19) / 20);
			ellM = M; ellOver = maxRow > 20 * M;
			std::vector<int2> ell((size_t)Pf * M * 20, int2{ 0, -1 });
			for (int i = 0; i < Pf; i++)
			{
				const int n = std::min(adjPtr[i + 1] - adjPtr[i], 20 * M);
				for (int e = 0; e < n; e++) ell[(size_t)i * M * 20 + e] = int2{ adjBlk[adjPtr[i] + e], adjCol[adjPtr[i] + e] };
			}
			d_ell.upload(ell, stream);
		}
		d_blkrow.upload(blkRow, stream); d_odBlocks.upload(odBlocks, stream); d_prodPtr.upload(prodPtr, stream);
		d_prodEa.upload(prodEa, stream); d_prodEb.upload(prodEb, stream); d_pePtr.upload(wholeGraph ? peAllPtr : pePtr, stream); d_peEdge.upload(wholeGraph ? peAll : peEdge, stream);
		const CoarseCfg cc = coarseConfig();
		allocSystem(nblk, cc);
		const int agg = cc.agg, cl = cc.cl, nc = cc.nc;
		lap("structure: uploads + allocs");
		// coarse-matrix assembly lists: fine blocks grouped by the coarse block (I,J) they fall into (both triangles)
		std::vector<int> cbI, cbJ, cbPtr(1, 0), cbBlk, adjRow(adjBlk.size());
		std::vector<Scalar> cbWi, cbWj;
		auto weight = [&](int pose) {     // same formula as agg_weight() on the device
			if (pose == Pf - 1 && Pf % agg == 1) return Scalar(0);
			return Scalar(2 * (pose % agg) + 1 - agg) / Scalar(agg);
		};
		for (int i = 0; i < Pf; i++) for (int a = adjPtr[i]; a < adjPtr[i + 1]; a++) adjRow[a] = i;
		if (nc > 0)
		{
			std::vector<uint64_t> ck; ck.reserve(adjBlk.size());
			for (int i = 0; i < Pf; i++)
				for (int a = adjPtr[i]; a < adjPtr[i + 1]; a++)
					ck.push_back(((uint64_t)((size_t)(i / agg) * nc + adjCol[a] / agg) << 32) | (uint32_t)a);
			std::sort(ck.begin(), ck.end());
			for (size_t x = 0; x < ck.size(); x++)
			{
				const int cbid = (int)(ck[x] >> 32);
				if (x == 0 || cbid != (int)(ck[x - 1] >> 32))
				{
					if (x) cbPtr.push_back((int)cbBlk.size());
					cbI.push_back(cbid / nc); cbJ.push_back(cbid % nc);
				}
				cbBlk.push_back(adjBlk[(uint32_t)ck[x]]);
				if (cl == 2) { cbWi.push_back(weight(adjRow[(uint32_t)ck[x]])); cbWj.push_back(weight(adjCol[(uint32_t)ck[x]])); }
			}
			cbPtr.push_back((int)cbBlk.size());
		}
		d_cbI.upload(cbI, stream); d_cbJ.upload(cbJ, stream); d_cbPtr.upload(cbPtr, stream); d_cbBlk.upload(cbBlk, stream); d_cbWi.upload(cbWi, stream); d_cbWj.upload(cbWj, stream);
		sync();
		lap("structure: coarse lists + sync");
		diagProdBlocks = 0; for (int k : odBlocks) diagProdBlocks += k >= 0 && blkRow[k] == h_colind[k];
		heavyBlocks = 0;        // (the plain list is sorted by length; the tile-grouped one is not: all blocks then take the 16-lane path)
		if (!rowGroupedBlocks(nprodLocal))
			for (int k : odBlocks) heavyBlocks += prodPtr[k + 1] - prodPtr[k] > BP_HEAVY;
		publishStructure(nblk, (int)waveLm.size() / 2, (int)bigLm.size(), (int)odBlocks.size(), (int)cbI.size(), ellM, ellOver, cc);
		hostPatternValid = true;
		const double dt = std::chrono::duration<double>(Clock::now() - t0).count();
		prof[1] += 0.5 * dt; prof[5] += 0.5 * dt;   // pattern of Hsc doubles as the "symbolic" phase of the reduced solver
	}

	// coarse level of the preconditioner: aggregates of consecutive free poses
	struct CoarseCfg { int agg, cl, nc, spmvRows; };
	CoarseCfg coarseConfig() const
	{
		int agg = pcgAggregate;
		const int cl = coarseLinear ? 2 : 1;
		// automatic size: coarse dimension <= ~700-960 (scripts/experiments/agg_sweep.py: iterations vs the O(Nc^3) inversion)
		// (small graphs want smaller aggregates: KITTI-07, 247 free poses: 24 / 16 / 12 / 8 / 6 / 4 poses -> 8.6 / 7.0 / 6.4 / 6.3 / 6.5 / 7.9 ms)
		// (round 4, fp32-stored inverse + unrolled pivot chain: KITTI-07 10 / 8 / 6 / 4 poses -> 4.00 / 3.65 / 3.39 / 4.30 ms, 250 / 212 / 195 / 238 iterations:
		// the floor of the small-graph rule went from 8 to 6, profiles/r04m_sweeps.txt)
		// (large graphs, inversion hidden under the PCG of earlier trials: S2M 44 / 40 / 36 / 32 poses -> 27.1 / 26.35 / 26.7 / 26.4 ms,
		// G4M 88 / 72 / 64 / 56 / 48 -> 65.1 / 59.6 / 54.8 / 54.0 / 58.3 ms: the aggregate count may grow from 115 to 180 with the graph)
		// (round 3, coarse inverse stored in fp32 -- its apply costs half: KITTI-00 24 / 20 / 16 poses -> 8.16 / 8.00 / 7.81 ms, S2M 44 / 40 / 36 /
		// 32 / 28 -> 25.8 / 25.0 / 24.8 / 24.6 / 26.0 ms, G4M 64 / 56 / 48 / 40 -> 51.7 / 50.5 / 52.8 / 58.2 ms: profiles/r03i_agg_sweep.txt)
		const int ncMax = std::min(180, std::max(115, (Pf + 31) / 32));
		if (agg < 0) agg = cl == 2 ? (Pf >= 1320 ? std::max(16, (Pf + ncMax - 1) / ncMax) : std::max(6, (Pf + 27) / 55)) : std::max(12, (Pf + 159) / 160);
		const int spmvRows = spmv_rows_for(Pf);
		if (agg > 0) agg = (agg + spmvRows - 1) / spmvRows * spmvRows;   // aggregates = whole SpMV workgroups (sys.qpart)
		int nc = agg > 0 ? (Pf + agg - 1) / agg : 0;
		// the two-level kernel keeps two coarse vectors in LDS and the dense inverse costs O(Nc^3): a user-chosen aggregate
		// that small for this many poses is widened
		while (agg > 0 && (cl * nc > 600 || sizeof(Scalar) * (12 * (size_t)cl * nc + 12 * (size_t)agg + 200) > 60 * 1024)) { agg *= 2; nc = (Pf + agg - 1) / agg; }
		if (nc < 2) { agg = 0; nc = 0; }
		return CoarseCfg{ agg, cl, nc, spmvRows };
	}

	// everything whose size follows from (E, Pf, Lf, nblk) and the coarse configuration
	int rzStrideCfg = 1, pqStrideCfg = 1;
	void allocSystem(int nblk, const CoarseCfg& c)
	{
		d_red.resize((size_t)36 * nblk + (size_t)12 * Pf);
		d_lmSys.resize((size_t)9 * Lf); d_lmInv.resize((size_t)8 * std::max(Lf, 1)); d_erec.resize((size_t)8 * E); d_xp.resize((size_t)6 * Pf); d_xl.resize((size_t)3 * Lf);
		d_minv.resize((size_t)36 * Pf);
		d_r.resize((size_t)6 * Pf); d_z.resize((size_t)6 * Pf); d_p0.resize((size_t)6 * Pf); d_p1.resize((size_t)6 * Pf); d_ap.resize((size_t)6 * Pf);
		d_red.zero(stream); d_lmSys.zero(stream); d_lmInv.zero(stream); d_xp.zero(stream); d_xl.zero(stream);
		reducedZeroed = true;
		for (auto& b : d_coarse) b.resize((size_t)36 * c.cl * c.cl * c.nc * c.nc);
		{
			const size_t n = (size_t)6 * c.cl * c.nc;
			for (auto& b : d_coarse32) b.resize(fp32Inverse() ? n * ((n + 3) & ~(size_t)3) : 0);
		}
		d_rc.resize((size_t)12 * c.cl * c.nc); d_r2.resize((size_t)6 * Pf);
		maxIterAlloc = pcgMaxIter > 0 ? pcgMaxIter : std::min(32768, std::max(64, 4 * 6 * Pf));
		const int gridSetup = (Pf + PCG_SETUP_POSES - 1) / PCG_SETUP_POSES, gridUpd = (Pf + 39) / 40, gridSpmv = (Pf + c.spmvRows - 1) / c.spmvRows;
		rzStrideCfg = std::max(1, std::max(std::max(gridSetup, gridUpd), c.nc)); pqStrideCfg = std::max(1, gridSpmv);
		d_rz.resize((size_t)5 * rzStrideCfg); d_pq.resize((size_t)4 * pqStrideCfg);
	}

	// kernel-argument structures from the device buffers (identical for the host-built and the device-built structure)
	// Order of the blocks in the Schur block pass for graphs beyond 2^19 products: the 16 blocks of a workgroup come from ONE tile of the
	// block matrix where possible -- the records they share then hit the CU's L1 after the first group's miss, and the pass is bound by the L1's
	// outstanding misses (PMC: 4.6 L1->L2 requests per product, 546 cycles each, the L1 stalled on pending misses for 70 % of the launch:
	// profiles/r03zw_pmc_schur_and_pcg_kernels.txt) --, tiles' leftovers re-chunked in tile order (neighbouring tiles share records too),
	// chunks ordered by their longest list, -1 padding.  KITTI-00: linearise + Schur 110.7 -> 102.3 us, S2M 399 -> 358 us.
	static bool rowGroupedBlocks(long long products) { return products > (1LL << 19); }
	// (flat arrays and radix / counting sorts: the grouping sits on the critical path of a NEW topology -- 1.4 ms at KITTI-00 as nested
	// vectors with comparison sorts, ~0.2 ms like this; the output is the same list)
	std::vector<int> rowGroupedOrder(const int* blkRow, const int* blkCol, const int* cnt, int nblk) const
	{
		// groups are 4 x 4 tiles of the block matrix: a-side records are shared by 4 blocks of a workgroup, b-side records by 4.
		// KITTI-00 linearise + Schur: pieces of one row 102.9 us, 2 x 8 tiles 99.3, 4 x 4 tiles 99.6; S2M 356 / 341.6 / 340.9 us
		// (profiles/r03fin3_block_order_tiles.txt)
		const int tr = 4, tc = 4;
		const long long nColTiles = (Pf + tc - 1) / tc;
		(void)nColTiles;
		// 1. blocks with products grouped by tile, in (tile row, tile column, block id) order, each tile's blocks then by list length
		//    descending (stable).  The blocks come sorted by (row, column) -- BSR order --, so a band of 4 block rows is 4 sorted runs: a
		//    4-way merge on the tile column visits every block once, and a tile holds at most 16 blocks (insertion sort).
		// 2. a tile with all 16 blocks is a whole chunk; the other tiles' blocks, in tile order, are re-chunked (neighbouring tiles share
		//    records too)
		std::vector<int> val; val.reserve(nblk);     // whole chunks, 16 entries each
		std::vector<int> rest; rest.reserve(nblk);
		std::vector<int> chunkStart;                 // chunk c = 16 consecutive entries of `val` from chunkStart[c] (whole chunks) or of `rest` (id - n)
		for (int k0 = 0; k0 < nblk;)
		{
			const int band = blkRow[k0] / tr;
			int head[4], end[4], nr = 0, k = k0;
			while (k < nblk && blkRow[k] / tr == band)
			{
				const int r = blkRow[k], b = k;
				while (k < nblk && blkRow[k] == r) k++;
				head[nr] = b; end[nr] = k; nr++;
			}
			for (;;)
			{
				int ct = 0x7fffffff;
				for (int x = 0; x < nr; x++) if (head[x] < end[x]) ct = std::min(ct, blkCol[head[x]] / tc);
				if (ct == 0x7fffffff) break;
				int tile[16], nt = 0;
				for (int x = 0; x < nr; x++)
					while (head[x] < end[x] && blkCol[head[x]] / tc == ct) { if (cnt[head[x]] > 0) tile[nt++] = head[x]; head[x]++; }
				for (int a2 = 1; a2 < nt; a2++)
				{
					const int v = tile[a2]; int b2 = a2;
					while (b2 > 0 && cnt[tile[b2 - 1]] < cnt[v]) { tile[b2] = tile[b2 - 1]; b2--; }
					tile[b2] = v;
				}
				if (nt == 16) { chunkStart.push_back((int)val.size()); val.insert(val.end(), tile, tile + 16); }
				else rest.insert(rest.end(), tile, tile + nt);
			}
			k0 = k;
		}
		const size_t nWhole = chunkStart.size();
		const size_t n = (size_t)nblk + 1;           // (ids >= n address `rest`)
		for (size_t i = 0; i < rest.size(); i += 16)
		{
			// (a chunk of leftovers: longest list first inside it, ties in the order they came)
			const size_t e = std::min(rest.size(), i + 16);
			for (size_t a2 = i + 1; a2 < e; a2++)
			{
				const int v = rest[a2]; size_t b2 = a2;
				while (b2 > i && cnt[rest[b2 - 1]] < cnt[v]) { rest[b2] = rest[b2 - 1]; b2--; }
				rest[b2] = v;
			}
			chunkStart.push_back((int)(n + i));
		}
		// 3. chunks by their longest list, descending (stable): counting sort over the lengths that occur
		const size_t nChunks = chunkStart.size();
		auto firstOf = [&](size_t c) { return chunkStart[c] < (int)n ? val[chunkStart[c]] : rest[chunkStart[c] - n]; };
		std::vector<std::pair<int, int>> byLen(nChunks);
		for (size_t c = 0; c < nChunks; c++) byLen[c] = std::make_pair(-cnt[firstOf(c)], (int)c);
		std::stable_sort(byLen.begin(), byLen.end(), [](const std::pair<int, int>& x, const std::pair<int, int>& y) { return x.first < y.first; });
		std::vector<int> od(nChunks * 16, -1);
		for (size_t o = 0; o < nChunks; o++)
		{
			const size_t c = (size_t)byLen[o].second;
			if (c < nWhole) for (int x = 0; x < 16; x++) od[o * 16 + x] = val[chunkStart[c] + x];
			else
			{
				const size_t b = (size_t)chunkStart[c] - n, e = std::min(rest.size(), b + 16);
				for (size_t x = b; x < e; x++) od[o * 16 + (x - b)] = rest[x];
			}
		}
		return od;
	}
	int diagProdBlocks = 0;      // diagonal blocks with products (duplicate observations), set by the structure builders
	int heavyBlocks = 0;         // blocks with more than BP_HEAVY products (the first ones of d_odBlocks), set by the structure builders
	void publishStructure(int nblk, int nWaves, int nBig, int nOd, int nCb, int ellM, int ellOver, const CoarseCfg& c)
	{
		const int agg = c.agg, cl = c.cl, nc = c.nc, spmvRows = c.spmvRows;
		const int gridSetup = (Pf + PCG_SETUP_POSES - 1) / PCG_SETUP_POSES, gridUpd = (Pf + 39) / 40, gridSpmv = (Pf + spmvRows - 1) / spmvRows;
		st = DeviceStructure();
		st.nWaves = nWaves; st.wave_lm = d_waveLm.data();
		st.nBig = nBig; st.big_lm = d_bigLm.data(); st.big_scratch_ofs = d_bigOfs.data(); st.big_hpl = d_bigHpl.data();
		st.nblk = nblk; st.hsc_rowptr = d_rowptr.data(); st.hsc_colind = d_colind.data();
		st.lm_nfree = d_lmNfree.data();
		st.adj_ptr = d_adjPtr.data(); st.adj_blk = d_adjBlk.data(); st.adj_col = d_adjCol.data();
		st.ell = d_ell.data(); st.ell_m = ellM; st.ell_over = ellOver;
		st.hsc_blkrow = d_blkrow.data(); st.nOd = nOd; st.nDiagProd = diagProdBlocks; st.od_blocks = d_odBlocks.data(); st.nHeavy = std::min(heavyBlocks, nOd);
		// (whole-wave blocks shorten the longest dependent chain of the block pass: 51 -> 37 us at KITTI-07; on graphs whose pass is bound by its
		// gathers they only add waves: 114 -> 122 us at KITTI-00, 405 -> 411 us at S2M -- profiles/r03z_block_pass_ab.txt)
		if (d_prodEa.size() > ((size_t)1 << 19)) st.nHeavy = 0;
		// (64-byte rows of the landmark inverses: block pass -3 us / landmark pass +5 us at KITTI-00, -16 / +3 us at S2M)
		st.inv_rows8 = Lf >= 250000;
		st.prod_ptr = d_prodPtr.data(); st.prod_ea = d_prodEa.data(); st.prod_eb = d_prodEb.data();
		if (!localRanges) fillProdLm();          // (a device-built partition needed it earlier)
		st.prod_lm = d_prodLm.data();
		st.prod_beg = localRanges ? d_prodBeg.data() : d_prodPtr.data(); st.prod_end = localRanges ? d_prodEnd.data() : d_prodPtr.data() + 1;
		st.pe_beg = localRanges ? d_peBeg.data() : d_pePtr.data(); st.pe_end = localRanges ? d_peEnd.data() : d_pePtr.data() + 1;
		st.pe_ptr = d_pePtr.data(); st.pe_edge = d_peEdge.data(); st.e_rec = d_erec.data();
		st.nCb = nCb; st.cb_I = d_cbI.data(); st.cb_J = d_cbJ.data(); st.cb_ptr = d_cbPtr.data(); st.cb_blk = d_cbBlk.data(); st.cb_wi = d_cbWi.data(); st.cb_wj = d_cbWj.data();
		sys = DeviceSystem();
		sys.hsc = d_red.data(); sys.bsc = d_red.data() + (size_t)36 * nblk; sys.bp = sys.bsc + (size_t)6 * Pf;
		sys.lm_sys = d_lmSys.data(); sys.lm_inv = d_lmInv.data(); sys.xp = d_xp.data(); sys.xl = d_xl.data(); sys.slots = slotsDev; sys.host_flags = flagsDev; sys.parts = d_parts.data();
		sys.maxdiag = d_maxdiag.data(); sys.fail = d_fail.data();
		sys.minv = d_minv.data(); sys.r = d_r.data(); sys.z = d_z.data(); sys.p0 = d_p0.data(); sys.p1 = d_p1.data(); sys.ap = d_ap.data();
		sys.rz = d_rz.data(); sys.pq = d_pq.data(); sys.iters = d_iters.data(); sys.kbase = d_kbase.data(); sys.ticket = d_ticket.data();
		dropPcgGraph();
		firstInvValid = false; firstInvPending = false; prevRunIters.clear();
		sys.rzStride = rzStrideCfg; sys.pqStride = pqStrideCfg; sys.npq = gridSpmv;
		sys.nrz0 = agg > 0 ? nc : gridSetup; sys.nrz = agg > 0 ? nc : gridUpd; sys.done = d_done.data();
		coarseValid = false;
		d_qpart.resize(agg > 0 ? (size_t)(agg / spmvRows) * 6 * cl * nc : 1); d_gjPivots.resize(2 * 32 * 32); sys.gj_pivots = d_gjPivots.data();
		sys.qpart = d_qpart.data();   // [workgroup within its aggregate][coarse unknown]
		d_qpart.zero(stream);        // sets of SpMV workgroups the last aggregate does not have are read as zeros by the two-level kernel
		d_hrow.resize((size_t)36 * 20 * ellM * Pf); sys.hrow = d_hrow.data();
		d_hrow.zero(stream);         // (padding slots are never written)
		sys.spmv_rows = spmvRows;
		sys.agg = agg; sys.nc = nc; sys.cl = agg > 0 ? cl : 1; sys.inv_agg = agg > 0 ? Scalar(1) / Scalar(agg) : Scalar(0); sys.acinv = d_coarse[0].data(); sys.rc = d_rc.data(); sys.r2 = d_r2.data();
		sys.acinv32 = fp32Inverse() && agg > 0 ? d_coarse32[0].data() : nullptr;
		haveStructure = true;
	}
	bool hostPatternValid = false;     // h_rowptr / h_colind describe the current structure (the device-built one downloads them on demand)
	void fillProdLm()
	{
		d_prodLm.resize(d_prodEa.size());
		topo::launch_gather_int(d_prodEa.data(), d_elm.data(), d_prodEa.size(), d_prodLm.data(), stream);
	}

	// ---- internal pose order ------------------------------------------------------------------------------------------
	int farOffset() const { return std::max(24, Pf / 8); }      // "far from the diagonal", in block columns
	void resetPoseOrder()
	{
		if (reorderActive) dropSnapshots();          // (they hold rows in the order that ends here)
		reorderActive = false;
		poseNewOfOld.resize(Pf); poseOldOfNew.resize(Pf);
		for (int i = 0; i < Pf; i++) poseNewOfOld[i] = poseOldOfNew[i] = i;
	}
	// rows of a per-pose array (`width` numbers per pose, first Pf rows) between the caller's and the internal order
	template <class T>
	void permutePoseArray(T* a, int width, bool toInternal) const
	{
		if (!reorderActive) return;
		std::vector<T> tmp(a, a + (size_t)width * Pf);
		for (int old = 0; old < Pf; old++)
		{
			const int nw = poseNewOfOld[old];
			const T* src = tmp.data() + (size_t)width * (toInternal ? old : nw);
			T* dst = a + (size_t)width * (toInternal ? nw : old);
			for (int k = 0; k < width; k++) dst[k] = src[k];
		}
	}
	void permuteStateRows(std::vector<Scalar>& state, std::vector<Scalar>& camv) const
	{
		permutePoseArray(state.data(), 4, true);
		permutePoseArray(state.data() + 4 * (size_t)Pt, 3, true);
		permutePoseArray(camv.data(), 5, true);
	}

	// keys (landmark, pose) of the raw device edge arrays -> sort permutation, sorted edge arrays, landmark pointers
	void runDeviceEdgeSort(bool withValues = true)
	{
		if (withValues && valuesPending) { valuesPending = false; HIP_TRY(hipStreamWaitEvent(stream, evValues, 0)); }     // (a re-sort under a new pose order: the values must have landed)
		d_mu.resize(E); d_mv.resize(E); d_mr.resize(E); d_w.resize(E);
		d_k64a.resize(E); d_k64b.resize(E); d_v32a.resize(E); d_perm.resize(E); d_counters.resize(topo::CNT_COUNT);
		d_epose.resize(E); d_elm.resize(E); d_lmptr.resize((size_t)Lt + 1);
		d_counters.zero(stream);
		topo::launch_edge_keys(d_rawEp.data(), d_rawEl.data(), d_rawDim.data(), E, Pt, Pf, Lt, Lf, d_k64a.data(), d_v32a.data(), d_counters.data(), stream);
		const size_t tb = topo::sort_temp_bytes(E);
		d_topoTemp.resize(std::max(tb, d_topoTemp.size()));
		HIP_TRY(topo::sort_u64_u32(d_topoTemp.data(), d_topoTemp.size(), d_k64a.data(), d_k64b.data(), d_v32a.data(), d_perm.data(), E, 32 + bitsFor(Lt), stream));
		topo::launch_gather_edges(d_perm.data(), d_rawEp.data(), d_rawEl.data(), d_rawDim.data(), d_rawMeas.data(), d_rawOmega.data(), E,
			d_epose.data(), d_elm.data(), withValues ? d_mu.data() : nullptr, d_mv.data(), d_mr.data(), d_w.data(), stream);
		topo::launch_segment_ptr(d_elm.data(), E, Lt, d_lmptr.data(), stream);
	}

	// Strongest-neighbour walk over the pose graph weighted by the number of Schur products per block (= co-visible landmarks):
	// start at the pose of smallest weighted degree, always step to the heaviest unvisited neighbour, when stuck continue from
	// the unvisited pose most strongly tied to the visited ones.  On a keyframe trajectory this IS the trajectory order, loop
	// closures included (consecutive frames share far more landmarks than revisits do); scripts/experiments/precond_experiment6.py.
	// (A bandwidth-minimising order is the wrong tool: RCM interleaves the laps of a revisited stretch, the coarse space then
	// cannot move one lap against the other and the PCG needs 1358 instead of 74 iterations.)
	std::vector<int> chainOrder(const std::vector<int>& rowptr, const std::vector<int>& colind, const std::vector<int>& prodPtr) const
	{
		const int n = Pf;
		std::vector<int> adjP(n + 1, 0);
		for (int i = 0; i < n; i++)
			for (int k = rowptr[i]; k < rowptr[i + 1]; k++) if (colind[k] != i) { adjP[i + 1]++; adjP[colind[k] + 1]++; }
		for (int i = 0; i < n; i++) adjP[i + 1] += adjP[i];
		std::vector<int> adjJ(adjP[n]), adjW(adjP[n]), cur(adjP.begin(), adjP.end() - 1);
		std::vector<long long> deg(n, 0);
		for (int i = 0; i < n; i++)
			for (int k = rowptr[i]; k < rowptr[i + 1]; k++)
			{
				const int j = colind[k];
				if (j == i) continue;
				const int w = std::max(1, prodPtr[k + 1] - prodPtr[k]);
				adjJ[cur[i]] = j; adjW[cur[i]++] = w; adjJ[cur[j]] = i; adjW[cur[j]++] = w;
				deg[i] += w; deg[j] += w;
			}
		std::vector<char> visited(n, 0);
		std::vector<int> order; order.reserve(n);
		std::vector<std::pair<int, int>> heap;            // (weight, -pose) of unvisited poses next to visited ones
		int at = -1;
		for (int i = 0; i < n; i++) if (deg[i] > 0 && (at < 0 || deg[i] < deg[at])) at = i;
		if (at < 0) at = 0;
		int nextUnvisited = 0;
		while ((int)order.size() < n)
		{
			visited[at] = 1; order.push_back(at);
			int best = -1, bw = -1;
			for (int x = adjP[at]; x < adjP[at + 1]; x++)
			{
				const int j = adjJ[x];
				if (visited[j]) continue;
				heap.emplace_back(adjW[x], -j); std::push_heap(heap.begin(), heap.end());
				if (adjW[x] > bw || (adjW[x] == bw && j < best)) { best = j; bw = adjW[x]; }
			}
			if (best >= 0) { at = best; continue; }
			at = -1;
			while (!heap.empty())
			{
				std::pop_heap(heap.begin(), heap.end());
				const int j = -heap.back().second; heap.pop_back();
				if (!visited[j]) { at = j; break; }
			}
			if (at < 0)
			{
				while (nextUnvisited < n && visited[nextUnvisited]) nextUnvisited++;
				if (nextUnvisited >= n) break;
				at = nextUnvisited;
			}
		}
		std::vector<int> newOfOld(n);
		for (int k = 0; k < n; k++) newOfOld[order[k]] = k;
		return newOfOld;
	}

	// renumber the free poses internally (device path only): state / camera rows, the pose index of every edge, then the edge
	// sort again; the structure has to be rebuilt afterwards
	void applyPoseOrder(const std::vector<int>& newOfOld)
	{
		std::vector<Scalar> state(d_state.size()), camv((size_t)5 * Pt);
		HIP_TRY(hipMemcpyAsync(state.data(), d_state.data(), sizeof(Scalar) * state.size(), hipMemcpyDeviceToHost, stream));
		HIP_TRY(hipMemcpyAsync(camv.data(), d_cam.data(), sizeof(Scalar) * camv.size(), hipMemcpyDeviceToHost, stream));
		sync();
		// back to the caller's order with the order in force, then into the new one
		permutePoseArray(state.data(), 4, false); permutePoseArray(state.data() + 4 * (size_t)Pt, 3, false); permutePoseArray(camv.data(), 5, false);
		poseNewOfOld = newOfOld;
		reorderActive = false;
		for (int i = 0; i < Pf; i++) { poseOldOfNew[newOfOld[i]] = i; if (newOfOld[i] != i) reorderActive = true; }
		permuteStateRows(state, camv);
		d_state.upload(state, stream); d_cam.upload(camv, stream);
		d_poseMap.upload(poseNewOfOld, stream);
		topo::launch_remap_poses(d_rawEpCaller.data(), d_poseMap.data(), E, Pf, d_rawEp.data(), stream);
		runDeviceEdgeSort();
		sync();          // the host vectors above go out of scope
		dropSnapshots(); // (round-3 advisor: a snapshot taken in the previous order would assign pose rows to the wrong poses)
		haveStructure = false; hostTopoValid = false; hostPatternValid = false;
	}

	// after a caller-order structure build: is the pose order bad enough to look for a better one?  true = renumbered,
	// build the structure again
	bool tryReorder(int nblk, int farBlocks)
	{
		const int offDiag = nblk - Pf;
		if (!poseReorder || Pf < 48 || offDiag <= 0 || 2 * (long long)farBlocks < offDiag) return false;
		std::vector<int> rp((size_t)Pf + 1), ci(nblk), pp((size_t)nblk + 1);
		HIP_TRY(hipMemcpyAsync(rp.data(), d_rowptr.data(), sizeof(int) * rp.size(), hipMemcpyDeviceToHost, stream));
		HIP_TRY(hipMemcpyAsync(ci.data(), d_colind.data(), sizeof(int) * ci.size(), hipMemcpyDeviceToHost, stream));
		HIP_TRY(hipMemcpyAsync(pp.data(), d_prodPtr.data(), sizeof(int) * pp.size(), hipMemcpyDeviceToHost, stream));
		sync();
		const std::vector<int> order = chainOrder(rp, ci, pp);
		// worth it only if the new order really is more local
		long long farNew = 0;
		for (int i = 0; i < Pf; i++)
			for (int k = rp[i]; k < rp[i + 1]; k++) farNew += std::abs(order[i] - order[ci[k]]) > farOffset();
		if (std::getenv("CUBA_HIP_DEBUG")) std::fprintf(stderr, "[cuba_hip] pose order: %d of %d off-diagonal blocks far from the diagonal, %lld after the walk\n", farBlocks, offDiag, farNew);
		if (2 * farNew >= farBlocks) return false;
		applyPoseOrder(order);
		return true;
	}

	// the host pipeline (landmark partitions, atomic Schur kernel) needs the sorted arrays the device path kept to itself
	void ensureHostTopology()
	{
		if (hostTopoValid || !devTopology) return;
		std::vector<uint32_t> p32(E);
		h_lmptr.resize((size_t)Lt + 1);
		std::vector<int>&sPose = h_spose[topoSlot], &sLm = h_slm[topoSlot];
		sPose.resize(E); sLm.resize(E);
		if (E)
		{
			HIP_TRY(hipMemcpyAsync(p32.data(), d_perm.data(), sizeof(uint32_t) * E, hipMemcpyDeviceToHost, stream));
			HIP_TRY(hipMemcpyAsync(sPose.data(), d_epose.data(), sizeof(int) * E, hipMemcpyDeviceToHost, stream));
			HIP_TRY(hipMemcpyAsync(sLm.data(), d_elm.data(), sizeof(int) * E, hipMemcpyDeviceToHost, stream));
		}
		HIP_TRY(hipMemcpyAsync(h_lmptr.data(), d_lmptr.data(), sizeof(int) * ((size_t)Lt + 1), hipMemcpyDeviceToHost, stream));
		sync();
		perm.assign(p32.begin(), p32.end());
		h_epose.resize(E);
		for (int i = 0; i < E; i++) h_epose[i] = sPose[i] & ~STEREO_BIT;
		hostTopoValid = true;
	}

	// ---------------------------------------------------------------------------------------------
	// Symbolic structure on the device (ba_structure.hip): radix sorts + scans + segment pointers.  Same outputs as the host
	// pipeline above (block pattern, product lists, pose edge lists, adjacency, fixed-width rows, coarse assembly lists, wave
	// list); three host synchronisations to learn the counts that size the next allocations.
	// ---------------------------------------------------------------------------------------------
	template <class T> T readBack(const T* dev)
	{
		T v;
		HIP_TRY(hipMemcpyAsync(&v, dev, sizeof(T), hipMemcpyDeviceToHost, stream));
		sync();
		return v;
	}
	void sortTemp(size_t n) { const size_t tb = std::max(topo::sort_temp_bytes(n), topo::scan_temp_bytes(n)); d_topoTemp.resize(std::max(tb, d_topoTemp.size())); }

	void buildStructureDevice()
	{
		const auto t0 = Clock::now();
		lap(nullptr);
		int* cnt = d_counters.data();
		d_counters.zero(stream);
		sortTemp((size_t)std::max(E, Lf + 1));
		// 1. per landmark: free-pose edges, pose pairs; exclusive scan -> first product id of every landmark
		d_lmNfree.resize(Lf); d_pairCount.resize((size_t)Lf + 1); d_lmPairBase.resize((size_t)Lf + 1);
		d_freeCount.resize((size_t)Lf + 1); d_freeScan.resize((size_t)Lf + 1);
		topo::launch_lm_pairs(d_lmptr.data(), d_epose.data(), Lf, Pf, d_lmNfree.data(), d_pairCount.data(), d_freeCount.data(), stream);
		HIP_TRY(topo::exclusive_scan_i64(d_topoTemp.data(), d_topoTemp.size(), d_pairCount.data(), d_lmPairBase.data(), (size_t)Lf + 1, stream));
		HIP_TRY(topo::exclusive_scan_i64(d_topoTemp.data(), d_topoTemp.size(), d_freeCount.data(), d_freeScan.data(), (size_t)Lf + 1, stream));
		// 2. per free pose: its edges in ascending (= landmark) order -- a stable sort by pose
		d_k32a.resize(E); d_k32b.resize(E); d_v32a.resize(E); d_v32b.resize(E); d_tmpI0.resize(E);
		topo::launch_pose_keys(d_epose.data(), E, Pf, d_k32a.data(), d_v32a.data(), stream);
		HIP_TRY(topo::sort_u32_u32(d_topoTemp.data(), d_topoTemp.size(), d_k32a.data(), d_k32b.data(), d_v32a.data(), d_v32b.data(), E, bitsFor(Pf), stream));
		d_peEdge.resize(E); d_pePtr.resize((size_t)Pf + 1);
		topo::launch_copy_u32_to_int(d_v32b.data(), d_peEdge.data(), E, stream);
		topo::launch_copy_u32_to_int(d_k32b.data(), d_tmpI0.data(), E, stream);
		topo::launch_segment_ptr(d_tmpI0.data(), E, Pf, d_pePtr.data(), stream);
		// 3. wave work list, pass 1 (counts per chunk of landmarks) + scan
		// landmark partition (multi-GPU): the block PATTERN, the adjacency and the coarse lists are global -- every rank must hold the same
		// reduced-system layout --, the wave list covers the landmarks [lo, hi) only, and the product / pose-edge lists (in landmark
		// order) are walked over the sub-ranges that belong to those landmarks
		const int lo = std::max(0, partLo), hi = partHi < 0 ? Lt : std::min(Lt, partHi);
		localRanges = partHi >= 0;
		const int nChunks = (hi - lo + topo::WAVE_CHUNK - 1) / topo::WAVE_CHUNK;
		d_chunk.resize((size_t)3 * std::max(1, nChunks));
		if (nChunks == 0) d_chunk.zero(stream);
		topo::launch_wave_count(d_lmptr.data(), lo, hi, d_chunk.data(), stream);
		topo::launch_wave_scan(d_chunk.data(), nChunks, cnt, stream);
		// ---- synchronisation 1: number of products, of free-pose edges, of waves ------------------------------------------
		int hc[topo::CNT_COUNT];
		long long npairs = 0, nFreeEdges = 0;      // sums over the free landmarks of n (n - 1) / 2 and of n (n = edges with a free pose)
		HIP_TRY(hipMemcpyAsync(&npairs, d_lmPairBase.data() + Lf, sizeof(long long), hipMemcpyDeviceToHost, stream));
		HIP_TRY(hipMemcpyAsync(&nFreeEdges, d_freeScan.data() + Lf, sizeof(long long), hipMemcpyDeviceToHost, stream));
		HIP_TRY(hipMemcpyAsync(hc, cnt, sizeof hc, hipMemcpyDeviceToHost, stream));
		int eRange[2] = { 0, E };
		if (localRanges)
		{
			HIP_TRY(hipMemcpyAsync(&eRange[0], d_lmptr.data() + lo, sizeof(int), hipMemcpyDeviceToHost, stream));
			HIP_TRY(hipMemcpyAsync(&eRange[1], d_lmptr.data() + hi, sizeof(int), hipMemcpyDeviceToHost, stream));
		}
		sync();
		lap("structure (device): pairs, pose lists, wave counts");
		if (Lf == 0) npairs = 0;
		if (npairs >= (1LL << 31) - Pf) throw ArgError{ "graph too dense: more than 2^31 Schur block products" };
		nmul = npairs + nFreeEdges;
		const int nWaves = hc[topo::CNT_NWAVES], nBig = hc[topo::CNT_NBIG];
		const long long bigEdges = (long long)hc[topo::CNT_BIGEDGES_LO] | ((long long)hc[topo::CNT_BIGEDGES_HI] << 31);
		d_waveLm.resize((size_t)2 * nWaves); d_bigLm.resize(nBig); d_bigOfs.resize(nBig); d_bigHpl.resize((size_t)bigEdges * 18);
		topo::launch_wave_write(d_lmptr.data(), lo, hi, d_chunk.data(), d_waveLm.data(), d_bigLm.data(), d_bigOfs.data(), stream);
		g.e_begin = eRange[0]; g.e_end = eRange[1];
		if (localRanges)
		{
			d_peBeg.resize(Pf); d_peEnd.resize(Pf);
			topo::launch_segment_subrange(d_pePtr.data(), Pf, d_peEdge.data(), g.e_begin, g.e_end, d_peBeg.data(), d_peEnd.data(), stream);
		}
		// 4. pattern entries (diagonal seeds + one per product), sorted by (row, column); head flags; block index of every entry
		const size_t nEnt = (size_t)Pf + (size_t)npairs;
		d_k64a.resize(nEnt); d_k64b.resize(nEnt); d_v64a.resize(nEnt); d_v64b.resize(nEnt);
		sortTemp(nEnt);
		topo::launch_pattern_entries(d_lmptr.data(), d_epose.data(), d_elm.data(), d_lmNfree.data(), d_lmPairBase.data(), E, Lf, Pf, d_k64a.data(), d_v64a.data(), stream);
		HIP_TRY(topo::sort_u64_u64(d_topoTemp.data(), d_topoTemp.size(), d_k64a.data(), d_k64b.data(), d_v64a.data(), d_v64b.data(), nEnt, 32 + bitsFor(Pf), stream));
		d_tmpI0.resize(std::max(nEnt, (size_t)E)); d_tmpI1.resize(std::max(nEnt, (size_t)E));
		topo::launch_entry_heads(d_k64b.data(), nEnt, d_tmpI0.data(), stream);
		int nblk = 0;
		if (nEnt)
		{
			HIP_TRY(topo::inclusive_scan_i32(d_topoTemp.data(), d_topoTemp.size(), d_tmpI0.data(), d_tmpI1.data(), nEnt, stream));
			// ---- synchronisation 2: number of blocks ------------------------------------------------------------------
			nblk = readBack(d_tmpI1.data() + (nEnt - 1));
		}
		lap("structure (device): entries sorted, blocks counted");
		// 5. blocks + product lists, row pointers
		d_colind.resize(nblk); d_blkrow.resize(nblk); d_prodPtr.resize((size_t)nblk + 1); d_prodEa.resize((size_t)npairs); d_prodEb.resize((size_t)npairs);
		d_rowptr.resize((size_t)Pf + 1);
		if (nblk == 0) { d_prodPtr.zero(stream); d_rowptr.zero(stream); }
		topo::launch_blocks_from_entries(d_k64b.data(), d_v64b.data(), d_tmpI1.data(), nEnt, Pf, d_colind.data(), d_blkrow.data(), d_prodPtr.data(),
			d_prodEa.data(), d_prodEb.data(), stream);
		topo::launch_segment_ptr(d_blkrow.data(), nblk, Pf, d_rowptr.data(), stream);
		// (the tile order of the Schur block pass -- a host computation over the block list -- needs only what exists from here on: its
		// inputs start their way to a page-locked staging block now, and the host works on them while the device runs steps 6-8)
		const bool earlyTiles = rowGroupedBlocks(npairs) && nblk > 0 && !localRanges;
		if (earlyTiles)
		{
			if (tileStageCap < (size_t)3 * nblk + 1)
			{
				if (h_tileStage) (void)hipHostFree(h_tileStage);
				h_tileStage = nullptr; tileStageCap = 0;
				HIP_TRY(hipHostMalloc((void**)&h_tileStage, sizeof(int) * ((size_t)3 * nblk + 1), hipHostMallocDefault));
				tileStageCap = (size_t)3 * nblk + 1;
			}
			if (!evTileInputs) HIP_TRY(hipEventCreateWithFlags(&evTileInputs, hipEventDisableTiming));
			HIP_TRY(hipMemcpyAsync(h_tileStage, d_blkrow.data(), sizeof(int) * nblk, hipMemcpyDeviceToHost, stream));
			HIP_TRY(hipMemcpyAsync(h_tileStage + nblk, d_colind.data(), sizeof(int) * nblk, hipMemcpyDeviceToHost, stream));
			HIP_TRY(hipMemcpyAsync(h_tileStage + 2 * (size_t)nblk, d_prodPtr.data(), sizeof(int) * ((size_t)nblk + 1), hipMemcpyDeviceToHost, stream));
			HIP_TRY(hipEventRecord(evTileInputs, stream));
		}
		// 6. blocks with products, longest list first
		const size_t n32 = std::max((size_t)std::max(nblk, E), (size_t)2 * nblk);
		d_k32a.resize(n32); d_k32b.resize(n32); d_v32a.resize(n32); d_v32b.resize(n32);
		sortTemp(n32);
		d_odBlocks.resize(nblk);
		if (localRanges)
		{
			fillProdLm();
			d_prodBeg.resize(nblk); d_prodEnd.resize(nblk);
			topo::launch_segment_subrange(d_prodPtr.data(), nblk, d_prodLm.data(), lo, hi, d_prodBeg.data(), d_prodEnd.data(), stream);
		}
		topo::launch_od_keys(localRanges ? d_prodBeg.data() : d_prodPtr.data(), localRanges ? d_prodEnd.data() : d_prodPtr.data() + 1,
			d_blkrow.data(), d_colind.data(), nblk, farOffset(), BP_HEAVY, d_k32a.data(), d_v32a.data(), cnt, stream);
		if (nblk) HIP_TRY(topo::sort_u32_u32(d_topoTemp.data(), d_topoTemp.size(), d_k32a.data(), d_k32b.data(), d_v32a.data(), d_v32b.data(), nblk, 32, stream));
		topo::launch_copy_u32_to_int(d_v32b.data(), d_odBlocks.data(), nblk, stream);
		// 7. symmetric adjacency: the lower part of every row comes from the (column, row)-sorted list of the off-diagonal blocks
		const int nAdj = std::max(0, 2 * nblk - Pf);
		d_lowerPtr.resize((size_t)Pf + 1); d_adjPtr.resize((size_t)Pf + 1); d_adjBlk.resize(nAdj); d_adjCol.resize(nAdj); d_adjRow.resize(nAdj);
		// (both scratch arrays serve step 8 as well: head flags / their scan over the nAdj = 2 nblk - Pf adjacency entries, which
		// exceeds E and Pf + npairs when most pose pairs share a single landmark)
		d_tmpI0.resize(std::max(std::max((size_t)nblk, (size_t)nAdj), d_tmpI0.size())); d_tmpI1.resize(std::max((size_t)nAdj, d_tmpI1.size()));
		d_k64a.resize(std::max((size_t)nblk, d_k64a.size())); d_k64b.resize(std::max((size_t)nblk, d_k64b.size()));
		topo::launch_transpose_keys(d_colind.data(), d_blkrow.data(), nblk, d_k64a.data(), d_v32a.data(), stream);
		if (nblk) HIP_TRY(topo::sort_u64_u32(d_topoTemp.data(), d_topoTemp.size(), d_k64a.data(), d_k64b.data(), d_v32a.data(), d_v32b.data(), nblk, 64, stream));
		topo::launch_keys_hi(d_k64b.data(), nblk, Pf, d_tmpI0.data(), stream);
		topo::launch_segment_ptr(d_tmpI0.data(), nblk, Pf, d_lowerPtr.data(), stream);
		topo::launch_adj_ptr(d_rowptr.data(), d_lowerPtr.data(), Pf, d_adjPtr.data(), cnt, stream);
		topo::launch_adj_fill(d_rowptr.data(), d_colind.data(), d_blkrow.data(), d_lowerPtr.data(), d_k64b.data(), d_v32b.data(), nblk, d_adjPtr.data(),
			d_adjBlk.data(), d_adjCol.data(), d_adjRow.data(), stream);
		// 8. coarse-matrix assembly lists: adjacency entries grouped by the coarse block they fall into (stable: entry order kept)
		const CoarseCfg cc = coarseConfig();
		d_cbI.resize(nAdj); d_cbJ.resize(nAdj); d_cbPtr.resize((size_t)nAdj + 1); d_cbBlk.resize(nAdj);
		d_cbWi.resize(cc.cl == 2 ? nAdj : 0); d_cbWj.resize(cc.cl == 2 ? nAdj : 0);
		if (cc.nc > 0 && nAdj > 0)
		{
			topo::launch_coarse_keys(d_adjRow.data(), d_adjCol.data(), nAdj, cc.agg, cc.nc, d_k32a.data(), d_v32a.data(), stream);
			HIP_TRY(topo::sort_u32_u32(d_topoTemp.data(), d_topoTemp.size(), d_k32a.data(), d_k32b.data(), d_v32a.data(), d_v32b.data(), nAdj, bitsFor((long long)cc.nc * cc.nc), stream));
			topo::launch_heads_u32(d_k32b.data(), nAdj, d_tmpI0.data(), stream);
			HIP_TRY(topo::inclusive_scan_i32(d_topoTemp.data(), d_topoTemp.size(), d_tmpI0.data(), d_tmpI1.data(), nAdj, stream));
			topo::launch_coarse_lists(d_k32b.data(), d_v32b.data(), d_tmpI1.data(), d_adjBlk.data(), d_adjRow.data(), d_adjCol.data(), nAdj, cc.agg, cc.nc, Pf, cc.cl,
				d_cbI.data(), d_cbJ.data(), d_cbPtr.data(), d_cbBlk.data(), d_cbWi.data(), d_cbWj.data(), cnt, stream);
		}
		allocSystem(nblk, cc);
		// ---- synchronisation 3: widest adjacency row, numbers of product blocks and of coarse blocks -----------------------------
		HIP_TRY(hipMemcpyAsync(hc, cnt, sizeof hc, hipMemcpyDeviceToHost, stream));
		std::vector<int> earlyOd;
		if (earlyTiles)
		{
			HIP_TRY(hipEventSynchronize(evTileInputs));
			const int* pp = h_tileStage + 2 * (size_t)nblk;
			std::vector<int> len(nblk);
			for (int k = 0; k < nblk; k++) len[k] = pp[k + 1] - pp[k];
			earlyOd = rowGroupedOrder(h_tileStage, h_tileStage + nblk, len.data(), nblk);
		}
		sync();
		lap("structure (device): adjacency, coarse lists, allocations");
		const int maxRow = hc[topo::CNT_MAXROW];
		const int ellM = std::min(3, (maxRow + 19) / 20), ellOver = maxRow > 20 * ellM;
		d_ell.resize((size_t)Pf * ellM * 20);
		topo::launch_ell(d_adjPtr.data(), d_adjBlk.data(), d_adjCol.data(), Pf, ellM, d_ell.data(), stream);
		if (!reorderActive && !reorderTried && tryReorder(nblk, hc[topo::CNT_FARBLOCKS]))
		{
			reorderTried = true;
			buildStructureDevice();            // once more, now in the internal pose order
			return;
		}
		reorderTried = false;
		diagProdBlocks = hc[topo::CNT_DIAGPROD]; heavyBlocks = hc[topo::CNT_NHEAVY];
		int nOdList = hc[topo::CNT_NOD];
		if (earlyTiles)
		{
			d_odBlocks.upload(earlyOd, stream);
			sync();          // `earlyOd` is a local
			nOdList = (int)earlyOd.size(); heavyBlocks = 0;
		}
		else if (rowGroupedBlocks(npairs) && nblk > 0)
		{
			// (landmark partitions: the list lengths are those of the rank's sub-ranges, known only after step 6)
			std::vector<int> hRow(nblk), hBeg(nblk), hEnd(nblk), hCol;
			hCol.resize(nblk); HIP_TRY(hipMemcpyAsync(hCol.data(), d_colind.data(), sizeof(int) * nblk, hipMemcpyDeviceToHost, stream));
			HIP_TRY(hipMemcpyAsync(hRow.data(), d_blkrow.data(), sizeof(int) * nblk, hipMemcpyDeviceToHost, stream));
			HIP_TRY(hipMemcpyAsync(hBeg.data(), localRanges ? d_prodBeg.data() : d_prodPtr.data(), sizeof(int) * nblk, hipMemcpyDeviceToHost, stream));
			HIP_TRY(hipMemcpyAsync(hEnd.data(), localRanges ? d_prodEnd.data() : d_prodPtr.data() + 1, sizeof(int) * nblk, hipMemcpyDeviceToHost, stream));
			sync();
			for (int k = 0; k < nblk; k++) hEnd[k] -= hBeg[k];          // list lengths
			const std::vector<int> od = rowGroupedOrder(hRow.data(), hCol.data(), hEnd.data(), nblk);
			d_odBlocks.upload(od, stream);
			sync();          // `od` is a local: the copy must have left it (round-3 advisor)
			nOdList = (int)od.size(); heavyBlocks = 0;
		}
		lap("structure (device): block order of the Schur pass");
		publishStructure(nblk, nWaves, nBig, nOdList, cc.nc > 0 ? hc[topo::CNT_NCB] : 0, ellM, ellOver, cc);
		hostPatternValid = false;
		lap("structure (device): published");
		if (std::getenv("CUBA_HIP_DEBUG")) std::fprintf(stderr, "[cuba_hip] structure (device): nblk %d products %lld waves %d big %d od %d coarse blocks %d max row %d\n",
			nblk, npairs, nWaves, nBig, hc[topo::CNT_NOD], hc[topo::CNT_NCB], maxRow);
		const double dt = std::chrono::duration<double>(Clock::now() - t0).count();
		prof[1] += 0.5 * dt; prof[5] += 0.5 * dt;
	}

	// the block pattern / values as the CALLER numbers the poses (introspection entry points; identity unless reorderActive)
	struct CallerBlock { uint64_t key; int src; bool transposed; };
	std::vector<CallerBlock> callerBlocks()
	{
		ensureHostPattern();
		std::vector<CallerBlock> b; b.reserve(h_colind.size());
		for (int i = 0; i < Pf; i++)
			for (int k = h_rowptr[i]; k < h_rowptr[i + 1]; k++)
			{
				int r = poseOldOfNew[i], c = poseOldOfNew[h_colind[k]];
				const bool tr = r > c;
				if (tr) std::swap(r, c);
				b.push_back(CallerBlock{ ((uint64_t)(uint32_t)r << 32) | (uint32_t)c, k, tr });
			}
		std::sort(b.begin(), b.end(), [](const CallerBlock& x, const CallerBlock& y) { return x.key < y.key; });
		return b;
	}

	void ensureHostPattern()
	{
		if (hostPatternValid) return;
		h_rowptr.resize((size_t)Pf + 1); h_colind.resize(st.nblk);
		HIP_TRY(hipMemcpyAsync(h_rowptr.data(), d_rowptr.data(), sizeof(int) * h_rowptr.size(), hipMemcpyDeviceToHost, stream));
		if (st.nblk) HIP_TRY(hipMemcpyAsync(h_colind.data(), d_colind.data(), sizeof(int) * h_colind.size(), hipMemcpyDeviceToHost, stream));
		sync();
		hostPatternValid = true;
	}

	void need() { if (!haveGraph) throw StateError{ "set_graph must be called first" }; buildStructure(); finishValues(); g.rk[0] = rk[0]; g.rk[1] = rk[1]; st.mixed = mixedPrecision ? 1 : 0; }

	double readSlots(int which)
	{
		sync();
		double s = 0;
		for (int i = 0; i < NSLOT; i++) s += slot(which * NSLOT + i);
		return s;
	}

	double computeErrors()
	{
		need();
		StageTimer tm(this, 2);
		launch_residual_chi2(g, d_parts.data(), slotsDev, nullptr, stream);   // its second stage writes all NSLOT entries of the slot group
		return readSlots(0);
	}

	// [hsc | bsc | bp] needs zeroing only where a block may have no writer: a landmark
	// partition leaves blocks without local products; otherwise the pose pass writes every diagonal block, bp and bsc
	// and the block pass every off-diagonal block
	// (force: the assemble-only mode writes the diagonal blocks' upper triangles and bp only -- off-diagonal blocks, bsc and the lower
	// triangles would otherwise keep a previous trial's values, which the stage API exposes through cuba_hip_get_array /
	// cuba_hip_reduction_buffer and a multi-GPU driver sums)
	void zeroReduced(bool force = false) { waitAssembled(); if (force || partHi >= 0 || !reducedZeroed) { d_red.zero(stream); reducedZeroed = true; } }
	bool reducedZeroed = false;

	// withBackup: the state is also copied into its backup (push() of the LM loop) -- inside the landmark pass's launch where possible
	void linearize(int mode, double lam, bool withBackup = false)
	{
		waitAssembled();            // an overlapped coarse assembly may still be reading the previous reduced matrix
		launch_linearize_dm(g, st, sys, mode, lam, stream, withBackup ? d_state.data() : nullptr, d_backup.data(), d_state.size());
	}

	// assemble only: Hpp -> diagonal blocks, bp, raw Hll/bl, landmark part of the max diagonal
	void assemble()
	{
		need();
		StageTimer tm(this, 3);
		zeroReduced(true);
		d_maxdiag.zero(stream);
		linearize(0, 0.0);
	}

	// max diagonal of the (possibly externally reduced) Hpp and of the local Hll
	void maxDiagonalParts(double* posePart, double* lmPart)
	{
		need();
		const double* hD = (const double*)hostStage();      // maxdiag slots hold bit patterns of non-negative doubles
		HIP_TRY(hipMemcpyAsync(hostStage(), d_maxdiag.data(), 8 * 64, hipMemcpyDeviceToHost, stream));
		sync();
		double v = 0;
		for (int i = 0; i < 64; i++) v = std::max(v, hD[i]);
		*lmPart = v;
		d_maxdiag.zero(stream);
		launch_pose_maxdiag(g, st, sys, stream);
		HIP_TRY(hipMemcpyAsync(hostStage(), d_maxdiag.data(), 8 * 64, hipMemcpyDeviceToHost, stream));
		sync();
		v = 0;
		for (int i = 0; i < 64; i++) v = std::max(v, hD[i]);
		*posePart = v;
	}

	void scaleParts(double lam, double* posePart, double* lmPart)
	{
		need();
		launch_pose_scale(g, sys, lam, slotsDev + 3 * NSLOT, stream);
		launch_landmark_scale(g, sys, lam, slotsDev + 2 * NSLOT, stream);
		sync();
		double a = 0, b = 0;
		for (int i = 0; i < NSLOT; i++) { b += slot(2 * NSLOT + i); a += slot(3 * NSLOT + i); }
		*posePart = a; *lmPart = b;
	}

	double maxDiagonal()
	{
		need();
		StageTimer tm(this, 3);
		zeroReduced(true);
		d_maxdiag.zero(stream);
		linearize(0, 0.0);
		launch_pose_maxdiag(g, st, sys, stream);
		HIP_TRY(hipMemcpyAsync(hostStage(), d_maxdiag.data(), 8 * 64, hipMemcpyDeviceToHost, stream));
		sync();
		double v = 0;   // bit patterns of non-negative doubles are doubles again
		for (int i = 0; i < 64; i++) v = std::max(v, ((const double*)hostStage())[i]);
		return v;
	}

	void schur(bool withBackup = false)
	{
		need();
		StageTimer tm(this, 4);
		zeroReduced();
		linearize(1, lambda, withBackup);
	}

	// A PCG that BREAKS DOWN (p.Ap <= 0 or a NaN -- not a solve that merely runs out of iterations) while the coarse inverse is stored in
	// fp32 is repeated once with fp64 storage, which the handle then keeps: rounding a symmetrised inverse to fp32 perturbs it by
	// ~6e-8 ||Ac^-1||, which can cost positive definiteness once lambda_max(block) / lambda_min(Ac) approaches 1e7 (weakly constrained
	// graphs at very small damping; round-3 advisor).  Counted in "precond_fp32_fallbacks".
	bool lastSolveBrokeDown = false;
	bool solveReduced()
	{
		const bool ok = solveReducedOnce();
		if (ok || !lastSolveBrokeDown || !fp32Inverse() || sys.agg <= 0) return ok;
		precondFp32 = false; sys.acinv32 = nullptr;
		dropPcgGraph();
		coarseValid = false; firstInvValid = false; firstInvPending = false;
		cntFp32Fallbacks++;
		if (std::getenv("CUBA_HIP_DEBUG")) std::fprintf(stderr, "[cuba_hip] PCG broke down with the fp32-stored coarse inverse: repeating the solve with fp64 storage\n");
		return solveReducedOnce();
	}

	bool solveReducedOnce()
	{
		lastSolveBrokeDown = false;
		need();
		StageTimer tm(this, 6);
		if (Pf == 0) return true;
		const int maxIter = maxIterAlloc;
		const Scalar tol2 = pcgTol * pcgTol;
		if (failDirty) { d_fail.zero(stream); failDirty = false; }      // (the device flag only changes when a solve fails, and every solve reports it)
		const bool twoLevel = sys.agg > 0;
		// an inversion that ran on the second stream under the previous trial's PCG: its result moves into the buffer the iteration
		// graphs read within the next launch
		const size_t invCount = (size_t)36 * sys.cl * sys.cl * sys.nc * sys.nc;
		bool takeInverse = false;
		if (twoLevel && coarseValid && pendingInv >= 0)
		{
			HIP_TRY(hipStreamWaitEvent(stream, evInverse, 0));     // normally long done
			takeInverse = true; pendingInv = -1;
		}
		// block-Jacobi inverses, r0 / z0, flags (clears `done` and the iteration offset) + row-ordered copy of the damped matrix for the SpMV
		if (fp32Inverse())     // the overlapped inversion left an fp32 copy in the staging buffer: that is what moves into the buffer in use
			launch_pcg_setup_expand(g, st, sys, lambda, stream, takeInverse ? reinterpret_cast<const Scalar*>(d_coarse32[1].data()) : nullptr,
				reinterpret_cast<Scalar*>(d_coarse32[0].data()), inv32Count() / 2);
		else launch_pcg_setup_expand(g, st, sys, lambda, stream, takeInverse ? d_coarse[0].data() : nullptr, d_coarse[2].data(), invCount);
		if (twoLevel)
		{
			// the sweep ping-pongs between two buffers: start in the one that leaves the inverse in d_coarse[0]
			const int gjSteps = (6 * sys.cl * sys.nc + 31) / 32, first = gjSteps & 1;
			{
				// The inverse in use lives in d_coarse[2] (the iteration graphs have the pointer baked in); d_coarse[0 / 1] are
				// the work buffers of the sweep, which leaves its result in d_coarse[0].
				ensureOverlapObjects();
				const size_t invBytes = sizeof(Scalar) * (size_t)36 * sys.cl * sys.cl * sys.nc * sys.nc;
				if (!coarseValid && coarseFirstReuse && firstInvValid)
				{
					// first solve of a run on a structure that has seen a run before: start with the inverse that run's first solve had
					// and let this trial's own inversion run on the other stream right away
					HIP_TRY(hipStreamWaitEvent(stream, evFirstInv, 0));       // (the copy the previous run's first trial left on the other stream: long done)
					if (fp32Inverse()) HIP_TRY(hipMemcpyAsync(d_coarse32[0].data(), d_firstInv32.data(), inv32Count() * sizeof(float), hipMemcpyDeviceToDevice, stream));
					else HIP_TRY(hipMemcpyAsync(d_coarse[2].data(), d_firstInv.data(), invBytes, hipMemcpyDeviceToDevice, stream));
					pendingInv = -1;                 // (a sweep the previous run left behind is simply overtaken: the streams order themselves)
					coarseValid = true; sideAge = overlapPeriod(); firstInvPending = true;      // (this trial's own matrix is inverted on the other stream, below)
				}
				else if (!coarseValid)
				{
					// first solve on this structure: nothing to overlap with, invert here
					drainInversion();
					(void)launch_coarse_setup(g, st, sys, d_coarse[first].data(), d_coarse[1 - first].data(), stream);
					if (fp32Inverse()) launch_coarse_to_fp32(d_coarse[0].data(), d_coarse32[0].data(), 6 * sys.cl * sys.nc, stream);
					else HIP_TRY(hipMemcpyAsync(d_coarse[2].data(), d_coarse[0].data(), invBytes, hipMemcpyDeviceToDevice, stream));
					coarseValid = true; cntCoarseRefresh++; cntCoarseInline++; sideAge = 0;
					if (coarseFirstReuse)
					{
						// every run keeps ONE schedule of overlapped inversions -- under trial 1, 1 + period, ... --, whether its first solve was
						// given an in-line inverse (here: the first run on a structure; the sweep under trial 1 then repeats this inversion) or
						// the carried-over one: repeating a run from the same estimate reproduces it bit for bit.  The sweep under trial 1
						// leaves its result for the next run's first solve.
						if (fp32Inverse()) d_firstInv32.resize(inv32Count()); else d_firstInv.resize(invCount);
						sideAge = overlapPeriod(); firstInvPending = true;
					}
				}
				sys.acinv = d_coarse[2].data();
				// this trial's matrix -> the inverse the next trial will use, on the other stream (after the copy above): every
				// trial for small coarse dimensions, every overlapPeriod()-th one beyond (the sweep's share of the CUs slows
				// the latency-bound PCG kernels it runs under)
				// (a run that started with the carried-over inverse inverts its first trial's matrix on the other stream IN ADDITION to
				// the regular schedule, which stays that of a run that inverted in line: repeating a run from the same estimate
				// reproduces it bit for bit)
				const bool regular = ++sideAge >= overlapPeriod();
				if (regular) sideAge = 0;
				if (regular)
				{
					HIP_TRY(hipEventRecord(evSetup, stream));
					HIP_TRY(hipStreamWaitEvent(gjStream, evSetup, 0));
					(void)launch_coarse_setup(g, st, sys, d_coarse[first].data(), d_coarse[1 - first].data(), gjStream, evAssembled);
					if (fp32Inverse()) launch_coarse_to_fp32(d_coarse[0].data(), d_coarse32[1].data(), 6 * sys.cl * sys.nc, gjStream);   // (staging: the iteration graphs read [0])
					if (firstInvPending)
					{
						if (fp32Inverse()) HIP_TRY(hipMemcpyAsync(d_firstInv32.data(), d_coarse32[1].data(), inv32Count() * sizeof(float), hipMemcpyDeviceToDevice, gjStream));
						else HIP_TRY(hipMemcpyAsync(d_firstInv.data(), d_coarse[0].data(), invBytes, hipMemcpyDeviceToDevice, gjStream));
						HIP_TRY(hipEventRecord(evFirstInv, gjStream));
						firstInvPending = false; firstInvValid = true;
					}
					HIP_TRY(hipEventRecord(evInverse, gjStream));
					pendingInv = 0;
					assemblePending = true; cntCoarseRefresh++;
				}
			}
			launch_pcg2_fused(g, sys, 0, 0, maxIter, tol2, 0, stream);
		}
		// first solve on this structure: the usual chunk lengths are ordered at once (the helper thread builds them while this solve runs
		// on plain launches), longest first -- the first batches of a run are the long ones
		if (useGraph && pcgGraphs.empty() && graphsOrderedFor != (const void*)sys.acinv)
		{
			for (int c = 64; c >= 4; c /= 2) (void)pcgGraphIfReady(c, maxIter, tol2);
			graphsOrderedFor = (const void*)sys.acinv;
		}
		// Iterations are enqueued in chunks (graphs of 4/8/.../256 iterations; chunk lengths are multiples of 4
		// because the kernels address their reduction slots by the chunk-local k & 3) and the host looks at the device
		// stop flag after each batch.  A launch after convergence still costs ~2.5 us per kernel and a look costs a
		// host round trip, so the first batch is sized from the previous solve of this run.
		volatile int* hInts = (volatile int*)((char*)h_pinned + 1024);   // fail, iterations done, stop flag: written by the device
		bool converged = false;
		int k0 = 0, looks = 0, eagerIters = 0;
		const auto tSolve0 = Clock::now();
		// prediction: within an LM run the damping shrinks geometrically and the iteration count grows by a fairly steady
		// factor from solve to solve, so extrapolate the last two counts of this run
		// (the larger of a linear and a geometric extrapolation, + 1 for the iteration in which the stop test fires: small graphs grow
		// by a few iterations per solve, large ones by a factor; every iteration enqueued past convergence costs ~5 us, a batch that
		// falls short costs a host look and is continued with a short one)
		int predicted = 32;
		if (runIters.size() >= 2)
		{
			const double a = runIters[runIters.size() - 2], b = runIters.back();
			const double lin = b + std::max(0.0, b - a), geo = b * std::min(1.35, std::max(1.0, b / std::max(1.0, a)));
			predicted = (int)std::max(lin, geo) + 1;
		}
		else if (runIters.size() == 1) predicted = (int)(1.35 * runIters[0]) + 2;
		else if (firstSolveIters > 0) predicted = firstSolveIters + 1;
		// a run that has repeated the previous run on this structure solve for solve so far (re-optimisation from the same estimate, a
		// sliding window that barely moved) most likely does so again: exactly that many iterations, no margin
		{
			const size_t k = runIters.size();
			if (repeatPrediction && k < prevRunIters.size() && std::equal(runIters.begin(), runIters.end(), prevRunIters.begin())) predicted = prevRunIters[k];
		}
		// (the last node of every iteration graph runs the stop test on the residual its chunk left: a batch of exactly the needed
		// length is recognised as converged)
		int target = (predicted + 3) / 4 * 4;
		while (k0 < maxIter && !converged)
		{
			int todo = std::max(4, std::min(target, maxIter) - k0);
			while (todo > 0)
			{
				int c = 256;
				for (; c > 4 && c > todo; c >>= 1) {}   // largest of 256, 128, ..., 4 that fits: few graphs per batch (each hand-over costs ~9 us)
				// a batch length that comes back (repeated runs on one structure) gets a graph of exactly that length: one hand-over per
				// batch instead of one per power of two.  Not on the first request -- the reference's timing protocol meets most lengths for
				// the first time inside its timed part, and an instantiation costs ~2 us per node.
				// (the exact graph is ordered on the second request and used from the moment it exists)
				if (useGraph && exactBatchGraphs && todo > c && todo % 4 == 0 && todo <= 128 && pcgGraphMaxIter == maxIter && pcgGraphTol2 == tol2)
				{
					if ((pcgGraphs.count(std::make_pair(todo, (const Scalar*)sys.acinv)) || ++batchRequests[todo] >= 2) && pcgGraphIfReady(todo, maxIter, tol2)) c = todo;
				}
				hipGraphExec_t exec = useGraph ? pcgGraphIfReady(c, maxIter, tol2) : nullptr;
				if (exec) { HIP_TRY(hipGraphLaunch(exec, stream)); noteReport(); }     // (every graph reports; the host waits for the last)
				else if (useGraph)
				{
					// the same chunk as plain launches: chunk-local iteration numbers + the advance / stop test / report node
					for (int k = 0; k < c; k++) enqueuePcgIteration(k, maxIter, tol2, stream);
					launch_pcg_advance(sys, c, stream, tol2); noteReport();
					eagerIters += c;
				}
				else for (int k = k0; k < k0 + c; k++) enqueuePcgIteration(k, maxIter, tol2, stream);
				k0 += c; todo -= c;
			}
			if (!useGraph) { launch_pcg_report(sys, stream); noteReport(); }      // (graphs and chunks of plain launches end with this report)
			waitReport();
			if (hInts[0] != 0) { lastSolveBrokeDown = true; cntPcgIters += hInts[1]; coarseValid = false; firstInvValid = false; firstInvPending = false; failDirty = true; return false; }
			if (hInts[2] != 0 || hInts[1] < std::min(k0, maxIter)) converged = true;   // the device-side stop test fired
			target = k0 + ((looks == 0 && k0 <= 96) ? 4 : std::max(8, k0 / 8 / 4 * 4));   // (a batch sized from the run's own history misses by a few iterations at most)
			looks++; cntPcgLooks++;
		}
		const auto tSolve1 = Clock::now();
		if (std::getenv("CUBA_HIP_DEBUG"))
		{
			// (debug only: r_0.z_0 of this solve from the partial sums the first preconditioner application left in slot 0)
			std::vector<Scalar> part((size_t)std::max(1, sys.nrz0));
			HIP_TRY(hipMemcpyAsync(part.data(), sys.rz, sizeof(Scalar) * part.size(), hipMemcpyDeviceToHost, stream));
			sync();
			double rz0 = 0; for (Scalar v : part) rz0 += (double)v;
			std::fprintf(stderr, "[cuba_hip] PCG: %d iterations, %d enqueued (%d as plain launches), %d host looks (prediction %d), lambda %.3e, r0.z0 %.6e, solve %.3f ms, graphs built so far %lld\n",
				hInts[1], k0, eagerIters, looks, predicted, lambda, rz0, 1e3 * std::chrono::duration<double>(tSolve1 - tSolve0).count(), (long long)gb.builds.load());
		}
		const int itersDone = hInts[1];
		cntPcgIters += itersDone; cntPcgEnqueued += k0;
		if (runIters.empty()) firstSolveIters = itersDone;
		runIters.push_back(itersDone);
		if (pcgHistory.size() >= 65536) pcgHistory.erase(pcgHistory.begin(), pcgHistory.begin() + 32768);   // drivers that never call set_graph again: keep the latest
		pcgHistory.push_back(converged ? itersDone : -itersDone);
		if (!converged)
		{
			// max_iter reached with the stop test still unsatisfied: never silent.  Reported like the reference's
			// failed factorisation (src/cuda_linear_solver.cpp:406-410 -> CudaBlockSolver::solve returns false ->
			// the LM loop rejects the trial and raises lambda, which also makes the next system easier), unless the
			// caller asked for the best iterate ("pcg_accept_unconverged").
			cntPcgUnconverged++;
			char buf[160];
			std::snprintf(buf, sizeof buf, "PCG stopped at max_iter = %d without reaching pcg_tol = %g", maxIter, pcgTol);
			lastError = buf;
			coarseValid = false;
			return acceptUnconverged;
		}
		return true;
	}

	void backSubstitute()
	{
		need();
		StageTimer tm(this, 4);
		launch_back_substitute(g, st, sys, lambda, stream);
	}

	bool solve()
	{
		schur();
		if (!solveReduced()) return false;
		backSubstitute();
		return true;
	}

	void update()
	{
		need();
		StageTimer tm(this, 7);
		launch_update_state(g, sys, stream);
	}

	// Stage-API version of sum x (lambda x + b): recomputed from xp/bp and xl/bl, valid for any lambda.
	double computeScale(double lam)
	{
		need();
		double a = 0, b = 0;
		scaleParts(lam, &a, &b);
		return a + b;
	}

	void push() { need(); HIP_TRY(hipMemcpyAsync(d_backup.data(), d_state.data(), d_state.size() * sizeof(Scalar), hipMemcpyDeviceToDevice, stream)); }
	void pop() { need(); HIP_TRY(hipMemcpyAsync(d_state.data(), d_backup.data(), d_state.size() * sizeof(Scalar), hipMemcpyDeviceToDevice, stream)); }

	// Levenberg-Marquardt, control flow of CudaBundleAdjustmentImpl::optimize (:793-857).
	int optimize(int niter, double* chi2Out)
	{
		lap(nullptr);
		need();
		lap("optimize: structure ready");
		coarseValid = false;          // a new LM run starts from a new lambda_0: never reuse the coarse inverse across runs
		startRunHistory();
		const int maxq = 10;
		const double tau = 1e-5;
		double nu = 2, lam = 0, F = 0;
		int done = 0;
		bool haveF = false;           // after an accepted step the objective at the new estimate is already known
		for (int it = 0; it < niter; it++)
		{
			if (!haveF) F = computeErrors();
			if (it == 0) { lam = tau * maxDiagonal(); lap("optimize: chi2 + max diagonal"); }
			int qn = 0;
			double rho = -1;
			for (; qn < maxq && rho < 0; qn++)
			{
				cntTrials++;
				lambda = lam;
				schur(true);          // (with the push() of the reference's loop: the backup of the state rides in the landmark pass's launch)
				const bool ok = solveReduced();
				double Fhat = 0, scale = 0;
				if (ok && !profile && partHi < 0 && fusedTail && trial_tail_parts(g, st) <= d_parts.size())
				{
					// back-substitution + update + evaluation in one pass over the edges (reads the pre-trial estimate from the backup
					// schur(true) has just made), then sums + report: two launches, one host look
					launch_trial_tail_fused(g, st, sys, (Scalar)lam, d_backup.data(), stream); noteReport();
					readEvaluate(true, &Fhat, &scale);
				}
				else if (ok && !profile && partHi < 0 && (size_t)st.nWaves + st.nBig + 64 + 3072 <= d_parts.size())
				{
					// back-substitution, update, evaluation, sums and report in four launches, one host look
					launch_trial_tail(g, st, sys, (Scalar)lam, stream); noteReport();
					readEvaluate(true, &Fhat, &scale);
				}
				else
				{
					if (ok) { backSubstitute(); update(); }
					evaluateTrial(lam, ok, &Fhat, &scale);                  // chi2 at the trial estimate + gain-ratio denominator, one host look
				}
				scale += 1e-3;
				rho = ok ? (F - Fhat) / scale : -1;
				if (rho > 0)
				{
					const double a = 1 - std::pow(2 * rho - 1, 3);
					lam *= std::max(1. / 3, std::min(a, 2. / 3));
					nu = 2;
					F = Fhat;
					haveF = true;
					break;
				}
				else
				{
					lam *= nu;
					nu *= 2;
					pop();
					haveF = true;      // F still describes the restored estimate
				}
			}
			if (chi2Out) chi2Out[it] = F;
			done = it + 1;
			lap("optimize: LM iteration");
			(void)hipStreamQuery(stream);      // non-blocking; the ticket waits never enter the runtime, this lets it retire finished commands
			if (qn == maxq || rho <= 0 || !std::isfinite(lam)) break;
		}
		lambda = lam;
		return done;
	}

	// chi2 of the trial estimate and sum x (lambda x + b) of the step that led to it, read back with ONE synchronisation
	void enqueueEvaluate(double lam, bool withScale)
	{
		launch_residual_chi2(g, d_parts.data(), slotsDev, nullptr, stream);
		if (withScale) launch_pose_scale(g, sys, lam, slotsDev + 3 * NSLOT, stream);
		launch_pcg_report(sys, stream); noteReport();       // ticket behind the results (which the kernels wrote into the mapped host block)
	}
	void readEvaluate(bool withScale, double* Fhat, double* scale)
	{
		waitReport();
		*Fhat = (double)slot(0);
		*scale = withScale ? (double)slot(NSLOT) + (double)slot(3 * NSLOT) : 0.0;   // landmark part (back_substitute) + pose part
	}
	void evaluateTrial(double lam, bool withScale, double* Fhat, double* scale)
	{
		StageTimer tm(this, 2);
		enqueueEvaluate(lam, withScale);
		readEvaluate(withScale, Fhat, scale);
	}

	// Fused version used by optimize(): the landmark part was accumulated by back_substitute (same lambda),
	// only the 6*Pf pose part is added here.
	double scaleOfLastSolve(double lam)
	{
		launch_pose_scale(g, sys, lam, slotsDev + 3 * NSLOT, stream);
		sync();
		double v = 0;
		for (int i = 0; i < NSLOT; i++) v += slot(NSLOT + i) + slot(3 * NSLOT + i);
		return v;
	}

	// Average device time per launch of the five hot kernels, measured with HIP events on this solver's
	// stream (bench.py's roofline leg).  Leaves the increments / reduced system in an undefined state.
	void timeKernels(int reps, double* msOut)
	{
		need();
		hipEvent_t e0, e1;
		HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
		// `reps` launches are captured into one hipGraph so that the events bracket device time, not the host's
		// launch cadence (plain launches of these few-microsecond kernels are host-bound)
		hipStream_t work = stream;
		auto timeit = [&](auto&& fn) {
			fn();
			hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
			hipStream_t cs = capStream();
			HIP_TRY(hipStreamSynchronize(work));
			HIP_TRY(hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed));
			stream = cs;                    // the launch helpers below enqueue on `stream`
			for (int i = 0; i < reps; i++) fn();
			stream = work;
			HIP_TRY(hipStreamEndCapture(cs, &graph));
			HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
			HIP_TRY(hipGraphLaunch(exec, stream));
			HIP_TRY(hipEventRecord(e0, stream));
			HIP_TRY(hipGraphLaunch(exec, stream));
			HIP_TRY(hipEventRecord(e1, stream));
			HIP_TRY(hipEventSynchronize(e1));
			float ms = 0;
			HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
			(void)hipGraphExecDestroy(exec); (void)hipGraphDestroy(graph);
			return (double)ms / reps;
		};
		const double lam = lambda > 0 ? lambda : 1.0;
		msOut[0] = timeit([&] { launch_residual_chi2(g, d_parts.data(), slotsDev, nullptr, stream); });
		zeroReduced();
		msOut[1] = timeit([&] { linearize(1, lam); });
		// a consistent reduced system for the PCG kernels
		zeroReduced();
		linearize(1, lam);
		d_fail.zero(stream);
		d_kbase.zero(stream);
		launch_pcg_setup(g, st, sys, lam, stream);
		launch_hsc_expand(g, st, sys, stream);
		if (sys.agg > 0)
		{
			drainInversion();
			sys.acinv = launch_coarse_setup(g, st, sys, d_coarse[0].data(), d_coarse[1].data(), stream);
			if (fp32Inverse()) launch_coarse_to_fp32(sys.acinv, d_coarse32[0].data(), 6 * sys.cl * sys.nc, stream);
			coarseValid = false;
			launch_pcg2_fused(g, sys, 0, 0, 1 << 30, -1.0, 0, stream);
		}
		msOut[2] = timeit([&] { launch_pcg_spmv(g, st, sys, 0, 1 << 30, -1.0, stream); });
		if (sys.agg > 0)
		{
			msOut[3] = timeit([&] { launch_pcg2_fused(g, sys, 0, 1, 1 << 30, -1.0, 1, stream); });
			msOut[5] = 0;
			msOut[6] = timeit([&] { launch_coarse_setup(g, st, sys, d_coarse[0].data(), d_coarse[1].data(), stream); });
		}
		else
		{
			msOut[3] = timeit([&] { launch_pcg_update(g, st, sys, 0, 1 << 30, -1.0, stream); });
			msOut[5] = 0; msOut[6] = 0;
		}
		msOut[4] = timeit([&] { launch_back_substitute(g, st, sys, lam, stream); });
		d_fail.zero(stream); failDirty = true;
		(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	}

	void chiSquares(double* out, bool wait = true)
	{
		need();
		launch_residual_chi2(g, d_parts.data(), slotsDev + 2 * NSLOT, d_perEdge.data(), stream);
		if (devTopology)
		{
			// sorted order -> the caller's order on the device (the permutation never left it)
			d_chiCaller.resize(E);
			topo::launch_unsort(d_perm.data(), d_perEdge.data(), E, d_chiCaller.data(), stream);
			if (E) HIP_TRY(hipMemcpyAsync(out, d_chiCaller.data(), sizeof(double) * E, hipMemcpyDeviceToHost, stream));
			if (wait) sync();
			return;
		}
		std::vector<double>& sorted = h_chiSorted; sorted.resize(E);
		downloadAsDouble(d_perEdge.data(), sorted.data(), (size_t)E);
		sync();
		parallelFor(E, [&](int i) { out[perm[i]] = sorted[i]; });     // back to the caller's edge order
	}
};

// -----------------------------------------------------------------------------------------------------
// C ABI
// -----------------------------------------------------------------------------------------------------
namespace
{
std::string g_createError;

template <typename F>
int guarded(cuba_hip_solver* s, F&& f)
{
	if (!s) return CUBA_HIP_ERR_INVALID_ARGUMENT;
	try
	{
		if (hipSetDevice(s->device) != hipSuccess) { s->lastError = "hipSetDevice failed"; return CUBA_HIP_ERR_RUNTIME; }
		f();
		return CUBA_HIP_OK;
	}
	catch (const HipError& e)
	{
		char buf[512];
		std::snprintf(buf, sizeof buf, "HIP error %d (%s) in `%s` at %s:%d", (int)e.code, hipGetErrorString(e.code), e.what, e.file, e.line);
		s->lastError = buf;
		return CUBA_HIP_ERR_RUNTIME;
	}
	catch (const StateError& e) { s->lastError = e.msg; return CUBA_HIP_ERR_STATE; }
	catch (const ArgError& e) { s->lastError = e.msg; return CUBA_HIP_ERR_INVALID_ARGUMENT; }
	catch (const std::exception& e) { s->lastError = e.what(); return CUBA_HIP_ERR_RUNTIME; }
}
}  // namespace

namespace
{
std::mutex g_pinMutex;
std::vector<void*> g_pinned;      // blocks handed out by cuba_hip_host_alloc that really are page-locked
}

extern "C" {

void* cuba_hip_host_alloc(size_t bytes)
{
	void* p = nullptr;
	int n = 0;
	if (bytes && hipGetDeviceCount(&n) == hipSuccess && n > 0 && hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess && p)
	{
		std::lock_guard<std::mutex> lk(g_pinMutex);
		g_pinned.push_back(p);
		return p;
	}
	(void)hipGetLastError();
	return std::malloc(bytes ? bytes : 1);
}

void cuba_hip_host_free(void* p)
{
	if (!p) return;
	{
		std::lock_guard<std::mutex> lk(g_pinMutex);
		auto it = std::find(g_pinned.begin(), g_pinned.end(), p);
		if (it != g_pinned.end()) { g_pinned.erase(it); (void)hipHostFree(p); return; }
	}
	std::free(p);
}

const char* cuba_hip_version(void) { return sizeof(Scalar) == 8 ? "cuba-hip 0.1 (gfx950, fp64)" : "cuba-hip 0.1 (gfx950, fp32)"; }
int cuba_hip_scalar_size(void) { return (int)sizeof(Scalar); }

int cuba_hip_create(int device, cuba_hip_solver** out)
{
	if (!out) return CUBA_HIP_ERR_INVALID_ARGUMENT;
	*out = nullptr;
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return CUBA_HIP_ERR_NO_DEVICE;
	if (device < 0 || device >= n) return CUBA_HIP_ERR_INVALID_ARGUMENT;
	if (hipSetDevice(device) != hipSuccess) return CUBA_HIP_ERR_RUNTIME;
	cuba_hip_solver* s = new (std::nothrow) cuba_hip_solver;
	if (!s) return CUBA_HIP_ERR_RUNTIME;
	s->device = device;
	if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) { delete s; return CUBA_HIP_ERR_RUNTIME; }
	s->ownStream = true;
	*out = s;
	return CUBA_HIP_OK;
}

int cuba_hip_destroy(cuba_hip_solver* s)
{
	if (!s) return CUBA_HIP_ERR_INVALID_ARGUMENT;
	(void)hipSetDevice(s->device);
	(void)hipStreamSynchronize(s->stream);
	delete s;
	return CUBA_HIP_OK;
}

const char* cuba_hip_last_error(const cuba_hip_solver* s) { return s ? s->lastError.c_str() : "null solver handle"; }

int cuba_hip_set_stream(cuba_hip_solver* s, void* hip_stream)
{
	return guarded(s, [&] {
		s->sync();
		if (s->ownStream && s->stream) { HIP_TRY(hipStreamDestroy(s->stream)); s->ownStream = false; }
		s->stream = (hipStream_t)hip_stream;
	});
}

int cuba_hip_set_option(cuba_hip_solver* s, const char* key, double value)
{
	return guarded(s, [&] {
		if (!key) throw ArgError{ "null key" };
		const std::string k(key);
		if (k == "pcg_tol") s->pcgTol = value;
		else if (k == "pcg_max_iter") { s->pcgMaxIter = (int)value; s->haveStructure = false; }
		else if (k == "fused_tail") s->fusedTail = value != 0;
		else if (k == "pcg_exact_batch_graphs") s->exactBatchGraphs = value != 0;
		else if (k == "pcg_repeat_prediction") s->repeatPrediction = value != 0;
		else if (k == "coarse_first_reuse") { s->coarseFirstReuse = value != 0; s->firstInvValid = false; s->firstInvPending = false; }
		else if (k == "precond_fp32") { s->precondFp32 = value != 0; s->haveStructure = false; s->coarseValid = false; s->dropPcgGraph(); }
		else if (k == "pose_reorder") { s->poseReorder = value != 0; s->haveStructure = false; }
		else if (k == "device_setup") { s->deviceSetup = value != 0; s->haveStructure = false; }
		else if (k == "mixed_precision") s->mixedPrecision = value != 0 && sizeof(Scalar) == 8;
		else if (k == "pcg_accept_unconverged") s->acceptUnconverged = value != 0;
		else if (k == "pcg_aggregate") { s->pcgAggregate = (int)value; s->haveStructure = false; }
		else if (k == "coarse_linear") { s->coarseLinear = value != 0; s->haveStructure = false; }
		else if (k == "pcg_graph") { s->useGraph = value != 0; s->dropPcgGraph(); }
		else if (k == "profile") s->profile = value != 0;
		else throw ArgError{ "unknown option: " + k };
	});
}


int cuba_hip_hint_unchanged(cuba_hip_solver* s, int same_edges, int same_values)
{
	return guarded(s, [&] { s->hintSameEdges = same_edges != 0; s->hintSameValues = same_edges != 0 && same_values != 0; });
}

int cuba_hip_set_graph(cuba_hip_solver* s, int Pt, int Pf, int Lt, int Lf, const double* q, const double* t, const double* cam,
	const double* Xw, int E, const int32_t* edge_pose, const int32_t* edge_landmark, const uint8_t* edge_dim,
	const double* meas, const double* omega)
{
	return guarded(s, [&] { s->setGraph(Pt, Pf, Lt, Lf, q, t, cam, Xw, E, edge_pose, edge_landmark, edge_dim, meas, omega); });
}

int cuba_hip_set_graph_begin(cuba_hip_solver* s, int Pt, int Pf, int Lt, int Lf, const double* q, const double* t, const double* cam,
	const double* Xw, int E, const int32_t* edge_pose, const int32_t* edge_landmark, const uint8_t* edge_dim,
	const double* meas, const double* omega)
{
	return guarded(s, [&] { s->setGraph(Pt, Pf, Lt, Lf, q, t, cam, Xw, E, edge_pose, edge_landmark, edge_dim, meas, omega, true); });
}

int cuba_hip_set_graph_end(cuba_hip_solver* s)
{
	return guarded(s, [&] {
		if (!s->haveGraph) throw StateError{ "set_graph_begin must be called first" };
		if (s->upStream) HIP_TRY(hipStreamSynchronize(s->upStream));      // the caller's meas / omega are free again
		s->finishValues();
	});
}

int cuba_hip_set_robust_kernel(cuba_hip_solver* s, int edge_type, int kind, double delta)
{
	return guarded(s, [&] {
		if (edge_type < 0 || edge_type > 1 || kind < 0 || kind > 2) throw ArgError{ "bad robust kernel" };
		s->rk[edge_type] = RobustKernel{ kind, (Scalar)delta };
	});
}

int cuba_hip_build_structure(cuba_hip_solver* s) { return guarded(s, [&] { s->buildStructure(); s->sync(); }); }

int cuba_hip_compute_errors(cuba_hip_solver* s, double* chi2)
{
	return guarded(s, [&] { if (!chi2) throw ArgError{ "null output" }; *chi2 = s->computeErrors(); });
}

int cuba_hip_build_system(cuba_hip_solver* s) { return guarded(s, [&] { s->need(); }); }

int cuba_hip_max_diagonal(cuba_hip_solver* s, double* v)
{
	return guarded(s, [&] { if (!v) throw ArgError{ "null output" }; *v = s->maxDiagonal(); });
}

int cuba_hip_set_lambda(cuba_hip_solver* s, double lambda) { return guarded(s, [&] { s->lambda = lambda; }); }
int cuba_hip_restore_diagonal(cuba_hip_solver* s) { return guarded(s, [&] { s->lambda = 0; }); }

int cuba_hip_schur(cuba_hip_solver* s) { return guarded(s, [&] { s->schur(); }); }

int cuba_hip_solve_reduced(cuba_hip_solver* s, int* ok)
{
	return guarded(s, [&] { const bool r = s->solveReduced(); if (ok) *ok = r ? 1 : 0; });
}

int cuba_hip_back_substitute(cuba_hip_solver* s) { return guarded(s, [&] { s->backSubstitute(); }); }

int cuba_hip_solve(cuba_hip_solver* s, int* ok)
{
	return guarded(s, [&] { const bool r = s->solve(); if (ok) *ok = r ? 1 : 0; });
}

int cuba_hip_update(cuba_hip_solver* s) { return guarded(s, [&] { s->update(); }); }

int cuba_hip_compute_scale(cuba_hip_solver* s, double lambda, double* scale)
{
	return guarded(s, [&] {
		if (!scale) throw ArgError{ "null output" };
		*scale = s->computeScale(lambda);
	});
}

int cuba_hip_snapshot_state_slot(cuba_hip_solver* s, int slot)
{
	return guarded(s, [&] {
		if (slot < 0 || slot >= CUBA_HIP_SNAPSHOT_SLOTS) throw ArgError{ "snapshot slot out of range" };
		s->need();
		DevBuf<Scalar>& b = s->d_snapshots[slot];
		b.resize(s->d_state.size());
		HIP_TRY(hipMemcpyAsync(b.data(), s->d_state.data(), s->d_state.size() * sizeof(Scalar), hipMemcpyDeviceToDevice, s->stream));
	});
}

int cuba_hip_restore_state_slot(cuba_hip_solver* s, int slot)
{
	return guarded(s, [&] {
		s->need();
		const auto it = s->d_snapshots.find(slot);
		if (it == s->d_snapshots.end() || it->second.size() != s->d_state.size()) throw StateError{ "no snapshot of this graph's estimates in that slot (cuba_hip_snapshot_state)" };
		HIP_TRY(hipMemcpyAsync(s->d_state.data(), it->second.data(), s->d_state.size() * sizeof(Scalar), hipMemcpyDeviceToDevice, s->stream));
	});
}

int cuba_hip_snapshot_state(cuba_hip_solver* s) { return cuba_hip_snapshot_state_slot(s, 0); }
int cuba_hip_restore_state(cuba_hip_solver* s) { return cuba_hip_restore_state_slot(s, 0); }

int cuba_hip_push(cuba_hip_solver* s) { return guarded(s, [&] { s->push(); }); }
int cuba_hip_pop(cuba_hip_solver* s) { return guarded(s, [&] { s->pop(); }); }

int cuba_hip_optimize(cuba_hip_solver* s, int niterations, double* chi2_per_iter, int* n_done)
{
	return guarded(s, [&] {
		if (niterations < 0) throw ArgError{ "negative iteration count" };
		const int n = s->optimize(niterations, chi2_per_iter);
		if (n_done) *n_done = n;
	});
}

int cuba_hip_get_solution(cuba_hip_solver* s, double* q, double* t, double* Xw)
{
	return guarded(s, [&] {
		if (!s->haveGraph) throw StateError{ "set_graph must be called first" };
		const Scalar* base = s->d_state.data();
		if (q) { s->downloadAsDouble(base, q, (size_t)4 * s->Pt); s->permutePoseArray(q, 4, false); }
		if (t) { s->downloadAsDouble(base + 4 * (size_t)s->Pt, t, (size_t)3 * s->Pt); s->permutePoseArray(t, 3, false); }
		if (Xw) s->downloadAsDouble(base + 7 * (size_t)s->Pt, Xw, (size_t)3 * s->Lt);
	});
}

int cuba_hip_set_solution(cuba_hip_solver* s, const double* q, const double* t, const double* Xw)
{
	return guarded(s, [&] {
		if (!s->haveGraph) throw StateError{ "set_graph must be called first" };
		s->buildStructure();          // the internal pose order (if any) is decided there
		Scalar* base = s->d_state.data();
		if (q) { std::vector<double> v(q, q + (size_t)4 * s->Pt); s->permutePoseArray(v.data(), 4, true); s->uploadFromDouble(base, v.data(), v.size()); }
		if (t) { std::vector<double> v(t, t + (size_t)3 * s->Pt); s->permutePoseArray(v.data(), 3, true); s->uploadFromDouble(base + 4 * (size_t)s->Pt, v.data(), v.size()); }
		if (Xw) s->uploadFromDouble(base + 7 * (size_t)s->Pt, Xw, (size_t)3 * s->Lt);
	});
}

int cuba_hip_chi_squares(cuba_hip_solver* s, double* out)
{
	return guarded(s, [&] { if (!out && s->E) throw ArgError{ "null output" }; s->chiSquares(out); });
}

int cuba_hip_chi_squares_begin(cuba_hip_solver* s, double* out)
{
	return guarded(s, [&] { if (!out && s->E) throw ArgError{ "null output" }; s->chiSquares(out, false); });     // (the host set-up path has finished when this returns)
}

int cuba_hip_chi_squares_end(cuba_hip_solver* s)
{
	return guarded(s, [&] { s->sync(); });
}

int cuba_hip_get_profile(cuba_hip_solver* s, double seconds[CUBA_HIP_PROFILE_ITEMS])
{
	return guarded(s, [&] { for (int i = 0; i < CUBA_HIP_PROFILE_ITEMS; i++) seconds[i] = s->prof[i]; });
}

int cuba_hip_get_counters(cuba_hip_solver* s, int64_t c[8])
{
	return guarded(s, [&] {
		c[0] = s->cntPcgIters; c[1] = s->cntTrials; c[2] = s->st.nblk; c[3] = s->nmul;
		c[4] = s->cntCoarseRefresh; c[5] = s->cntPcgLooks; c[6] = s->cntPcgEnqueued; c[7] = 6 * (int64_t)s->sys.cl * s->sys.nc;
	});
}

int cuba_hip_get_counter(cuba_hip_solver* s, const char* name, int64_t* value)
{
	return guarded(s, [&] {
		if (!name || !value) throw ArgError{ "null argument" };
		const std::string k(name);
		if (k == "pcg_iterations") *value = s->cntPcgIters;
		else if (k == "lm_trials") *value = s->cntTrials;
		else if (k == "coarse_refreshes") *value = s->cntCoarseRefresh;
		else if (k == "coarse_inline_inversions") *value = s->cntCoarseInline;
		else if (k == "pcg_host_looks") *value = s->cntPcgLooks;
		else if (k == "pcg_iterations_enqueued") *value = s->cntPcgEnqueued;
		else if (k == "pcg_unconverged_solves") *value = s->cntPcgUnconverged;
		else if (k == "pcg_graph_instantiations") *value = s->gb.builds.load();
		else if (k == "precond_fp32_fallbacks") *value = s->cntFp32Fallbacks;
		else if (k == "graph_uploads") *value = s->cntUploads;
		else throw ArgError{ "unknown counter: " + k };
	});
}

int cuba_hip_get_pcg_history(cuba_hip_solver* s, int32_t* iterations, int capacity, int* n_solves, int64_t* n_unconverged)
{
	return guarded(s, [&] {
		const int n = (int)s->pcgHistory.size();
		if (n_solves) *n_solves = n;
		if (n_unconverged) *n_unconverged = s->cntPcgUnconverged;
		if (iterations) for (int i = 0; i < std::min(n, capacity); i++) iterations[i] = s->pcgHistory[i];
	});
}

int cuba_hip_get_hsc_structure(cuba_hip_solver* s, int32_t* row_ptr, int32_t* col_ind, int* nblk)
{
	return guarded(s, [&] {
		s->need();
		s->ensureHostPattern();
		if (nblk) *nblk = (int)s->h_colind.size();
		if (!s->reorderActive)
		{
			if (row_ptr) std::memcpy(row_ptr, s->h_rowptr.data(), sizeof(int) * s->h_rowptr.size());
			if (col_ind) std::memcpy(col_ind, s->h_colind.data(), sizeof(int) * s->h_colind.size());
			return;
		}
		const auto blocks = s->callerBlocks();
		if (row_ptr)
		{
			for (int i = 0; i <= s->Pf; i++) row_ptr[i] = 0;
			for (const auto& b : blocks) row_ptr[(int)(b.key >> 32) + 1]++;
			for (int i = 0; i < s->Pf; i++) row_ptr[i + 1] += row_ptr[i];
		}
		if (col_ind) for (size_t k = 0; k < blocks.size(); k++) col_ind[k] = (int)(uint32_t)blocks[k].key;
	});
}

int cuba_hip_get_array(cuba_hip_solver* s, int which, double* out, size_t* count)
{
	return guarded(s, [&] {
		s->need();
		const Scalar* src = nullptr; size_t n = 0;
		switch (which)
		{
		case CUBA_HIP_ARRAY_BP: src = s->sys.bp; n = (size_t)6 * s->Pf; break;
		case CUBA_HIP_ARRAY_BSC: src = s->sys.bsc; n = (size_t)6 * s->Pf; break;
		case CUBA_HIP_ARRAY_XP: src = s->sys.xp; n = (size_t)6 * s->Pf; break;
		case CUBA_HIP_ARRAY_XL: src = s->sys.xl; n = (size_t)3 * s->Lf; break;
		case CUBA_HIP_ARRAY_LM_SYS: src = s->sys.lm_sys; n = (size_t)9 * s->Lf; break;
		case CUBA_HIP_ARRAY_HSC: src = s->sys.hsc; n = (size_t)36 * s->st.nblk; break;
		default: throw ArgError{ "unknown array id" };
		}
		if (count) *count = n;
		if (out && n)
		{
			s->downloadAsDouble(src, out, n);
			if (s->reorderActive)
			{
				// back to the caller's pose numbering
				if (which == CUBA_HIP_ARRAY_BP || which == CUBA_HIP_ARRAY_BSC || which == CUBA_HIP_ARRAY_XP) s->permutePoseArray(out, 6, false);
				else if (which == CUBA_HIP_ARRAY_HSC)
				{
					const std::vector<double> v(out, out + n);
					const auto blocks = s->callerBlocks();
					for (size_t k = 0; k < blocks.size(); k++)
					{
						const double* src36 = v.data() + 36 * (size_t)blocks[k].src;
						double* dst = out + 36 * k;
						for (int c = 0; c < 6; c++)
							for (int r = 0; r < 6; r++) dst[c * 6 + r] = blocks[k].transposed ? src36[r * 6 + c] : src36[c * 6 + r];
					}
				}
			}
		}
	});
}

int cuba_hip_time_kernels(cuba_hip_solver* s, int reps, double ms_per_launch[CUBA_HIP_TIMED_KERNELS])
{
	return guarded(s, [&] {
		if (reps <= 0 || !ms_per_launch) throw ArgError{ "bad arguments" };
		s->timeKernels(reps, ms_per_launch);
	});
}

int cuba_hip_set_partition(cuba_hip_solver* s, int landmark_begin, int landmark_end)
{
	return guarded(s, [&] {
		if (!s->haveGraph) throw StateError{ "set_graph must be called first" };
		if (landmark_begin == 0 && landmark_end == -1)          // remove the restriction: the handle evaluates the whole graph again
		{
			if (s->partHi >= 0) s->haveStructure = false;
			s->partLo = 0; s->partHi = -1;
			return;
		}
		if (landmark_begin < 0 || landmark_end > s->Lt || landmark_begin > landmark_end) throw ArgError{ "bad landmark range" };
		s->partLo = landmark_begin; s->partHi = landmark_end;
		s->haveStructure = false;
	});
}

int cuba_hip_assemble(cuba_hip_solver* s) { return guarded(s, [&] { s->assemble(); }); }

int cuba_hip_max_diagonal_parts(cuba_hip_solver* s, double* pose_part, double* landmark_part)
{
	return guarded(s, [&] { if (!pose_part || !landmark_part) throw ArgError{ "null output" }; s->maxDiagonalParts(pose_part, landmark_part); });
}

int cuba_hip_compute_scale_parts(cuba_hip_solver* s, double lambda, double* pose_part, double* landmark_part)
{
	return guarded(s, [&] { if (!pose_part || !landmark_part) throw ArgError{ "null output" }; s->scaleParts(lambda, pose_part, landmark_part); });
}

int cuba_hip_device_pointer(cuba_hip_solver* s, int which, void** device_ptr, size_t* count)
{
	return guarded(s, [&] {
		s->need();
		Scalar* p = nullptr; size_t n = 0;
		switch (which)
		{
		case CUBA_HIP_ARRAY_BP: p = s->sys.bp; n = (size_t)6 * s->Pf; break;
		case CUBA_HIP_ARRAY_BSC: p = s->sys.bsc; n = (size_t)6 * s->Pf; break;
		case CUBA_HIP_ARRAY_XP: p = s->sys.xp; n = (size_t)6 * s->Pf; break;
		case CUBA_HIP_ARRAY_XL: p = s->sys.xl; n = (size_t)3 * s->Lf; break;
		case CUBA_HIP_ARRAY_LM_SYS: p = s->sys.lm_sys; n = (size_t)9 * s->Lf; break;
		case CUBA_HIP_ARRAY_HSC: p = s->sys.hsc; n = (size_t)36 * s->st.nblk; break;
		case CUBA_HIP_ARRAY_STATE: p = s->d_state.data(); n = s->d_state.size(); break;
		default: throw ArgError{ "unknown array id" };
		}
		if (device_ptr) *device_ptr = p;
		if (count) *count = n;
	});
}

int cuba_hip_get_sizes(cuba_hip_solver* s, int sizes[5])
{
	return guarded(s, [&] {
		if (!s->haveGraph) throw StateError{ "set_graph must be called first" };
		sizes[0] = s->Pt; sizes[1] = s->Pf; sizes[2] = s->Lt; sizes[3] = s->Lf; sizes[4] = s->E;
	});
}

int cuba_hip_debug_dense_inverse(int device, int n, const double* A, double* Ainv)
{
	if (n <= 0 || !A || !Ainv) return CUBA_HIP_ERR_INVALID_ARGUMENT;
	if (hipSetDevice(device) != hipSuccess) return CUBA_HIP_ERR_NO_DEVICE;
	try
	{
		const size_t nn = (size_t)n * n;
		std::vector<Scalar> h(A, A + nn);
		DevBuf<Scalar> w0, w1;
		w0.upload(h, nullptr); w1.resize(nn);
		DevBuf<Scalar> piv; piv.resize(2 * 32 * 32);
		Scalar* res = launch_dense_inverse(w0.data(), w1.data(), n, piv.data(), nullptr);
		HIP_TRY(hipMemcpy(h.data(), res, sizeof(Scalar) * nn, hipMemcpyDeviceToHost));
		for (size_t i = 0; i < nn; i++) Ainv[i] = (double)h[i];
		return CUBA_HIP_OK;
	}
	catch (const HipError&) { return CUBA_HIP_ERR_RUNTIME; }
}

int cuba_hip_begin_run(cuba_hip_solver* s)
{
	return guarded(s, [&] { s->need(); s->coarseValid = false; s->startRunHistory(); });
}

int cuba_hip_get_stream(cuba_hip_solver* s, void** hip_stream)
{
	return guarded(s, [&] { if (!hip_stream) throw ArgError{ "null output" }; *hip_stream = (void*)s->stream; });
}

int cuba_hip_evaluate_device(cuba_hip_solver* s, double lambda, int with_scale, void** device_scalars3)
{
	return guarded(s, [&] {
		s->need();
		if (!device_scalars3) throw ArgError{ "null output" };
		s->d_eval.resize(4);
		launch_residual_chi2(s->g, s->d_parts.data(), s->slotsDev, nullptr, s->stream);
		if (with_scale) launch_pose_scale(s->g, s->sys, lambda, s->slotsDev + 3 * NSLOT, s->stream);
		launch_collect_eval(s->sys, s->d_eval.data(), s->stream);
		*device_scalars3 = s->d_eval.data();
	});
}

int cuba_hip_reduction_buffer(cuba_hip_solver* s, void** device_ptr, size_t* count)
{
	return guarded(s, [&] {
		s->need();
		if (device_ptr) *device_ptr = s->d_red.data();
		if (count) *count = s->d_red.size();
	});
}

}  // extern "C"
