#pragma once
// ba_solver.hpp -- the solver handle behind the C ABI: one bundle-adjustment problem on one MI355X (state, options, device buffers).
// Implemented in ba_setup.hip (graph upload + structure analysis), ba_lm.hip (stages, reduced solve, Levenberg-Marquardt loop) and
// ba_solver.hip (the C ABI itself).
//
// (was) ba_solver.hip -- host orchestration of one bundle-adjustment problem on one MI355X + the C ABI
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

// (a named namespace: the handle is one type across the translation units that implement it, and an exception thrown in one of
// them is caught by the C ABI in another)
namespace cubahip_host
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
			// (debugging aid, CUBA_HIP_POISON=1: fresh device memory is filled with 0x7f bytes -- 1.4e306 as a double, 3.4e38 as a float, 2^31 - 8 * 2^20 as an index (finite, so that a masked 0 x garbage stays 0) --
			// so that a read of memory nobody wrote shows at once instead of depending on what the allocator hands back)
			// (CUBA_HIP_POISON=2: a different finite byte pattern for every allocation, so that two handles that should agree bit for bit
			// stop doing so if either one reads memory it did not write)
			static const int poison = std::getenv("CUBA_HIP_POISON") ? std::max(1, std::atoi(std::getenv("CUBA_HIP_POISON"))) : 0;
			static std::atomic<unsigned> poisonCount{ 0 };
			static const int pattern[4] = { 0x3f, 0x40, 0x3e, 0x41 };
			if (n && poison) { HIP_TRY(hipMemset(ptr_, poison >= 2 ? pattern[poisonCount++ & 3] : 0x7f, n * sizeof(T))); HIP_TRY(hipDeviceSynchronize()); }     // (the handle's streams do not wait for the null stream)
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

// Solver handles alive in this process.  With more than one, no handle instantiates or replays hipGraphs: every PCG batch goes out as
// plain launches of the same kernels with the same arguments (bit-identical results, ~2 % slower for a handle that runs alone).
// Measured on this runtime (profiles/r05c_* ... r05l_*): dependent-kernel chains of two streams overlap perfectly, two KITTI-00 graphs
// optimised side by side from two host threads take 8.9 / 8.9 ms per run (6.8 alone: 1.5 x the throughput) -- but only in a process that
// has NEVER instantiated a hipGraph.  Once one handle has (even if it destroyed them since, and whether or not anything is launched as a
// graph any more) the same pair takes 15.4 / 11.7 ms: no overlap at all.  So: a process that will drive several handles at once starts
// with CUBA_HIP_GRAPHS=0 in its environment (or "pcg_graph" = 0 on every handle before its first solve); the automatic rule below only
// keeps a second handle from adding graphs of its own.
extern std::atomic<int> g_liveHandles;

constexpr int CUBA_HIP_BATCH_MAX = 64;       // graphs per cuba_hip_optimize_batch call

}  // namespace cubahip_host
using namespace cubahip_host;

struct cuba_hip_solver;
int cuba_hip_optimize_batch_impl(cuba_hip_solver** hs, int n, int niter, double* chi2, int* nDone);

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
	int spmvUpper = -1;          // option "spmv_upper": three-launch PCG iteration on the upper-triangular storage (ba_pcg.hip): -1 = automatic
	                             // (graphs whose SpMV is bound by bytes: more than 1536 free poses), 0 / 1 = off / on
	DevBuf<Scalar> d_tq; DevBuf<int> d_lowpos;
	// device-side set-up (ba_structure.hip): the edge sort and the whole symbolic structure are built on the GPU; the host
	// pipeline below stays as the independent cross-check ("device_setup" = 0) and for graphs without edges
	bool deviceSetup = true;
	bool devTopology = false;    // the sorted edge arrays / permutation exist on the device only (host copies are stale)
	bool hostTopoValid = false;  // perm / h_lmptr / h_epose / h_spose / h_slm describe the current graph
	static int bitsFor(long long n) { int b = 1; while ((1LL << b) < n + 1) b++; return b; }
	DevBuf<int> d_rawEp, d_rawEl, d_counters, d_tmpI0, d_tmpI1, d_adjRow, d_lowerPtr, d_chunk;
	DevBuf<uint8_t> d_rawDim;
	DevBuf<double> d_rawMeas, d_rawOmega, d_chiCaller;
	// upload for one rank of a landmark partition (cuba_hip_set_graph_partition): ids + packed {meas, omega} of the edges the rank owns
	DevBuf<int> d_ownIds; DevBuf<double> d_ownVals;
	std::vector<int> h_ownIds; std::vector<double> h_ownVals;
	int64_t cntValueBytes = 0;   // bytes of measurements + information that crossed PCIe (all uploads of this handle)
	bool valuesPartial = false;  // the last upload was a rank's (cuba_hip_set_graph_partition): only its landmark range has values on the device
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
	// Internal landmark order (round 5).  Edges are sorted by (landmark, pose), and everything landmark-major -- the records the Schur passes
	// gather, the per-landmark systems -- inherits the locality of the landmark numbering.  The caller's order follows
	// the vertex ids; map points of a SLAM system are numbered in creation order (neighbouring ids see the same keyframes), arbitrary ids
	// are not (the synthetic benchmark graphs draw every track's first frame at random: with ids in creation order G4M's linearise + Schur
	// takes 670 instead of 771 us, the evaluation 40 instead of 57 us, back-substitution 89 instead of 120 us: profiles/r05q_*).  So the free
	// landmarks are renumbered internally by (first observing pose, last observing pose) of the internal pose order; every host-pointer
	// entry point keeps the caller's numbering.  Off for landmark partitions (their ranges are in the caller's numbering) and in the host pipeline.
	bool lmReorder = true;       // option "landmark_reorder"
	bool lmOrderActive = false;
	DevBuf<int> d_lmMap, d_rawElCaller, d_lmFirst, d_lmLast;
	DevBuf<Scalar> d_rowTmp;
	// (a landmark partition names its range in the caller's numbering: only the whole range leaves the internal order alone)
	bool landmarkOrderAllowed() const { return lmReorder && Lf > 1 && E > 0 && (partHi < 0 || (partLo <= 0 && partHi >= Lt)); }
	void switchLandmarkOrder(bool on);                             // between the caller's and the internal order on a live graph (device path)
	void computeLandmarkOrder();                                   // d_rawEp (current pose numbering) + d_rawElCaller -> d_lmMap, d_rawEl
	void resetLandmarkOrder();                                     // identity: d_rawEl = d_rawElCaller
	void landmarkRowsInPlace(Scalar* rows, int nrows, int width, bool toInternal);
	const Scalar* landmarkRowsForCaller(const Scalar* rows, int nrows, int width);      // rows in the caller's order (a temporary when the order is active)
	bool mixedPrecision = false; // fp64 library: records + per-edge arithmetic of the pose / block passes in fp32 (sums, reduced system, PCG in fp64)
	bool profile = false;

	// host copy of the problem (solver order) and of the sort permutation
	int Pt = 0, Pf = 0, Lt = 0, Lf = 0, E = 0;
	bool haveGraph = false, haveStructure = false;
	int partLo = 0, partHi = -1; // landmark range [partLo, partHi) this handle evaluates (-1 = all): multi-GPU partition
	// Reduction in parts (landmark partitions, option "reduction_chunks"): the reduced matrix is cut at block rows into ranges of about
	// equal size, the block list of the Schur pass is grouped by range (order inside a range kept), and the pass runs range by range --
	// a multi-GPU driver starts summing a finished range over the ranks while the next one is computed (cuba_hip_schur_part).
	struct RedPart { BlockPassRange od; size_t blkBegin, blkEnd; };
	int redChunks = 0;           // 0 = automatic (one part per 8 MiB of the reduced matrix, at most 8), n = that many parts at most
	std::vector<RedPart> redParts;   // empty: one part, the block list as the structure builders left it
	void cutReductionParts();
	int schurParts() const { return redParts.empty() ? 1 : (int)redParts.size(); }
	// part 0: landmark pass, pose pass and the blocks of range 0; part c: the blocks of range c.  ranges = {offset, count} of the part's
	// block range in the reduction buffer, then {offset, count} of [bsc | bp] (part 0) or {0, 0}
	void schurPart(int part, size_t ranges[4]);
	bool partsByCaller = false;  // (schurPart(0) in flight: linearize() leaves the further ranges to the caller)
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
	DevBuf<long long> d_lmPairBase;     // exclusive scan of the per-landmark product counts (device structure build)
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
	bool firstInvValid = false, firstInvPending = false;
	bool heuristics = true;     // option "heuristics": the two run-to-run memories (first solve's coarse inverse of the previous run, repeat prediction of the batch lengths)
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
	bool useGraph = graphsByDefault();
	static bool graphsByDefault() { const char* e = std::getenv("CUBA_HIP_GRAPHS"); return e && e[0] == '1'; }      // opt-in (round 6)
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
	void graphWorker();
	// the graph of `chunk` iterations if it exists; otherwise it is ordered (once) and nullptr returned
	hipGraphExec_t pcgGraphIfReady(int chunk, int maxIter, Scalar tol2);
	void dropPcgGraph();
	std::map<int, int> batchRequests;      // how often a batch of this length was asked for since the graphs were dropped

	void enqueuePcgIteration(int k, int maxIter, Scalar tol2, hipStream_t s)
	{
		if (sys.agg > 0 && sys.upper) launch_pcg_upper_iteration(g, st, sys, k, maxIter, tol2, s);
		else if (sys.agg > 0)
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
	int pendingInv = -1;                // >= 0 while a coarse inversion runs on the second stream (evInverse marks its end)
	bool assemblePending = false;       // the other stream may still be reading hsc
	void ensureOverlapObjects();
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
	void startRunHistory() { if (!runIters.empty()) prevRunIters.swap(runIters); runIters.clear(); directSticky = false; }
	int firstSolveIters = 0;     // ... and of the first solve of the previous run

	double lambda = 0;
	int maxIterAlloc = 0;
	long long nmul = 0;
	int64_t cntPcgIters = 0, cntTrials = 0, cntCoarseRefresh = 0, cntPcgLooks = 0, cntPcgEnqueued = 0, cntPcgUnconverged = 0;
	int64_t cntPcgPlain = 0;          // PCG iterations enqueued as plain launches while hipGraphs were switched on (graph not built yet, or another handle active)
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
		if (h_lmRing) (void)hipHostFree(h_lmRing);
		if (h_batchTab) (void)hipHostFree(h_batchTab);
		if (h_gjTab) { (void)hipHostFree(h_gjTab); (void)hipEventDestroy(evGjTab); }
		for (hipEvent_t e : batchEvents) (void)hipEventDestroy(e);
		if (ownStream && stream) (void)hipStreamDestroy(stream);
	}

	void sync() { HIP_TRY(hipStreamSynchronize(stream)); }

	// Completion of the work enqueued so far, learnt from the ticket the last reporting kernel writes into the mapped host
	// block: a spin on host memory sees it ~1 us after the kernel, hipStreamSynchronize only after ~20 us.
	bool failDirty = true;       // the device-side failure flag of the PCG may be non-zero
	int expectedTicket = 0;
	bool hintSameEdges = false, hintSameValues = false;   // cuba_hip_hint_unchanged: promises about the next set_graph call
	void noteReport() { expectedTicket++; }
	void waitReport();

	// host double <-> device Scalar transfers (plain copies in the fp64 build, staged conversion in the fp32 build)
	void downloadAsDouble(const Scalar* dsrc, double* hdst, size_t n);
	void uploadFromDouble(Scalar* ddst, const double* hsrc, size_t n);

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
	bool sortedValuesValid = false;     // d_mu / d_mv / d_mr / d_w hold the gathered values of the last upload (false while a two-step upload is open or was abandoned)
	int* h_tileStage = nullptr; size_t tileStageCap = 0; hipEvent_t evTileInputs = nullptr;     // page-locked staging of the tile-order inputs
	void finishValues();
	void setGraph(int Pt_, int Pf_, int Lt_, int Lf_, const double* q, const double* t, const double* cam, const double* Xw,
		int E_, const int32_t* ep, const int32_t* el, const uint8_t* edim, const double* meas, const double* omega, bool deferValues = false,
		int ownLo = 0, int ownHi = -1);      // ownHi >= 0: the handle is restricted to the landmarks [ownLo, ownHi) and reads the values of their edges only

	// ---------------------------------------------------------------------------------------------
	// Symbolic structure: Hsc pattern from landmark co-visibility (ref: HschurSparseBlockMatrix::
	// constructFromVertices, src/sparse_block_matrix.cpp:55-133 -- here sort/unique instead of a dense
	// P x P map, and every free pose always owns its diagonal block), destination block of every Schur
	// product (ref: findHschureMulBlockIndicesKernel, cuda_block_solver.cu:979-1000), symmetric adjacency
	// for the PCG, wave work list.
	// ---------------------------------------------------------------------------------------------
	void buildStructure();

	// coarse level of the preconditioner: aggregates of consecutive free poses
	struct CoarseCfg { int agg, cl, nc, spmvRows; };
	CoarseCfg coarseConfig() const;

	// everything whose size follows from (E, Pf, Lf, nblk) and the coarse configuration
	int rzStrideCfg = 1, pqStrideCfg = 1;
	void allocSystem(int nblk, const CoarseCfg& c);

	// kernel-argument structures from the device buffers (identical for the host-built and the device-built structure)
	// Order of the blocks in the Schur block pass for graphs beyond 2^19 products: the 16 blocks of a workgroup come from ONE tile of the
	// block matrix where possible -- the records they share then hit the CU's L1 after the first group's miss, and the pass is bound by the L1's
	// outstanding misses (PMC: 4.6 L1->L2 requests per product, 546 cycles each, the L1 stalled on pending misses for 70 % of the launch:
	// profiles/r03zw_pmc_schur_and_pcg_kernels.txt) --, tiles' leftovers re-chunked in tile order (neighbouring tiles share records too),
	// chunks ordered by their longest list, -1 padding.  KITTI-00: linearise + Schur 110.7 -> 102.3 us, S2M 399 -> 358 us.
	static bool rowGroupedBlocks(long long products) { return products > (1LL << 19); }
	// (flat arrays and radix / counting sorts: the grouping sits on the critical path of a NEW topology -- 1.4 ms at KITTI-00 as nested
	// vectors with comparison sorts, ~0.2 ms like this; the output is the same list)
	std::vector<int> rowGroupedOrder(const int* blkRow, const int* blkCol, const int* cnt, int nblk) const;
	int diagProdBlocks = 0;      // diagonal blocks with products (duplicate observations), set by the structure builders
	int heavyBlocks = 0;         // blocks with more than BP_HEAVY products (the first ones of d_odBlocks), set by the structure builders
	void publishStructure(int nblk, int nWaves, int nBig, int nOd, int nCb, int ellM, int ellOver, const CoarseCfg& c);
	bool hostPatternValid = false;     // h_rowptr / h_colind describe the current structure (the device-built one downloads them on demand)
	void fillProdLm();

	// ---- internal pose order ------------------------------------------------------------------------------------------
	int farOffset() const { return std::max(24, Pf / 8); }      // "far from the diagonal", in block columns
	void resetPoseOrder();
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
	void permuteStateRows(std::vector<Scalar>& state, std::vector<Scalar>& camv) const;

	// keys (landmark, pose) of the raw device edge arrays -> sort permutation, sorted edge arrays, landmark pointers
	void runDeviceEdgeSort(bool withValues = true);

	// Strongest-neighbour walk over the pose graph weighted by the number of Schur products per block (= co-visible landmarks):
	// start at the pose of smallest weighted degree, always step to the heaviest unvisited neighbour, when stuck continue from
	// the unvisited pose most strongly tied to the visited ones.  On a keyframe trajectory this IS the trajectory order, loop
	// closures included (consecutive frames share far more landmarks than revisits do); scripts/experiments/precond_experiment6.py.
	// (A bandwidth-minimising order is the wrong tool: RCM interleaves the laps of a revisited stretch, the coarse space then
	// cannot move one lap against the other and the PCG needs 1358 instead of 74 iterations.)
	std::vector<int> chainOrder(const std::vector<int>& rowptr, const std::vector<int>& colind, const std::vector<int>& prodPtr) const;

	// renumber the free poses internally (device path only): state / camera rows, the pose index of every edge, then the edge
	// sort again; the structure has to be rebuilt afterwards
	void applyPoseOrder(const std::vector<int>& newOfOld);

	// after a caller-order structure build: is the pose order bad enough to look for a better one?  true = renumbered,
	// build the structure again
	bool tryReorder(int nblk, int farBlocks);

	// the host pipeline (landmark partitions, atomic Schur kernel) needs the sorted arrays the device path kept to itself
	void ensureHostTopology();

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

	void buildStructureDevice();

	// the block pattern / values as the CALLER numbers the poses (introspection entry points; identity unless reorderActive)
	struct CallerBlock { uint64_t key; int src; bool transposed; };
	std::vector<CallerBlock> callerBlocks();

	void ensureHostPattern();

	void need() { if (!haveGraph) throw StateError{ "set_graph must be called first" }; buildStructure(); finishValues(); g.rk[0] = rk[0]; g.rk[1] = rk[1]; st.mixed = mixedPrecision ? 1 : 0; }

	double readSlots(int which)
	{
		sync();
		double s = 0;
		for (int i = 0; i < NSLOT; i++) s += slot(which * NSLOT + i);
		return s;
	}

	double computeErrors();

	// [hsc | bsc | bp] needs zeroing only where a block may have no writer: a landmark
	// partition leaves blocks without local products; otherwise the pose pass writes every diagonal block, bp and bsc
	// and the block pass every off-diagonal block
	// (force: the assemble-only mode writes the diagonal blocks' upper triangles and bp only -- off-diagonal blocks, bsc and the lower
	// triangles would otherwise keep a previous trial's values, which the stage API exposes through cuba_hip_get_array /
	// cuba_hip_reduction_buffer and a multi-GPU driver sums)
	void zeroReduced(bool force = false) { waitAssembled(); if (force || partHi >= 0 || !reducedZeroed) { d_red.zero(stream); reducedZeroed = true; } }
	bool reducedZeroed = false;

	// withBackup: the state is also copied into its backup (push() of the LM loop) -- inside the landmark pass's launch where possible
	void linearize(int mode, double lam, bool withBackup = false);

	// assemble only: Hpp -> diagonal blocks, bp, raw Hll/bl, landmark part of the max diagonal
	void assemble();

	// max diagonal of the (possibly externally reduced) Hpp and of the local Hll
	void maxDiagonalParts(double* posePart, double* lmPart);

	void scaleParts(double lam, double* posePart, double* lmPart);

	double maxDiagonal();

	void schur(bool withBackup = false);

	// A PCG that BREAKS DOWN (p.Ap <= 0 or a NaN -- not a solve that merely runs out of iterations) while the coarse inverse is stored in
	// fp32 is repeated once with fp64 storage, which the handle then keeps: rounding a symmetrised inverse to fp32 perturbs it by
	// ~6e-8 ||Ac^-1||, which can cost positive definiteness once lambda_max(block) / lambda_min(Ac) approaches 1e7 (weakly constrained
	// graphs at very small damping; round-3 advisor).  Counted in "precond_fp32_fallbacks".
	bool lastSolveBrokeDown = false;
	int lastFailCode = 0;        // device failure code of the last PCG (1: a diagonal block is not positive definite, 2: p.Ap <= 0, 3: NaN)
	bool solveReduced();

	// Exact reduced solve (ba_direct.hip: sparse tile Cholesky, tile products on the matrix cores), the counterpart of the reference's
	// SparseLinearSolver::solve (src/cuda_linear_solver.cpp:386-415: exact, false only on a non-positive pivot).  A solve goes there when
	// the PCG has used its iteration budget without meeting pcg_tol, when it broke down, and -- for the rest of the LM run -- once one
	// solve of the run had to: such systems (nearly singular Hsc: most observations of many poses at zero robust weight) are the ones an
	// iterative solve is the wrong tool for.  "reduced_solver" = 1 sends EVERY solve there (the reference's behaviour).  Ordering and
	// symbolic analysis happen once per structure, at the first exact solve (the reference's SparseLinearSolver::initialize, :278-348);
	// the only limit is the factor's fill: "direct_max_tiles" 32 x 32 tiles (a pure function of the block pattern -- every rank of a
	// partitioned run decides alike); beyond, the old failure report stands (the trial is rejected like a failed factorisation of the
	// reference).
	bool directFallback = true;         // option "direct_fallback"
	bool directAlways = false;          // option "reduced_solver" = 1
	int directAfter = 0;                // option "direct_after": PCG iterations a solve may use before it is handed over; 0 = automatic (Pf / 4 in 128 ... 384)
	int directMaxTiles = 1 << 20;       // option "direct_max_tiles" (2^20 tiles = 16 GiB of factor in fp64)
	int directSlack = -1;               // option "direct_slack": multiple-elimination slack of the ordering, -1 = automatic
	bool directSticky = false;          // a solve of the current run went to the exact solver: the run's remaining solves go there at once
	bool directRefused = false;         // the factor's fill exceeds direct_max_tiles (decided once per structure)
	bool directPlanValid = false;       // directPlan describes the current structure
	double directSeconds = 0, directPlanSeconds = 0;    // host-side wall time of the last exact solve / of the symbolic phase (reporting only)
	SparseCholPlan directPlan;
	SparseChol directDev;
	DevBuf<Scalar> d_scTiles, d_scTilesT, d_scY, d_scRinv;
	DevBuf<int> d_scInts, d_scFail;
	int64_t cntDirect = 0, cntDirectFailed = 0;
	bool directUsable() const { return directFallback && !directRefused && Pf > 0; }
	int pcgBudget(int maxIter) const;
	bool ensureDirectPlan();
	bool solveDirect();

	struct CoarseJob { cuba_hip_solver* h; int first; bool firstInvCopy; bool ownEvent; };
	void launchCoarseJobs(std::vector<CoarseJob>& jobs, hipEvent_t common = nullptr);
	GjJob* h_gjTab = nullptr; DevBuf<unsigned char> d_gjTab; hipEvent_t evGjTab = nullptr;
	struct SolveCtx
	{
		int maxIter = 0, budget = 0, predicted = 32, k0 = 0, looks = 0, eagerIters = 0;
		int itersDone = -1;                 // >= 0: the iterations that count for this solve (a batch ran past the graph's own budget)
		Scalar tol2 = 0;
		bool twoLevel = false, direct = false, graphs = false, converged = false, result = false;
		bool batched = false;               // the iterations run in another handle's launch chain (cuba_hip_optimize_batch): no hipGraphs
		std::vector<CoarseJob>* deferCoarse = nullptr;      // ... and the overlapped coarse inversion is only decided here, enqueued by launchCoarseJobs
		bool deferLaunch = false;           // ... and neither are the set-up launch and the first preconditioner application (when `deferred` comes back true)
		bool deferred = false;
		const Scalar* copySrc = nullptr; Scalar* copyDst = nullptr; size_t copyCount = 0;      // the set-up launch's copy of a fresh coarse inverse
		volatile int* hInts = nullptr;
		Clock::time_point tSolve0;
	};
	bool solveBegin(SolveCtx& sc);
	bool solveBrokeDown(SolveCtx& sc);
	bool solveEnd(SolveCtx& sc);
	bool solveReducedOnce();
	bool retryWithFp64Inverse(bool ok);

	void backSubstitute();

	bool solve();

	void update();

	// Stage-API version of sum x (lambda x + b): recomputed from xp/bp and xl/bl, valid for any lambda.
	double computeScale(double lam);

	void push() { need(); HIP_TRY(hipMemcpyAsync(d_backup.data(), d_state.data(), d_state.size() * sizeof(Scalar), hipMemcpyDeviceToDevice, stream)); }
	void pop() { need(); HIP_TRY(hipMemcpyAsync(d_state.data(), d_backup.data(), d_state.size() * sizeof(Scalar), hipMemcpyDeviceToDevice, stream)); }

	// Levenberg-Marquardt, control flow of CudaBundleAdjustmentImpl::optimize (:793-857).
	int optimize(int niter, double* chi2Out);
	// The same loop with the decision of every trial taken ON THE DEVICE (gain ratio, acceptance, next damping: lm_decide in ba_edge.hip):
	// the host enqueues trial n + 1 behind trial n's tail without having seen its outcome -- the kernels read the damping from device
	// memory, a rejected trial is undone by a conditional restore launch -- and learns the outcomes one trial late from device-mapped
	// records, at the look its next reduced solve needs anyway.  One host look per trial instead of two; only a trial whose outcome may
	// END the run (last iteration, tenth rejection in a row) is waited for.  Same arithmetic as the host loop, bit for bit.
	int optimizeDeviceDecision(int niter, double* chi2Out);
	// (the same run in steps: cuba_hip_optimize_batch interleaves the steps of several handles and batches their PCG iterations)
	struct LmRun
	{
		static constexpr int maxq = 10;
		int niter = 0, enq = 0, seen = 0, done = 0, rejRun = 0;
		bool stop = false;
		double lam = 0, F = 0, tagBase = 0;
		double* chi2Out = nullptr;
		LmDevice lm;
	};
	void lmRunBegin(LmRun& r, int niter, double* chi2Out);
	void lmAbsorb(LmRun& r, int upto);
	bool lmBeforeTrial(LmRun& r);
	bool lmAfterSolveHost(LmRun& r);
	bool lmAfterSolve(LmRun& r, bool ok);
	bool fullyBatchable() const;
	int lmRunEnd(LmRun& r);
	bool batchable() const;
	std::vector<hipEvent_t> batchEvents;        // (the lead handle of a batch owns the join / fork events and the device table)
	BatchEntry* h_batchTab = nullptr; int batchTabEntries = 0;
	DevBuf<unsigned char> d_batchTab;
	DevBuf<double> d_lmState; DevBuf<Scalar> d_lamS;
	double* h_lmRing = nullptr; double* lmRingDev = nullptr; uint64_t lmRunNonce = 0; int64_t cntLateRecords = 0;
	int64_t cntHostLooks = 0;           // waits of the host for a device report (PCG looks + LM decisions it had to see)

	// chi2 of the trial estimate and sum x (lambda x + b) of the step that led to it, read back with ONE synchronisation
	void enqueueEvaluate(double lam, bool withScale);
	void readEvaluate(bool withScale, double* Fhat, double* scale);
	void evaluateTrial(double lam, bool withScale, double* Fhat, double* scale);

	// Fused version used by optimize(): the landmark part was accumulated by back_substitute (same lambda),
	// only the 6*Pf pose part is added here.

	// Average device time per launch of the five hot kernels, measured with HIP events on this solver's
	// stream (bench.py's roofline leg).  Leaves the increments / reduced system in an undefined state.
	void timeKernels(int reps, double* msOut);

	void chiSquares(double* out, bool wait = true);
};
