// ba_dist.hip -- native multi-GPU driver behind include/cuba_hip_dist.h: landmark-partitioned Levenberg-Marquardt, one
// solver handle (= one GPU) per rank, RCCL collectives enqueued on the solver's own stream.  Uses only the public C ABI of
// libcuba_hip.so, so the single-GPU library carries no RCCL dependency.  The control flow is that of
// CudaBundleAdjustmentImpl::optimize (/root/reference/src/cuda_bundle_adjustment.cpp:793-857).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "../../include/cuba_hip_dist.h"

namespace
{
struct Fail { int status; std::string msg; };

#define SOLVER_TRY(expr)                                                                            \
	do {                                                                                            \
		const int rc__ = (expr);                                                                    \
		if (rc__ != CUBA_HIP_OK) throw Fail{ rc__, std::string(#expr ": ") + cuba_hip_last_error(d->s) }; \
	} while (0)
#define HIP_TRY(expr)                                                                               \
	do {                                                                                            \
		const hipError_t e__ = (expr);                                                              \
		if (e__ != hipSuccess) throw Fail{ CUBA_HIP_ERR_RUNTIME, std::string(#expr ": ") + hipGetErrorString(e__) }; \
	} while (0)
#define NCCL_TRY(expr)                                                                              \
	do {                                                                                            \
		const ncclResult_t r__ = (expr);                                                            \
		if (r__ != ncclSuccess) throw Fail{ CUBA_HIP_ERR_RUNTIME, std::string(#expr ": ") + ncclGetErrorString(r__) }; \
	} while (0)
}  // namespace

struct cuba_hip_dist
{
	cuba_hip_solver* s = nullptr;
	int rank = 0, world = 1;
	int lmBegin = 0, lmEnd = 0, lmTotal = 0;
	ncclComm_t comm = nullptr;
	bool ownComm = false;
	cuba_hip_comm_ops ops = { nullptr, nullptr, nullptr };
	bool custom = false;
	int scalarSize = 8;
	hipStream_t stream = nullptr;
	void* red = nullptr; size_t redCount = 0;      // [Hsc | bsc | bp] of the solver
	double* hostScalars = nullptr;                  // pinned staging of the evaluation read-back
	long long nLarge = 0, nSmall = 0, largeElems = 0, nTrials = 0;
	std::string lastError;
	bool partitionSet = false;
	// reduction in parts (cuba_hip_schur_part): the sum of part c runs on `commStream` while the solver's stream computes part c + 1
	int nParts = 1;
	bool exchange = false;           // collectives are issued: more than one rank (or a 1-rank RCCL communicator under CUBA_HIP_DIST_SINGLE_RANK_COLLECTIVES,
	                                 // which lets a one-GPU box run the stream / event choreography of the parts over the real library)
	hipStream_t commStream = nullptr;
	hipEvent_t evPart = nullptr, evSummed = nullptr;
	long long nOverlapped = 0;

	// (also the failure path of create(): whatever bind() already acquired is released)
	~cuba_hip_dist()
	{
		if (ownComm && comm) (void)ncclCommDestroy(comm);
		if (hostScalars) (void)hipHostFree(hostScalars);
		if (evPart) (void)hipEventDestroy(evPart);
		if (evSummed) (void)hipEventDestroy(evSummed);
		if (commStream) (void)hipStreamDestroy(commStream);
	}

	void allreduce(void* buf, size_t count, bool max) { allreduceOn(stream, buf, count, max); }

	void allreduceOn(hipStream_t on, void* buf, size_t count, bool max)
	{
		if (!exchange || count == 0) return;
		if (custom)
		{
			const int rc = (max ? ops.allreduce_max : ops.allreduce_sum)(ops.ctx, buf, count, scalarSize, (void*)on);
			if (rc != 0) throw Fail{ CUBA_HIP_ERR_RUNTIME, "custom all-reduce failed" };
			return;
		}
		NCCL_TRY(ncclAllReduce(buf, buf, count, scalarSize == 8 ? ncclDouble : ncclFloat, max ? ncclMax : ncclSum, comm, on));
	}

	// cuba_hip_schur + the sum of the reduction buffer over the ranks.  One part: the pass, then one all-reduce in-stream.  Several parts:
	// part c's ranges are summed on the second stream behind an event, while the solver's stream goes on with part c + 1; the solver's
	// stream then waits for the last sum.  (Collectives of one communicator are issued in the same order on every rank: the cuts are
	// functions of the global block pattern.)
	void schurAndSum()
	{
		if (nParts <= 1 || !exchange)
		{
			if (cuba_hip_schur(s) != CUBA_HIP_OK) throw Fail{ CUBA_HIP_ERR_RUNTIME, std::string("cuba_hip_schur: ") + cuba_hip_last_error(s) };
			allreduce(red, redCount, false); nLarge++; largeElems += (long long)redCount;
			return;
		}
		char* base = (char*)red;
		for (int c = 0; c < nParts; c++)
		{
			size_t r[4] = { 0, 0, 0, 0 };
			if (cuba_hip_schur_part(s, c, r) != CUBA_HIP_OK) throw Fail{ CUBA_HIP_ERR_RUNTIME, std::string("cuba_hip_schur_part: ") + cuba_hip_last_error(s) };
			HIP_TRY(hipEventRecord(evPart, stream));
			HIP_TRY(hipStreamWaitEvent(commStream, evPart, 0));
			for (int k = 0; k < 4; k += 2)
			{
				if (!r[k + 1]) continue;
				allreduceOn(commStream, base + r[k] * (size_t)scalarSize, r[k + 1], false);
				nLarge++; largeElems += (long long)r[k + 1];
				if (c + 1 < nParts) nOverlapped++;
			}
		}
		HIP_TRY(hipEventRecord(evSummed, commStream));
		HIP_TRY(hipStreamWaitEvent(stream, evSummed, 0));
	}

	// read `n` device scalars back (one stream synchronisation)
	void readScalars(const void* dev, int n, double* out)
	{
		HIP_TRY(hipMemcpyAsync(hostScalars, dev, (size_t)n * scalarSize, hipMemcpyDeviceToHost, stream));
		HIP_TRY(hipStreamSynchronize(stream));
		for (int i = 0; i < n; i++) out[i] = scalarSize == 8 ? hostScalars[i] : (double)((const float*)hostScalars)[i];
	}
};

namespace
{
void bind(cuba_hip_dist* d, cuba_hip_solver* s, int rank, int world, int lb, int le)
{
	if (!s || rank < 0 || world < 1 || rank >= world) throw Fail{ CUBA_HIP_ERR_INVALID_ARGUMENT, "bad rank / world" };
	d->s = s; d->rank = rank; d->world = world; d->lmBegin = lb; d->lmEnd = le;
	d->scalarSize = cuba_hip_scalar_size();
	void* st = nullptr;
	SOLVER_TRY(cuba_hip_get_stream(s, &st));
	d->stream = (hipStream_t)st;
	int sizes[5] = { 0, 0, 0, 0, 0 };
	SOLVER_TRY(cuba_hip_get_sizes(s, sizes));
	d->lmTotal = sizes[2];
	if (lb < 0 || le > d->lmTotal || lb > le) throw Fail{ CUBA_HIP_ERR_INVALID_ARGUMENT, "bad landmark range" };
	SOLVER_TRY(cuba_hip_set_partition(s, lb, le));
	d->partitionSet = true;
	SOLVER_TRY(cuba_hip_build_structure(s));
	SOLVER_TRY(cuba_hip_reduction_buffer(s, &d->red, &d->redCount));
	SOLVER_TRY(cuba_hip_schur_parts(s, &d->nParts));
	d->exchange = world > 1 || std::getenv("CUBA_HIP_DIST_SINGLE_RANK_COLLECTIVES") != nullptr;
	if (d->nParts > 1 && d->exchange)
	{
		HIP_TRY(hipStreamCreateWithFlags(&d->commStream, hipStreamNonBlocking));
		HIP_TRY(hipEventCreateWithFlags(&d->evPart, hipEventDisableTiming));
		HIP_TRY(hipEventCreateWithFlags(&d->evSummed, hipEventDisableTiming));
	}
	HIP_TRY(hipHostMalloc((void**)&d->hostScalars, 64, hipHostMallocDefault));
	// fault injection for tests/test_dist.py: a failure AFTER the handle was bound (what a refused ncclCommInitRank looks like)
	if (std::getenv("CUBA_HIP_DIST_TEST_FAIL_AFTER_BIND")) throw Fail{ CUBA_HIP_ERR_RUNTIME, "injected failure after bind" };
}

template <class F>
int guarded(cuba_hip_dist* d, F&& f)
{
	if (!d) return CUBA_HIP_ERR_INVALID_ARGUMENT;
	try { f(); return CUBA_HIP_OK; }
	catch (const Fail& e) { d->lastError = e.msg; return e.status; }
	catch (const std::exception& e) { d->lastError = e.what(); return CUBA_HIP_ERR_RUNTIME; }
}

template <class F>
int create(cuba_hip_dist** out, F&& init)
{
	if (!out) return CUBA_HIP_ERR_INVALID_ARGUMENT;
	*out = nullptr;
	cuba_hip_dist* d = new (std::nothrow) cuba_hip_dist;
	if (!d) return CUBA_HIP_ERR_RUNTIME;
	const int rc = guarded(d, [&] { init(d); });
	if (rc != CUBA_HIP_OK)
	{
		std::fprintf(stderr, "[cuba_hip_dist] create failed: %s\n", d->lastError.c_str());
		// the caller's solver handle gets its full landmark range back: left restricted it would silently optimise a subset
		// of the landmarks when used single-GPU afterwards
		if (d->partitionSet && d->s) (void)cuba_hip_set_partition(d->s, 0, -1);
		delete d;
		return rc;
	}
	*out = d;
	return CUBA_HIP_OK;
}
}  // namespace

extern "C" {

int cuba_hip_dist_unique_id(void* id128)
{
	static_assert(sizeof(ncclUniqueId) == CUBA_HIP_DIST_UNIQUE_ID_BYTES, "unique id size");
	if (!id128) return CUBA_HIP_ERR_INVALID_ARGUMENT;
	ncclUniqueId id;
	if (ncclGetUniqueId(&id) != ncclSuccess) return CUBA_HIP_ERR_RUNTIME;
	std::memcpy(id128, &id, sizeof id);
	return CUBA_HIP_OK;
}

int cuba_hip_dist_create_rccl(cuba_hip_solver* s, const void* id128, int rank, int world, int landmark_begin, int landmark_end, cuba_hip_dist** out)
{
	return create(out, [&](cuba_hip_dist* d) {
		if (!id128) throw Fail{ CUBA_HIP_ERR_INVALID_ARGUMENT, "null unique id" };
		bind(d, s, rank, world, landmark_begin, landmark_end);
		ncclUniqueId id;
		std::memcpy(&id, id128, sizeof id);
		NCCL_TRY(ncclCommInitRank(&d->comm, world, id, rank));
		d->ownComm = true;
	});
}

int cuba_hip_dist_attach_rccl(cuba_hip_solver* s, void* nccl_comm, int rank, int world, int landmark_begin, int landmark_end, cuba_hip_dist** out)
{
	return create(out, [&](cuba_hip_dist* d) {
		if (!nccl_comm) throw Fail{ CUBA_HIP_ERR_INVALID_ARGUMENT, "null communicator" };
		bind(d, s, rank, world, landmark_begin, landmark_end);
		d->comm = (ncclComm_t)nccl_comm;
	});
}

int cuba_hip_dist_create_custom(cuba_hip_solver* s, const cuba_hip_comm_ops* ops, int rank, int world, int landmark_begin, int landmark_end, cuba_hip_dist** out)
{
	return create(out, [&](cuba_hip_dist* d) {
		if (!ops || !ops->allreduce_sum || !ops->allreduce_max) throw Fail{ CUBA_HIP_ERR_INVALID_ARGUMENT, "incomplete collective table" };
		bind(d, s, rank, world, landmark_begin, landmark_end);
		d->ops = *ops; d->custom = true;
	});
}

int cuba_hip_dist_optimize(cuba_hip_dist* d, int niterations, double* chi2_per_iter, int* n_done)
{
	if (n_done) *n_done = 0;
	return guarded(d, [&] {
		if (niterations < 0) throw Fail{ CUBA_HIP_ERR_INVALID_ARGUMENT, "negative iteration count" };
		const int maxq = 10;
		const double tau = 1e-5;
		double nu = 2, lam = 0, F = 0;
		bool haveF = false;
		void* ev = nullptr;
		double h[3];
		SOLVER_TRY(cuba_hip_begin_run(d->s));
		for (int it = 0; it < niterations; it++)
		{
			if (!haveF)
			{
				SOLVER_TRY(cuba_hip_evaluate_device(d->s, 0.0, 0, &ev));
				d->allreduce(ev, 1, false); d->nSmall++;
				d->readScalars(ev, 1, h);
				F = h[0];
			}
			if (it == 0)
			{
				// lambda_0 = tau * max diag: Hpp needs the sum over the ranks, Hll is local
				SOLVER_TRY(cuba_hip_assemble(d->s));
				d->allreduce(d->red, d->redCount, false); d->nLarge++; d->largeElems += (long long)d->redCount;
				double posePart = 0, lmPart = 0;
				SOLVER_TRY(cuba_hip_max_diagonal_parts(d->s, &posePart, &lmPart));
				if (d->world > 1)
				{
					// the landmark part is a max over the ranks: one scalar through the evaluation scratch
					if (d->scalarSize == 8) d->hostScalars[0] = lmPart; else ((float*)d->hostScalars)[0] = (float)lmPart;
					HIP_TRY(hipMemcpyAsync(ev, d->hostScalars, d->scalarSize, hipMemcpyHostToDevice, d->stream));
					d->allreduce(ev, 1, true); d->nSmall++;
					d->readScalars(ev, 1, h);
					lmPart = h[0];
				}
				lam = tau * std::max(posePart, lmPart);
			}
			int qn = 0;
			double rho = -1;
			for (; qn < maxq && rho < 0; qn++)
			{
				d->nTrials++;
				SOLVER_TRY(cuba_hip_push(d->s));
				SOLVER_TRY(cuba_hip_set_lambda(d->s, lam));
				d->schurAndSum();                                     // the one large exchange of the trial (in parts: under the pass itself)
				int ok = 0;
				SOLVER_TRY(cuba_hip_solve_reduced(d->s, &ok));       // replicated; bit-identical on every rank
				if (ok)
				{
					SOLVER_TRY(cuba_hip_back_substitute(d->s));
					SOLVER_TRY(cuba_hip_update(d->s));
				}
				SOLVER_TRY(cuba_hip_evaluate_device(d->s, lam, ok, &ev));
				d->allreduce(ev, 2, false); d->nSmall++;              // {chi2, landmark scale part}; the pose part is replicated
				d->readScalars(ev, 3, h);
				const double Fhat = h[0];
				const double scale = (ok ? h[1] + h[2] : 0.0) + 1e-3;
				rho = ok ? (F - Fhat) / scale : -1;
				if (rho > 0)
				{
					const double t3 = 2 * rho - 1;
					const double a = 1 - t3 * t3 * t3;          // (the expressions of the device-resident decision, ba_edge.hip: lm_decide)
					lam *= std::max(1. / 3, std::min(a, 2. / 3));
					nu = 2;
					F = Fhat;
					haveF = true;
					break;
				}
				lam *= nu;
				nu *= 2;
				SOLVER_TRY(cuba_hip_pop(d->s));
				haveF = true;            // F still describes the restored estimate
			}
			if (chi2_per_iter) chi2_per_iter[it] = F;
			if (n_done) *n_done = it + 1;
			if (qn == maxq || rho <= 0 || !std::isfinite(lam)) break;
		}
	});
}

int cuba_hip_dist_complete_solution(cuba_hip_dist* d)
{
	return guarded(d, [&] {
		if (d->world == 1) return;
		// state = [q(4 Pt) | t(3 Pt) | Xw(3 Lt)]: every rank zeroes the landmarks it does not own, the sum is the full set
		void* state = nullptr; size_t n = 0;
		SOLVER_TRY(cuba_hip_device_pointer(d->s, CUBA_HIP_ARRAY_STATE, &state, &n));
		char* X = (char*)state + (n - 3 * (size_t)d->lmTotal) * d->scalarSize;
		const size_t sz = (size_t)d->scalarSize;
		if (d->lmBegin > 0) HIP_TRY(hipMemsetAsync(X, 0, 3 * (size_t)d->lmBegin * sz, d->stream));
		if (d->lmEnd < d->lmTotal) HIP_TRY(hipMemsetAsync(X + 3 * (size_t)d->lmEnd * sz, 0, 3 * (size_t)(d->lmTotal - d->lmEnd) * sz, d->stream));
		d->allreduce(X, 3 * (size_t)d->lmTotal, false); d->nLarge++; d->largeElems += 3LL * d->lmTotal;
		HIP_TRY(hipStreamSynchronize(d->stream));
	});
}

int cuba_hip_dist_get_counters(cuba_hip_dist* d, long long c[4])
{
	return guarded(d, [&] { c[0] = d->nLarge; c[1] = d->nSmall; c[2] = d->largeElems; c[3] = d->nTrials; });
}

int cuba_hip_dist_reduction_parts(cuba_hip_dist* d, int* n_parts, long long* overlapped)
{
	return guarded(d, [&] { if (n_parts) *n_parts = d->exchange ? d->nParts : 1; if (overlapped) *overlapped = d->nOverlapped; });
}

const char* cuba_hip_dist_last_error(const cuba_hip_dist* d) { return d ? d->lastError.c_str() : "null driver handle"; }

int cuba_hip_dist_destroy(cuba_hip_dist* d)
{
	if (!d) return CUBA_HIP_ERR_INVALID_ARGUMENT;
	if (d->stream) (void)hipStreamSynchronize(d->stream);
	if (d->commStream) (void)hipStreamSynchronize(d->commStream);
	delete d;            // (the destructor releases the communicator and the pinned block; the solver handle is the caller's)
	return CUBA_HIP_OK;
}

}  // extern "C"
