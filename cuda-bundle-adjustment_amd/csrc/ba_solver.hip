// ba_solver.hip -- the C ABI (include/cuba_hip.h) over the solver handle of ba_solver.hpp.
//
// There is deliberately no CPU fallback: every entry point fails with CUBA_HIP_ERR_NO_DEVICE / CUBA_HIP_ERR_RUNTIME when no gfx950 device
// is usable.
#include "ba_solver.hpp"

using namespace cubahip;


// -----------------------------------------------------------------------------------------------------
// C ABI
// -----------------------------------------------------------------------------------------------------
namespace cubahip_host
{
std::atomic<int> g_liveHandles{ 0 };
}

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
	g_liveHandles.fetch_add(1, std::memory_order_relaxed);
	*out = s;
	return CUBA_HIP_OK;
}

int cuba_hip_destroy(cuba_hip_solver* s)
{
	if (!s) return CUBA_HIP_ERR_INVALID_ARGUMENT;
	(void)hipSetDevice(s->device);
	(void)hipStreamSynchronize(s->stream);
	delete s;
	g_liveHandles.fetch_sub(1, std::memory_order_relaxed);
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
		else if (k == "heuristics") { s->heuristics = value != 0; s->firstInvValid = false; s->firstInvPending = false; }
		else if (k == "precond_fp32") { s->precondFp32 = value != 0; s->haveStructure = false; s->coarseValid = false; s->dropPcgGraph(); }
		else if (k == "pose_reorder") { s->poseReorder = value != 0; s->haveStructure = false; }
		else if (k == "device_setup") { s->deviceSetup = value != 0; s->haveStructure = false; }
		else if (k == "mixed_precision") s->mixedPrecision = value != 0 && sizeof(Scalar) == 8;
		else if (k == "pcg_accept_unconverged") s->acceptUnconverged = value != 0;
		else if (k == "direct_fallback") { s->directFallback = value != 0; s->directSticky = false; }
		else if (k == "direct_after") s->directAfter = (int)value;
		else if (k == "direct_max_tiles") { s->directMaxTiles = (int)value; s->directRefused = false; s->directPlanValid = false; }
		else if (k == "direct_slack") { s->directSlack = (int)value; s->directRefused = false; s->directPlanValid = false; }
		else if (k == "reduced_solver")
		{
			if (value != 0 && value != 1) throw ArgError{ "reduced_solver: 0 = block PCG with the exact solver as fallback, 1 = exact sparse Cholesky for every solve" };
			s->directAlways = value != 0;
		}
		else if (k == "pcg_aggregate") { s->pcgAggregate = (int)value; s->haveStructure = false; }
		else if (k == "coarse_linear") { s->coarseLinear = value != 0; s->haveStructure = false; }
		else if (k == "landmark_reorder") s->lmReorder = value != 0;         // (takes effect with the next cuba_hip_set_graph)
		else if (k == "reduction_chunks")
		{
			if (value < 0 || value > 64) throw ArgError{ "reduction_chunks must lie in 0 .. 64" };
			s->redChunks = (int)value; s->haveStructure = false;
		}
		else if (k == "spmv_upper") { s->spmvUpper = (int)value; s->haveStructure = false; s->dropPcgGraph(); }
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

int cuba_hip_set_graph_partition(cuba_hip_solver* s, int Pt, int Pf, int Lt, int Lf,
	const double* q, const double* t, const double* cam, const double* Xw,
	int E, const int32_t* edge_pose, const int32_t* edge_landmark, const uint8_t* edge_dim,
	const double* meas, const double* omega, int landmark_begin, int landmark_end)
{
	return guarded(s, [&] {
		if (landmark_end < 0) throw ArgError{ "bad landmark range" };
		s->setGraph(Pt, Pf, Lt, Lf, q, t, cam, Xw, E, edge_pose, edge_landmark, edge_dim, meas, omega, false, landmark_begin, landmark_end);
	});
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

int cuba_hip_schur_parts(cuba_hip_solver* s, int* n_parts)
{
	return guarded(s, [&] { s->need(); if (n_parts) *n_parts = s->schurParts(); });
}

int cuba_hip_schur_part(cuba_hip_solver* s, int part, size_t ranges[4])
{
	return guarded(s, [&] { size_t r[4]; s->schurPart(part, r); if (ranges) std::copy(r, r + 4, ranges); });
}

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

int cuba_hip_optimize_batch(cuba_hip_solver** handles, int n, int niterations, double* chi2_per_iter, int* n_done, int* batched_solves)
{
	if (!handles || n <= 0 || n > CUBA_HIP_BATCH_MAX || !n_done) return CUBA_HIP_ERR_INVALID_ARGUMENT;
	for (int i = 0; i < n; i++)
	{
		if (!handles[i]) return CUBA_HIP_ERR_INVALID_ARGUMENT;
		for (int j = 0; j < i; j++) if (handles[j] == handles[i]) return CUBA_HIP_ERR_INVALID_ARGUMENT;
	}
	// (an error is recorded on the first handle: cuba_hip_last_error(handles[0]))
	return guarded(handles[0], [&] {
		if (niterations < 0) throw ArgError{ "negative iteration count" };
		const int b = cuba_hip_optimize_batch_impl(handles, n, niterations, chi2_per_iter, n_done);
		if (batched_solves) *batched_solves = b;
	});
}

int cuba_hip_get_solution(cuba_hip_solver* s, double* q, double* t, double* Xw)
{
	return guarded(s, [&] {
		if (!s->haveGraph) throw StateError{ "set_graph must be called first" };
		const Scalar* base = s->d_state.data();
		if (q) { s->downloadAsDouble(base, q, (size_t)4 * s->Pt); s->permutePoseArray(q, 4, false); }
		if (t) { s->downloadAsDouble(base + 4 * (size_t)s->Pt, t, (size_t)3 * s->Pt); s->permutePoseArray(t, 3, false); }
		if (Xw) s->downloadAsDouble(s->landmarkRowsForCaller(base + 7 * (size_t)s->Pt, s->Lt, 3), Xw, (size_t)3 * s->Lt);
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
		if (Xw)
		{
			s->uploadFromDouble(base + 7 * (size_t)s->Pt, Xw, (size_t)3 * s->Lt);
			if (s->lmOrderActive) s->landmarkRowsInPlace(base + 7 * (size_t)s->Pt, s->Lt, 3, true);
		}
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
		else if (k == "pcg_iterations_plain_launches") *value = s->cntPcgPlain;
		else if (k == "host_looks") *value = s->cntHostLooks;
		else if (k == "exact_solve_fallbacks") *value = s->cntDirect;
		else if (k == "exact_solve_failures") *value = s->cntDirectFailed;
		else if (k == "graph_uploads") *value = s->cntUploads;
		else if (k == "value_bytes_uploaded") *value = s->cntValueBytes;
		else if (k == "late_decision_records") *value = s->cntLateRecords;
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
			// (landmark-indexed arrays: back to the caller's landmark numbering)
			if (which == CUBA_HIP_ARRAY_XL) src = s->landmarkRowsForCaller(src, s->Lf, 3);
			else if (which == CUBA_HIP_ARRAY_LM_SYS) src = s->landmarkRowsForCaller(src, s->Lf, 9);
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
		// (after a rank's upload the device holds the values of that rank's landmarks only: any other range would silently evaluate zeros)
		if (s->valuesPartial && (landmark_end < 0 || landmark_begin < s->partLo || landmark_end > s->partHi))
			throw StateError{ "cuba_hip_set_graph_partition uploaded this range's values only: a wider range needs a new upload" };
		if (landmark_begin == 0 && landmark_end == -1)          // remove the restriction: the handle evaluates the whole graph again
		{
			if (s->partHi >= 0) s->haveStructure = false;
			s->partLo = 0; s->partHi = -1;
			s->switchLandmarkOrder(s->landmarkOrderAllowed());      // (back to the internal landmark order a partition had ended)
			return;
		}
		if (landmark_begin < 0 || landmark_end > s->Lt || landmark_begin > landmark_end) throw ArgError{ "bad landmark range" };
		if (s->partHi >= 0 && landmark_begin == s->partLo && landmark_end == s->partHi) return;      // (cuba_hip_set_graph_partition set it already)
		s->partLo = landmark_begin; s->partHi = landmark_end;
		s->haveStructure = false;
		// the range is in the caller's landmark numbering: unless it is the whole range, the internal landmark order ends here (rows of the
		// estimates back to the caller's order, edges sorted again)
		s->switchLandmarkOrder(s->landmarkOrderAllowed());
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
		launch_coarse_finish(res, res, n, nullptr);
		HIP_TRY(hipMemcpy(h.data(), res, sizeof(Scalar) * nn, hipMemcpyDeviceToHost));
		for (size_t i = 0; i < nn; i++) Ainv[i] = (double)h[i];
		return CUBA_HIP_OK;
	}
	catch (const HipError&) { return CUBA_HIP_ERR_RUNTIME; }
}

int cuba_hip_debug_dense_solve(int device, int n, const double* A, const double* b, double* x, int* not_positive_definite)
{
	return cuba_hip_debug_sparse_solve(device, n, A, b, x, not_positive_definite, -1, nullptr);
}

int cuba_hip_debug_sparse_solve(int device, int n, const double* A, const double* b, double* x, int* not_positive_definite, int slack, int32_t stats[4])
{
	if (n <= 0 || n % 6 != 0 || !A || !b || !x) return CUBA_HIP_ERR_INVALID_ARGUMENT;
	if (hipSetDevice(device) != hipSuccess) return CUBA_HIP_ERR_NO_DEVICE;
	try
	{
		// the matrix as the reduced system would hold it: the 6 x 6 blocks on or above the diagonal that are not identically zero (the
		// diagonal ones always), one block per (row, column)
		const int P = n / 6;
		std::vector<Scalar> blocks; std::vector<int> blkrow, colind, rowptr(1, 0);
		for (int bi = 0; bi < P; bi++)
		{
			for (int bj = bi; bj < P; bj++)
			{
				bool any = bi == bj;
				for (int c = 0; c < 6 && !any; c++) for (int r = 0; r < 6; r++) any = any || A[(size_t)(6 * bj + c) * n + 6 * bi + r] != 0.0;
				if (!any) continue;
				blkrow.push_back(bi); colind.push_back(bj);
				for (int c = 0; c < 6; c++) for (int r = 0; r < 6; r++) blocks.push_back((Scalar)A[(size_t)(6 * bj + c) * n + 6 * bi + r]);
			}
			rowptr.push_back((int)colind.size());
		}
		SparseCholPlan plan;
		if (!sparse_chol_plan(P, rowptr.data(), colind.data(), slack, (size_t)1 << 22, plan)) return CUBA_HIP_ERR_RUNTIME;
		if (stats) { stats[0] = plan.T; stats[1] = plan.nTiles; stats[2] = plan.nLevels; stats[3] = plan.slack; }
		std::vector<Scalar> hb(b, b + n);
		DevBuf<Scalar> dBlocks, dB, dTiles, dTilesT, dY, dRinv, dX; DevBuf<int> dRow, dCol, dFail;
		dBlocks.upload(blocks, nullptr); dB.upload(hb, nullptr); dRow.upload(blkrow, nullptr); dCol.upload(colind, nullptr);
		DevBuf<int> dColPtr, dRowIdx, dColOf, dGPtr, dGather, dLvlTiles, dLvlCols, dBlkTile, dPos;
		dColPtr.upload(plan.colPtr, nullptr); dRowIdx.upload(plan.rowIdx, nullptr); dColOf.upload(plan.colOfTile, nullptr); dGPtr.upload(plan.gPtr, nullptr);
		dGather.upload(plan.gather, nullptr); dLvlTiles.upload(plan.wgRec, nullptr); dLvlCols.upload(plan.lvlCols, nullptr);
		dBlkTile.upload(plan.blkTile, nullptr); dPos.upload(plan.posOfSeg, nullptr);
		dTiles.resize((size_t)SC_TT * ((size_t)plan.nTiles + 1)); dTilesT.resize((size_t)SC_TT * plan.nTiles);
		dY.resize((size_t)SC_T * plan.T); dRinv.resize((size_t)SC_T * plan.T); dFail.resize(1); dX.resize(n);
		SparseChol d;
		d.tiles = dTiles.data(); d.tilesT = dTilesT.data(); d.y = dY.data(); d.rinv = dRinv.data(); d.fail = dFail.data();
		d.colPtr = dColPtr.data(); d.rowIdx = dRowIdx.data(); d.colOfTile = dColOf.data(); d.gPtr = dGPtr.data(); d.gather = dGather.data();
		d.wgRec = dLvlTiles.data(); d.lvlCols = dLvlCols.data(); d.blkTile = dBlkTile.data(); d.posOfSeg = dPos.data();
		d.T = plan.T; d.Pf = P; d.nTiles = plan.nTiles;
		DeviceStructure st; DeviceSystem sys;
		st.nblk = (int)blkrow.size(); st.hsc_blkrow = dRow.data(); st.hsc_colind = dCol.data();
		sys.hsc = dBlocks.data(); sys.bsc = dB.data();
		launch_sparse_chol_fill(st, sys, d, nullptr);
		launch_sparse_chol_solve(d, plan, dX.data(), nullptr);
		std::vector<Scalar> hx(n); int flag = 0;
		HIP_TRY(hipMemcpy(hx.data(), dX.data(), sizeof(Scalar) * n, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(&flag, dFail.data(), sizeof(int), hipMemcpyDeviceToHost));
		for (int i = 0; i < n; i++) x[i] = (double)hx[i];
		if (not_positive_definite) *not_positive_definite = flag;
		return CUBA_HIP_OK;
	}
	catch (const HipError&) { return CUBA_HIP_ERR_RUNTIME; }
}

// The symbolic phase alone (host only: needs no device).  which: 0 header {T, nTiles, nLevels, slack, gather entries, nblk}, 1 posOfSeg,
// 2 colPtr, 3 rowIdx, 4 gPtr, 5 gather (4 ints per entry), 6 lvlPtr, 7 lvlTiles, 8 lvlColPtr, 9 lvlCols, 10 blkTile
int cuba_hip_debug_sparse_plan(int n_poses, const int32_t* row_ptr, const int32_t* col_ind, int slack, int which, int32_t* out, size_t capacity, size_t* count)
{
	if (n_poses <= 0 || !row_ptr || !col_ind || !count) return CUBA_HIP_ERR_INVALID_ARGUMENT;
	static thread_local SparseCholPlan plan;
	static thread_local std::vector<int> key;
	std::vector<int> k(row_ptr, row_ptr + n_poses + 1);
	k.insert(k.end(), col_ind, col_ind + row_ptr[n_poses]);
	k.push_back(slack);
	if (k != key)
	{
		if (!sparse_chol_plan(n_poses, row_ptr, col_ind, slack, (size_t)1 << 22, plan)) return CUBA_HIP_ERR_RUNTIME;
		key.swap(k);
	}
	const std::vector<int> header{ plan.T, plan.nTiles, plan.nLevels, plan.slack, (int)(plan.gather.size() / 4), (int)plan.blkTile.size() };
	const std::vector<int>* src[] = { &header, &plan.posOfSeg, &plan.colPtr, &plan.rowIdx, &plan.gPtr, &plan.gather, &plan.lvlPtr, &plan.lvlTiles,
		&plan.lvlColPtr, &plan.lvlCols, &plan.blkTile };
	if (which < 0 || which > 10) return CUBA_HIP_ERR_INVALID_ARGUMENT;
	*count = src[which]->size();
	if (out) std::memcpy(out, src[which]->data(), sizeof(int) * std::min(capacity, src[which]->size()));
	return CUBA_HIP_OK;
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
