// ba_lm.hip -- the per-iteration stages, the reduced solve (batches of PCG iterations as hipGraphs, coarse inverse on a second stream) and
// the Levenberg-Marquardt loop: counterpart of CudaBlockSolver::computeErrors ... update and of CudaBundleAdjustmentImpl::optimize
// (/root/reference/src/cuda_bundle_adjustment.cpp:368-510, 793-857).
#include "ba_solver.hpp"

using namespace cubahip;

void cuba_hip_solver::graphWorker()
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

hipGraphExec_t cuba_hip_solver::pcgGraphIfReady(int chunk, int maxIter, Scalar tol2)
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

void cuba_hip_solver::dropPcgGraph()
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

void cuba_hip_solver::ensureOverlapObjects()
{
	if (gjStream) return;
	// (low priority: the sweep fills the gaps of the latency-bound PCG kernels it runs under; confining it to every n-th CU instead
	// was measured at <= 1 %, profiles/r03*)
	int prioLow = 0, prioHigh = 0;
	HIP_TRY(hipDeviceGetStreamPriorityRange(&prioLow, &prioHigh));
	HIP_TRY(hipStreamCreateWithPriority(&gjStream, hipStreamNonBlocking, prioLow));
	HIP_TRY(hipEventCreateWithFlags(&evSetup, hipEventDisableTiming));
	HIP_TRY(hipEventCreateWithFlags(&evAssembled, hipEventDisableTiming));
	HIP_TRY(hipEventCreateWithFlags(&evInverse, hipEventDisableTiming));
	HIP_TRY(hipEventCreateWithFlags(&evFirstInv, hipEventDisableTiming));
	HIP_TRY(hipEventRecord(evFirstInv, gjStream));
}

void cuba_hip_solver::waitReport()
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

void cuba_hip_solver::downloadAsDouble(const Scalar* dsrc, double* hdst, size_t n)
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

void cuba_hip_solver::uploadFromDouble(Scalar* ddst, const double* hsrc, size_t n)
{
	if (!n) return;
	std::vector<Scalar> tmp(hsrc, hsrc + n);
	HIP_TRY(hipMemcpyAsync(ddst, tmp.data(), n * sizeof(Scalar), hipMemcpyHostToDevice, stream));
	sync();
}

double cuba_hip_solver::computeErrors()
{
	need();
	StageTimer tm(this, 2);
	launch_residual_chi2(g, d_parts.data(), slotsDev, nullptr, stream);   // its second stage writes all NSLOT entries of the slot group
	return readSlots(0);
}

void cuba_hip_solver::linearize(int mode, double lam, bool withBackup)
{
	waitAssembled();            // an overlapped coarse assembly may still be reading the previous reduced matrix
	const bool parts = mode == 1 && !redParts.empty();
	launch_linearize_dm(g, st, sys, mode, lam, stream, withBackup ? d_state.data() : nullptr, d_backup.data(), d_state.size(), parts ? &redParts[0].od : nullptr);
	if (parts && !partsByCaller)
		for (size_t c = 1; c < redParts.size(); c++) launch_block_pass(g, st, sys, redParts[c].od, stream);
}

void cuba_hip_solver::schurPart(int part, size_t ranges[4])
{
	need();
	if (part < 0 || part >= schurParts()) throw ArgError{ "cuba_hip_schur_part: no such part" };
	if (part == 0)
	{
		StageTimer tm(this, 4);
		zeroReduced();
		partsByCaller = true;
		try { linearize(1, lambda, false); } catch (...) { partsByCaller = false; throw; }
		partsByCaller = false;
	}
	else
	{
		StageTimer tm(this, 4);
		launch_block_pass(g, st, sys, redParts[part].od, stream);
	}
	const size_t nblk = (size_t)st.nblk;
	ranges[0] = redParts.empty() ? 0 : 36 * redParts[part].blkBegin;
	ranges[1] = redParts.empty() ? 36 * nblk : 36 * (redParts[part].blkEnd - redParts[part].blkBegin);
	ranges[2] = part == 0 ? 36 * nblk : 0;
	ranges[3] = part == 0 ? d_red.size() - 36 * nblk : 0;
}

void cuba_hip_solver::assemble()
{
	need();
	StageTimer tm(this, 3);
	zeroReduced(true);
	d_maxdiag.zero(stream);
	linearize(0, 0.0);
}

void cuba_hip_solver::maxDiagonalParts(double* posePart, double* lmPart)
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

void cuba_hip_solver::scaleParts(double lam, double* posePart, double* lmPart)
{
	need();
	launch_pose_scale(g, sys, lam, slotsDev + 3 * NSLOT, stream);
	launch_landmark_scale(g, sys, lam, slotsDev + 2 * NSLOT, stream);
	sync();
	double a = 0, b = 0;
	for (int i = 0; i < NSLOT; i++) { b += slot(2 * NSLOT + i); a += slot(3 * NSLOT + i); }
	*posePart = a; *lmPart = b;
}

double cuba_hip_solver::maxDiagonal()
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

void cuba_hip_solver::schur(bool withBackup)
{
	need();
	StageTimer tm(this, 4);
	zeroReduced();
	linearize(1, lambda, withBackup);
}

bool cuba_hip_solver::solveReduced() { return retryWithFp64Inverse(solveReducedOnce()); }

bool cuba_hip_solver::retryWithFp64Inverse(bool ok)
{
	// (only a loss of positive definiteness ALONG THE ITERATION, code 2, can come from the fp32 rounding of the coarse inverse: a NaN or a
	// diagonal block that is not positive definite would fail with fp64 storage just the same -- round-4 advisor)
	if (ok || !lastSolveBrokeDown || lastFailCode != 2 || !fp32Inverse() || sys.agg <= 0) return ok;
	precondFp32 = false; sys.acinv32 = nullptr;
	dropPcgGraph();
	coarseValid = false; firstInvValid = false; firstInvPending = false;
	cntFp32Fallbacks++;
	if (std::getenv("CUBA_HIP_DEBUG")) std::fprintf(stderr, "[cuba_hip] PCG broke down with the fp32-stored coarse inverse: repeating the solve with fp64 storage\n");
	const int64_t directBefore = cntDirect;
	const bool ok2 = solveReducedOnce();
	if (!ok2 || cntDirect != directBefore)
	{
		// the fp64-stored inverse did not make the PCG converge either: storage precision was not the cause, the handle keeps its fp32 copy
		precondFp32 = true; sys.acinv32 = d_coarse32[0].data();
		dropPcgGraph();
		coarseValid = false; firstInvValid = false; firstInvPending = false;
	}
	return ok2;
}

// PCG iterations a solve may use before it is handed to the exact solver.  An exact solve costs what 50 (KITTI-07 shape) ... 150-200
// (KITTI-00, S2M, G4M shapes) iterations cost -- a launch per level of the elimination tree against two or three launches per iteration,
// DESIGN.md section 4 --, and the hand-over waits for about twice that: Pf / 4 iterations, at least 128, at most 384.  Within a run the
// damping shrinks and the iteration count grows from solve to solve (x 1.3-2), so a solve that needs this many announces solves that
// need more; and the iteration count is the one measure of the system's conditioning the PCG has -- pcg_tol bounds the residual, the
// error of the increment is up to cond(M^-1 Hsc) times larger (on the KITTI-00-size Tukey start of tests/test_ref_lm.py the solves of
// 370 and 800 iterations at the default tolerance are what takes the run off the reference's: 2.6e-5 on chi2 with them, 3e-9 when they
// go to the exact solver).  A function of the graph's size alone: the decision follows from the call sequence, never from timing or from
// the memory that happens to be free (every rank of a partitioned run hands over at the same iteration).
int cuba_hip_solver::pcgBudget(int maxIter) const
{
	const int it = directAfter > 0 ? directAfter : std::min(384, std::max(128, Pf / 4));
	return std::min(maxIter, std::max(4, it / 4 * 4));
}

bool cuba_hip_solver::ensureDirectPlan()
{
	if (directPlanValid) return !directRefused;
	const auto t0 = Clock::now();
	ensureHostPattern();
	directPlanValid = true;
	if (!sparse_chol_plan(Pf, h_rowptr.data(), h_colind.data(), directSlack, (size_t)std::max(1, directMaxTiles), directPlan))
	{
		directRefused = true;
		char buf[200];
		std::snprintf(buf, sizeof buf, "exact reduced solve unavailable: the factor of the %d-pose reduced matrix needs more than direct_max_tiles = %d tiles", Pf, directMaxTiles);
		lastError = buf;
		return false;
	}
	const SparseCholPlan& p = directPlan;
	// all index arrays in one allocation, one copy
	std::vector<int> ints;
	auto put = [&](const std::vector<int>& v) { const size_t o = ints.size(); ints.insert(ints.end(), v.begin(), v.end()); while (ints.size() % 4) ints.push_back(0); return o; };
	const size_t oColPtr = put(p.colPtr), oRowIdx = put(p.rowIdx), oColOf = put(p.colOfTile), oGPtr = put(p.gPtr), oGather = put(p.gather),
		oWgRec = put(p.wgRec), oLvlCols = put(p.lvlCols), oBlkTile = put(p.blkTile), oPos = put(p.posOfSeg);
	d_scInts.upload(ints, stream);
	d_scTiles.resize((size_t)SC_TT * ((size_t)p.nTiles + 1)); d_scTilesT.resize((size_t)SC_TT * std::max(1, p.nTiles));
	d_scY.resize((size_t)SC_T * p.T); d_scRinv.resize((size_t)SC_T * p.T); d_scFail.resize(1);
	sync();                             // (`ints` leaves scope)
	SparseChol& d = directDev;
	d.tiles = d_scTiles.data(); d.tilesT = d_scTilesT.data(); d.y = d_scY.data(); d.rinv = d_scRinv.data(); d.fail = d_scFail.data();
	const int* base = d_scInts.data();
	d.colPtr = base + oColPtr; d.rowIdx = base + oRowIdx; d.colOfTile = base + oColOf; d.gPtr = base + oGPtr; d.gather = base + oGather;
	d.wgRec = base + oWgRec; d.lvlCols = base + oLvlCols; d.blkTile = base + oBlkTile; d.posOfSeg = base + oPos;
	d.T = p.T; d.Pf = Pf; d.nTiles = p.nTiles;
	directPlanSeconds = std::chrono::duration<double>(Clock::now() - t0).count();
	if (std::getenv("CUBA_HIP_DEBUG"))
		std::fprintf(stderr, "[cuba_hip] exact reduced solve, symbolic phase: %d poses -> %d tile columns, %d tiles (%.1f MB), %d levels, %lld tile products, slack %d, %.2f ms\n",
			Pf, p.T, p.nTiles, p.tileBytes() / 1e6, p.nLevels, p.entries, p.slack, 1e3 * directPlanSeconds);
	return true;
}

bool cuba_hip_solver::solveDirect()
{
	if (!ensureDirectPlan()) return false;
	const auto t0 = Clock::now();
	launch_sparse_chol_fill(st, sys, directDev, stream);
	launch_sparse_chol_solve(directDev, directPlan, sys.xp, stream);
	int* hflag = (int*)hostStage();
	HIP_TRY(hipMemcpyAsync(hflag, directDev.fail, sizeof(int), hipMemcpyDeviceToHost, stream));
	sync();
	directSeconds = std::chrono::duration<double>(Clock::now() - t0).count();
	cntDirect++;
	directSticky = true;
	coarseValid = false;
	if (std::getenv("CUBA_HIP_DEBUG"))
		std::fprintf(stderr, "[cuba_hip] exact reduced solve: %d unknowns, %.3f ms, lambda %.3e%s\n", 6 * Pf, 1e3 * directSeconds, lambda, *hflag ? " -- NON-POSITIVE PIVOT" : "");
	if (*hflag)
	{
		// the reference's failed factorisation (src/cuda_linear_solver.cpp:406-410): the LM loop rejects the trial and raises lambda
		cntDirectFailed++;
		lastError = "exact reduced solve: non-positive pivot (the damped reduced matrix is not positive definite)";
		return false;
	}
	return true;
}

// The reduced solve of one handle in three parts, so that cuba_hip_optimize_batch can run the middle one -- the iterations -- for several
// handles in one launch chain: solveBegin (set-up launch, coarse inverse schedule, first preconditioner application, batch-length
// prediction; true = the solve is finished, sc.result is its outcome), the iteration loop, solveEnd (bookkeeping, hand-over to the
// exact solver) / solveBrokeDown (the device reported a failure).
bool cuba_hip_solver::solveBegin(SolveCtx& sc)
{
	lastSolveBrokeDown = false;
	need();
	if (Pf == 0) { sc.result = true; return true; }
	const int maxIter = sc.maxIter = maxIterAlloc;
	const Scalar tol2 = sc.tol2 = pcgTol * pcgTol;
	if (failDirty) { d_fail.zero(stream); failDirty = false; }      // (the device flag only changes when a solve fails, and every solve reports it)
	const bool twoLevel = sc.twoLevel = sys.agg > 0;
	const bool direct = sc.direct = directUsable();
	if (direct && (directSticky || directAlways))
	{
		// an earlier solve of this run needed the exact solver (or the caller wants every solve exact): this one goes there at once
		// (the set-up launch damps the diagonal blocks; it may raise the device failure flag for a diagonal block that is not positive
		// definite, which the next PCG solve must not inherit)
		launch_pcg_setup(g, st, sys, lambda, stream);
		failDirty = true;
		coarseValid = false;
		if (pcgHistory.size() >= 65536) pcgHistory.erase(pcgHistory.begin(), pcgHistory.begin() + 32768);
		pcgHistory.push_back(0);
		sc.result = solveDirect();
		return true;
	}
	// iterations before the solve is handed to the exact solver (a caller who asked for the best iterate at max_iter gets max_iter)
	sc.budget = direct && !acceptUnconverged ? pcgBudget(maxIter) : maxIter;
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
	// (cuba_hip_optimize_batch issues this launch and the first preconditioner application for all its graphs at once -- except for the
	// first solve of a run, which puts a copy or an in-line inversion between the two)
	const bool deferred = sc.deferLaunch && twoLevel && coarseValid;
	{
		const bool f32 = fp32Inverse();      // the overlapped inversion left an fp32 copy in the staging buffer: that is what moves into the buffer in use
		const Scalar* cpSrc = !takeInverse ? nullptr : f32 ? reinterpret_cast<const Scalar*>(d_coarse32[1].data()) : d_coarse[0].data();
		Scalar* cpDst = f32 ? reinterpret_cast<Scalar*>(d_coarse32[0].data()) : d_coarse[2].data();
		const size_t cpCount = f32 ? inv32Count() / 2 : invCount;
		if (deferred) { sc.deferred = true; sc.copySrc = cpSrc; sc.copyDst = cpDst; sc.copyCount = cpCount; }
		else launch_pcg_setup_expand(g, st, sys, lambda, stream, cpSrc, cpDst, cpCount);
	}
	if (twoLevel)
	{
		// the sweep ping-pongs between two buffers: start in the one that leaves the inverse in d_coarse[0]
		const int gjSteps = (6 * sys.cl * sys.nc + 31) / 32, first = gjSteps & 1;
		{
			// The inverse in use lives in d_coarse[2] (the iteration graphs have the pointer baked in); d_coarse[0 / 1] are
			// the work buffers of the sweep, which leaves its result in d_coarse[0].
			ensureOverlapObjects();
			const size_t invBytes = sizeof(Scalar) * (size_t)36 * sys.cl * sys.cl * sys.nc * sys.nc;
			if (!coarseValid && heuristics && firstInvValid)
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
				else launch_coarse_finish(d_coarse[0].data(), d_coarse[2].data(), 6 * sys.cl * sys.nc, stream);
				coarseValid = true; cntCoarseRefresh++; cntCoarseInline++; sideAge = 0;
				if (heuristics)
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
				if (!deferred) HIP_TRY(hipEventRecord(evSetup, stream));        // (deferred: the batch's common event behind the batched set-up launch)
				if (sc.deferCoarse)
				{
					// cuba_hip_optimize_batch sweeps the coarse matrices of all its graphs together (one launch per step for all of them, on
					// the first handle's side stream): only the decision is taken here, launchCoarseJobs enqueues the work
					sc.deferCoarse->push_back(CoarseJob{ this, first, firstInvPending, !deferred });
					if (firstInvPending) { firstInvPending = false; firstInvValid = true; }
				}
				else
				{
					HIP_TRY(hipStreamWaitEvent(gjStream, evSetup, 0));
					(void)launch_coarse_setup(g, st, sys, d_coarse[first].data(), d_coarse[1 - first].data(), gjStream, evAssembled);
					if (fp32Inverse()) launch_coarse_to_fp32(d_coarse[0].data(), d_coarse32[1].data(), 6 * sys.cl * sys.nc, gjStream);   // (staging: the iteration graphs read [0])
					else launch_coarse_finish(d_coarse[0].data(), d_coarse[0].data(), 6 * sys.cl * sys.nc, gjStream);                     // (the sweep leaves -inverse in the upper triangle)
					if (firstInvPending)
					{
						if (fp32Inverse()) HIP_TRY(hipMemcpyAsync(d_firstInv32.data(), d_coarse32[1].data(), inv32Count() * sizeof(float), hipMemcpyDeviceToDevice, gjStream));
						else HIP_TRY(hipMemcpyAsync(d_firstInv.data(), d_coarse[0].data(), invBytes, hipMemcpyDeviceToDevice, gjStream));
						HIP_TRY(hipEventRecord(evFirstInv, gjStream));
						firstInvPending = false; firstInvValid = true;
					}
					HIP_TRY(hipEventRecord(evInverse, gjStream));
				}
				pendingInv = 0;
				assemblePending = true; cntCoarseRefresh++;
			}
		}
		if (!deferred) launch_pcg2_fused(g, sys, 0, 0, maxIter, tol2, 0, stream);
	}
	// first solve on this structure: the usual chunk lengths are ordered at once (the helper thread builds them while this solve runs
	// on plain launches), longest first -- the first batches of a run are the long ones
	// (graphs only while this is the one handle of the process: see g_liveHandles)
	const bool graphs = sc.graphs = useGraph && !sc.batched && g_liveHandles.load(std::memory_order_relaxed) <= 1;
	if (useGraph && !graphs && (!pcgGraphs.empty() || graphsOrderedFor)) dropPcgGraph();
	if (graphs && pcgGraphs.empty() && graphsOrderedFor != (const void*)sys.acinv)
	{
		for (int c = 64; c >= 4; c /= 2) (void)pcgGraphIfReady(c, maxIter, tol2);
		graphsOrderedFor = (const void*)sys.acinv;
	}
	sc.hInts = (volatile int*)((char*)h_pinned + 1024);   // fail, iterations done, stop flag: written by the device
	sc.tSolve0 = Clock::now();
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
		if (heuristics && k < prevRunIters.size() && std::equal(runIters.begin(), runIters.end(), prevRunIters.begin())) predicted = prevRunIters[k];
	}
	// (the last node of every iteration graph runs the stop test on the residual its chunk left: a batch of exactly the needed
	// length is recognised as converged)
	sc.predicted = predicted;
	return false;
}

bool cuba_hip_solver::solveBrokeDown(SolveCtx& sc)
{
	lastSolveBrokeDown = true; lastFailCode = sc.hInts[0];
	cntPcgIters += sc.hInts[1]; coarseValid = false; firstInvValid = false; firstInvPending = false; failDirty = true;
	// (code 1, a diagonal block that is not positive definite, fails a Cholesky factorisation just the same; code 2 with the
	// fp32-stored inverse is first repeated with fp64 storage by solveReduced)
	if (sc.direct && lastFailCode != 1 && !(lastFailCode == 2 && fp32Inverse() && sc.twoLevel))
	{
		if (pcgHistory.size() >= 65536) pcgHistory.erase(pcgHistory.begin(), pcgHistory.begin() + 32768);
		pcgHistory.push_back(-sc.hInts[1]);
		d_fail.zero(stream); failDirty = false;
		return solveDirect();
	}
	return false;
}

bool cuba_hip_solver::solveReducedOnce()
{
	StageTimer tm(this, 6);
	SolveCtx sc;
	if (solveBegin(sc)) return sc.result;
	const int maxIter = sc.maxIter; const Scalar tol2 = sc.tol2; const bool graphs = sc.graphs; const int predicted = sc.predicted;
	// Iterations are enqueued in chunks (graphs of 4/8/.../256 iterations; chunk lengths are multiples of 4
	// because the kernels address their reduction slots by the chunk-local k & 3) and the host looks at the device
	// stop flag after each batch.  A launch after convergence still costs ~2.5 us per kernel and a look costs a
	// host round trip, so the first batch is sized from the previous solve of this run.
	bool converged = false;
	int k0 = 0, looks = 0, eagerIters = 0;
	int target = (predicted + 3) / 4 * 4;
	while (k0 < sc.budget && !converged)
	{
		int todo = std::max(4, std::min(target, sc.budget) - k0);
		while (todo > 0)
		{
			// without hipGraphs (the default): the whole batch as one chunk of plain launches -- chunk-local iteration numbers, then the
			// advance / stop test / report launch, so that a batch of exactly the needed length is recognised as converged without a
			// further look; with hipGraphs: the largest of 256, 128, ..., 4 that fits (few graphs per batch, each hand-over costs ~9 us)
			int c = graphs ? 256 : todo;
			for (; graphs && c > 4 && c > todo; c >>= 1) {}
			// a batch length that comes back (repeated runs on one structure) gets a graph of exactly that length: one hand-over per
			// batch instead of one per power of two.  Not on the first request -- the reference's timing protocol meets most lengths for
			// the first time inside its timed part, and an instantiation costs ~2 us per node.
			// (the exact graph is ordered on the second request and used from the moment it exists)
			if (graphs && todo > c && todo % 4 == 0 && todo <= 128 && pcgGraphMaxIter == maxIter && pcgGraphTol2 == tol2)
			{
				if ((pcgGraphs.count(std::make_pair(todo, (const Scalar*)sys.acinv)) || ++batchRequests[todo] >= 2) && pcgGraphIfReady(todo, maxIter, tol2)) c = todo;
			}
			hipGraphExec_t exec = graphs ? pcgGraphIfReady(c, maxIter, tol2) : nullptr;
			if (exec) { HIP_TRY(hipGraphLaunch(exec, stream)); noteReport(); }     // (every graph reports; the host waits for the last)
			else
			{
				// the chunk as plain launches: chunk-local iteration numbers + the advance / stop test / report node
				for (int k = 0; k < c; k++) enqueuePcgIteration(k, maxIter, tol2, stream);
				launch_pcg_advance(sys, c, stream, tol2); noteReport();
				if (graphs) { eagerIters += c; cntPcgPlain += c; }
			}
			k0 += c; todo -= c;
		}
		waitReport();
		if (sc.hInts[0] != 0) return solveBrokeDown(sc);
		if (sc.hInts[2] != 0 || sc.hInts[1] < std::min(k0, maxIter)) converged = true;   // the device-side stop test fired
		target = k0 + ((looks == 0 && k0 <= 96) ? 4 : std::max(8, k0 / 8 / 4 * 4));   // (a batch sized from the run's own history misses by a few iterations at most)
		looks++; cntPcgLooks++; cntHostLooks++;
	}
	sc.converged = converged; sc.k0 = k0; sc.looks = looks; sc.eagerIters = eagerIters;
	return solveEnd(sc);
}

bool cuba_hip_solver::solveEnd(SolveCtx& sc)
{
	const int maxIter = sc.maxIter; const bool direct = sc.direct; const bool converged = sc.converged;
	const int k0 = sc.k0, looks = sc.looks, eagerIters = sc.eagerIters, predicted = sc.predicted;
	const auto tSolve1 = Clock::now();
	if (std::getenv("CUBA_HIP_DEBUG"))
	{
		// (debug only: r_0.z_0 of this solve from the partial sums the first preconditioner application left in slot 0)
		std::vector<Scalar> part((size_t)std::max(1, sys.nrz0));
		HIP_TRY(hipMemcpyAsync(part.data(), sys.rz, sizeof(Scalar) * part.size(), hipMemcpyDeviceToHost, stream));
		sync();
		double rz0 = 0; for (Scalar v : part) rz0 += (double)v;
		std::fprintf(stderr, "[cuba_hip] PCG: %d iterations, %d enqueued (%d as plain launches), %d host looks (prediction %d), lambda %.3e, r0.z0 %.6e, solve %.3f ms, graphs built so far %lld\n",
			sc.hInts[1], k0, eagerIters, looks, predicted, lambda, rz0, 1e3 * std::chrono::duration<double>(tSolve1 - sc.tSolve0).count(), (long long)gb.builds.load());
	}
	const int itersDone = sc.itersDone >= 0 ? sc.itersDone : sc.hInts[1];
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
		coarseValid = false;
		if (direct && !acceptUnconverged) return solveDirect();      // exact solve instead of a failure report (ba_direct.hip)
		char buf[160];
		std::snprintf(buf, sizeof buf, "PCG stopped at max_iter = %d without reaching pcg_tol = %g", maxIter, pcgTol);
		lastError = buf;
		return acceptUnconverged;
	}
	return true;
}

void cuba_hip_solver::backSubstitute()
{
	need();
	StageTimer tm(this, 4);
	launch_back_substitute(g, st, sys, lambda, stream);
}

bool cuba_hip_solver::solve()
{
	schur();
	if (!solveReduced()) return false;
	backSubstitute();
	return true;
}

void cuba_hip_solver::update()
{
	need();
	StageTimer tm(this, 7);
	launch_update_state(g, sys, stream);
}

double cuba_hip_solver::computeScale(double lam)
{
	need();
	double a = 0, b = 0;
	scaleParts(lam, &a, &b);
	return a + b;
}

// ---- the device-decided LM run in steps (shared by cuba_hip_optimize and cuba_hip_optimize_batch) ---------------------------------------
void cuba_hip_solver::lmRunBegin(LmRun& r, int niter, double* chi2Out)
{
	const double tau = 1e-5;
	if (!h_lmRing)
	{
		// (coherent, like the flag block: the host reads a record as soon as it sees the ticket that follows it -- from the very kernel that wrote
		// the record, which has not ended then; memory mapped without the flag is only guaranteed to show a kernel's stores once the kernel has
		// completed, and a run then now and again reported 0 or a stale value as an iteration's chi2 while its estimates were right)
		HIP_TRY(hipHostMalloc((void**)&h_lmRing, sizeof(double) * LM_RING * LM_REC, hipHostMallocMapped | hipHostMallocCoherent));
		HIP_TRY(hipHostGetDevicePointer((void**)&lmRingDev, h_lmRing, 0));
	}
	coarseValid = false;
	startRunHistory();
	r = LmRun();
	r.niter = niter; r.chi2Out = chi2Out; r.stop = niter <= 0;
	r.F = computeErrors();
	r.lam = tau * maxDiagonal();
	{
		double* st8 = reinterpret_cast<double*>(hostStage());          // (pinned staging block)
		st8[0] = r.F; st8[1] = r.lam; st8[2] = 2.0; st8[3] = 0.0; st8[4] = 0.0; st8[5] = 1.0; st8[6] = 0.0; st8[7] = (double)LmRun::maxq;
		r.tagBase = 65536.0 * (double)(++lmRunNonce); st8[8] = r.tagBase;      // (every record of this run carries tagBase + trial number + 1)
		Scalar* l1 = reinterpret_cast<Scalar*>(st8 + 12);
		l1[0] = (Scalar)r.lam;
		HIP_TRY(hipMemcpyAsync(d_lmState.data(), st8, sizeof(double) * 9, hipMemcpyHostToDevice, stream));
		HIP_TRY(hipMemcpyAsync(d_lamS.data(), l1, sizeof(Scalar), hipMemcpyHostToDevice, stream));
	}
	r.lm.state = d_lmState.data(); r.lm.lam = d_lamS.data(); r.lm.ring = lmRingDev;
}

// outcomes of the trials [seen, upto): every one of them is complete (the caller has waited for a report that follows them in the stream)
void cuba_hip_solver::lmAbsorb(LmRun& r, int upto)
{
	std::atomic_thread_fence(std::memory_order_acquire);
	for (; r.seen < upto && !r.stop; r.seen++)
	{
		const volatile double* rec = h_lmRing + (size_t)(r.seen % LM_RING) * LM_REC;
		// the record must be THIS run's record of THIS trial.  The ticket the caller waited for is stored after it by the same device thread
		// behind a system-scope release (publish_report) -- and still, once in a few hundred short runs, a record had not landed when its
		// ticket had (round 6: an iteration's chi2 came back as 0): whatever reorders the two on the way to host memory, a stream
		// synchronisation ends it
		const double tag = r.tagBase + (double)(r.seen + 1);
		if (rec[7] != tag)
		{
			cntLateRecords++;
			sync();
			std::atomic_thread_fence(std::memory_order_acquire);
			if (rec[7] != tag) throw StateError{ "the device's decision record of an LM trial never arrived" };
		}
		const bool acc = rec[5] != 0.0;
		const double rho = rec[2];
		r.lam = rec[3]; r.F = rec[4];
		if (acc) r.rejRun = 0; else r.rejRun++;
		// the reference's loops: an iteration ends with an accepted trial, with the maxq-th rejection, or with a rejected trial whose
		// rho is not < 0; the run ends after niter iterations, or on `qn == maxq || rho <= 0 || !isfinite(lambda)` (:851)
		const bool iterationEnds = acc || r.rejRun == LmRun::maxq || !(rho < 0);
		if (!iterationEnds) continue;
		if (r.chi2Out) r.chi2Out[r.done] = r.F;
		r.done++;
		if (r.done == r.niter || r.rejRun == LmRun::maxq || rho <= 0 || !std::isfinite(r.lam)) r.stop = true;
		r.rejRun = 0;
	}
}

// before trial r.enq is enqueued: false = the run has ended
bool cuba_hip_solver::lmBeforeTrial(LmRun& r)
{
	if (r.stop) return false;
	if (r.enq > r.seen)
	{
		// trial enq - 1 is still undecided as far as the host knows.  Trial enq is needed whatever its outcome -- unless that
		// outcome can end the run: the last iteration, or the maxq-th rejection in a row
		const bool safe = r.done + 1 < r.niter && r.rejRun + 1 < LmRun::maxq;
		if (!safe) { waitReport(); cntHostLooks++; lmAbsorb(r, r.enq); if (r.stop) return false; }
	}
	cntTrials++;
	lambda = -1.0;                 // (every kernel of the trial reads the damping from device memory: launch_lambda)
	return true;
}

// after the reduced solve of trial r.enq (whose host looks follow every earlier trial's decision in the stream): false = the run has ended
bool cuba_hip_solver::lmAfterSolveHost(LmRun& r)
{
	lmAbsorb(r, r.enq);
	if (r.stop) { cntTrials--; return false; }      // (an outcome nobody could foresee ended the run: the device has halted, this trial is void)
	return true;
}

bool cuba_hip_solver::lmAfterSolve(LmRun& r, bool ok)
{
	if (!lmAfterSolveHost(r)) return false;
	if (ok) launch_trial_tail_fused(g, st, sys, (Scalar)-1, d_backup.data(), stream, &r.lm);
	else launch_lm_decide_failed(sys, r.lm, stream);
	noteReport();
	launch_restore_if_rejected(d_state.data(), d_backup.data(), d_state.size(), r.lm, stream);
	r.enq++;
	(void)hipStreamQuery(stream);
	return true;
}

int cuba_hip_solver::lmRunEnd(LmRun& r)
{
	if (r.seen < r.enq) { waitReport(); cntHostLooks++; lmAbsorb(r, r.enq); }
	lambda = r.lam;
	return r.done;
}

int cuba_hip_solver::optimizeDeviceDecision(int niter, double* chi2Out)
{
	LmRun r;
	lmRunBegin(r, niter, chi2Out);
	while (lmBeforeTrial(r))
	{
		schur(true);
		const bool ok = solveReduced();
		if (!lmAfterSolve(r, ok)) break;
	}
	return lmRunEnd(r);
}

// every launch of a trial batched (one stream for all graphs): the standard launch sequence only
bool cuba_hip_solver::fullyBatchable() const
{
	return batchable() && st.nBig == 0 && st.nDiagProd == 0 && st.nOd > 0 && st.nWaves > 0 && redParts.empty();
}

bool cuba_hip_solver::batchable() const
{
	return !profile && partHi < 0 && Pf > 0 && Lf > 0 && E > 0 && trial_tail_parts(g, st) <= d_parts.size() && batch_kernel_class(g, sys) >= 0;
}

// The overlapped coarse inversions a batch's handles decided on in this trial (solveBegin, `regular` schedule), enqueued together on
// THIS handle's side stream: per graph what the solo path enqueues -- wait for its set-up launch, zero + assemble, [sweep], conversion,
// copy for the next run's first solve, events -- with the sweeps of all graphs as one launch per step (launch_dense_inverse_batch).
void cuba_hip_solver::launchCoarseJobs(std::vector<CoarseJob>& jobs, hipEvent_t common)
{
	if (jobs.empty()) return;
	ensureOverlapObjects();
	hipStream_t side = gjStream;
	const int m = (int)jobs.size();
	if (!h_gjTab) { HIP_TRY(hipHostMalloc((void**)&h_gjTab, sizeof(GjJob) * CUBA_HIP_BATCH_MAX, hipHostMallocDefault)); HIP_TRY(hipEventCreateWithFlags(&evGjTab, hipEventDisableTiming)); HIP_TRY(hipEventRecord(evGjTab, side)); }
	d_gjTab.resize(sizeof(GjJob) * CUBA_HIP_BATCH_MAX);
	HIP_TRY(hipEventSynchronize(evGjTab));          // (the previous table's copy has long left the pinned block)
	int nMax = 0;
	for (int a = 0; a < m; a++)
	{
		cuba_hip_solver* h = jobs[a].h;
		const int Nc = 6 * h->sys.cl * h->sys.nc;
		HIP_TRY(hipStreamWaitEvent(side, jobs[a].ownEvent || !common ? h->evSetup : common, 0));
		launch_coarse_assemble(h->g, h->st, h->sys, h->d_coarse[jobs[a].first].data(), side);
		GjJob& j = h_gjTab[a];
		j.buf[0] = h->d_coarse[jobs[a].first].data(); j.buf[1] = h->d_coarse[1 - jobs[a].first].data();
		j.piv[0] = h->sys.gj_pivots; j.piv[1] = h->sys.gj_pivots + 32 * 32;
		j.n = Nc; j.tiles = (Nc + 31) / 32;
		nMax = std::max(nMax, Nc);
	}
	for (int a = 0; a < m; a++) HIP_TRY(hipEventRecord(jobs[a].h->evAssembled, side));      // from here on no sweep reads a reduced matrix
	HIP_TRY(hipMemcpyAsync(d_gjTab.data(), h_gjTab, sizeof(GjJob) * m, hipMemcpyHostToDevice, side));
	HIP_TRY(hipEventRecord(evGjTab, side));
	launch_dense_inverse_batch(reinterpret_cast<const GjJob*>(d_gjTab.data()), m, nMax, side);
	for (int a = 0; a < m; a++)
	{
		cuba_hip_solver* h = jobs[a].h;
		const int Nc = 6 * h->sys.cl * h->sys.nc;
		const size_t invBytes = sizeof(Scalar) * (size_t)Nc * Nc;
		// (the sweep of Nc / 32 steps started in d_coarse[first] with first = steps & 1: its result is in d_coarse[0])
		if (h->fp32Inverse()) launch_coarse_to_fp32(h->d_coarse[0].data(), h->d_coarse32[1].data(), Nc, side);
		else launch_coarse_finish(h->d_coarse[0].data(), h->d_coarse[0].data(), Nc, side);
		if (jobs[a].firstInvCopy)
		{
			if (h->fp32Inverse()) HIP_TRY(hipMemcpyAsync(h->d_firstInv32.data(), h->d_coarse32[1].data(), h->inv32Count() * sizeof(float), hipMemcpyDeviceToDevice, side));
			else HIP_TRY(hipMemcpyAsync(h->d_firstInv.data(), h->d_coarse[0].data(), invBytes, hipMemcpyDeviceToDevice, side));
			HIP_TRY(hipEventRecord(h->evFirstInv, side));
		}
		HIP_TRY(hipEventRecord(h->evInverse, side));
	}
	jobs.clear();
}

// cuba_hip_optimize_batch, every launch of a trial batched: all graphs share the first handle's stream for the duration of the call, and
// per trial the host issues -- for ALL graphs together -- landmark pass + Schur pass, the PCG set-up launch, the first preconditioner
// application, the iterations, the trial tail + sums / decision / report + conditional restore, and on the side stream one sweep of all
// coarse matrices that are due.  What stays per graph: host bookkeeping, the first solve of a run (a copy or an in-line inversion sits
// between its set-up launch and its first preconditioner application), exact solves, failed solves.
static int cuba_hip_optimize_batch_full(cuba_hip_solver** hs, int n, int niter, double* chi2, int* nDone)
{
	cuba_hip_solver* lead = hs[0];
	hipStream_t bs = lead->stream;
	struct StreamGuard
	{
		cuba_hip_solver** hs; int n; std::vector<hipStream_t> own;
		StreamGuard(cuba_hip_solver** h, int m, hipStream_t s) : hs(h), n(m), own((size_t)m)
		{
			for (int i = 0; i < n; i++) { own[i] = hs[i]->stream; (void)hipStreamSynchronize(own[i]); hs[i]->stream = s; }
		}
		~StreamGuard() { if (n > 0) (void)hipStreamSynchronize(hs[0]->stream); for (int i = 0; i < n; i++) hs[i]->stream = own[i]; }
	} streamGuard(hs, n, bs);
	std::vector<cuba_hip_solver::LmRun> runs((size_t)n);
	std::vector<cuba_hip_solver::SolveCtx> ctx((size_t)n);
	lead->ensureOverlapObjects();
	std::vector<cuba_hip_solver::CoarseJob> coarseJobs;
	// three tables (pinned host + device), one per phase of a trial: a phase's copy has long executed when the next trial rewrites it
	// (the host waits for the PCG reports of every trial in between)
	if (lead->h_batchTab && lead->batchTabEntries < 3 * CUBA_HIP_BATCH_MAX) { (void)hipHostFree(lead->h_batchTab); lead->h_batchTab = nullptr; }
	if (!lead->h_batchTab) { HIP_TRY(hipHostMalloc((void**)&lead->h_batchTab, sizeof(BatchEntry) * CUBA_HIP_BATCH_MAX * 3, hipHostMallocDefault)); lead->batchTabEntries = 3 * CUBA_HIP_BATCH_MAX; }
	lead->d_batchTab.resize(sizeof(BatchEntry) * CUBA_HIP_BATCH_MAX * 3);
	BatchEntry* hTab[3]; const BatchEntry* dTab[3];
	for (int k = 0; k < 3; k++) { hTab[k] = lead->h_batchTab + (size_t)k * CUBA_HIP_BATCH_MAX; dTab[k] = reinterpret_cast<const BatchEntry*>(lead->d_batchTab.data()) + (size_t)k * CUBA_HIP_BATCH_MAX; }
	auto upload = [&](int k, int m) { HIP_TRY(hipMemcpyAsync(const_cast<BatchEntry*>(dTab[k]), hTab[k], sizeof(BatchEntry) * m, hipMemcpyHostToDevice, bs)); };
	for (int i = 0; i < n; i++) hs[i]->lmRunBegin(runs[i], niter, chi2 ? chi2 + (size_t)i * niter : nullptr);
	const bool mixed = hs[0]->st.mixed != 0;
	std::vector<int> act;
	std::vector<char> okv((size_t)n), inPcg((size_t)n);
	int batchSolves = 0;
	for (;;)
	{
		act.clear();
		for (int i = 0; i < n; i++) if (hs[i]->lmBeforeTrial(runs[i])) act.push_back(i);
		if (act.empty()) break;
		const int m = (int)act.size();
		// ---- phase A: linearise + Schur (the state backup rides in the landmark pass's launch)
		unsigned lmMax = 0, schurMax = 0;
		for (int a = 0; a < m; a++)
		{
			cuba_hip_solver* h = hs[act[a]];
			h->zeroReduced();
			BatchEntry& e = hTab[0][a];
			e = BatchEntry();
			e.g = h->g; e.st = h->st; e.sys = h->sys;
			batch_fill_linearize(h->g, h->st, h->d_state.data(), h->d_backup.data(), h->d_state.size(), e.t);
			lmMax = std::max(lmMax, e.t.lmGrid); schurMax = std::max(schurMax, e.t.schurGrid);
		}
		upload(0, m);
		launch_batch_linearize(dTab[0], m, lmMax, schurMax, mixed, bs);
		// ---- phase B: front half of the reduced solves
		unsigned setupMax = 0; int gridSpmvMax = 0, ncMax = 0; size_t ldsMax = 0; int target = 4, budgetMax = 0, nPcg = 0, nFused = 0;
		Scalar tol2 = 0;
		for (int a = 0; a < m; a++)
		{
			const int i = act[a];
			cuba_hip_solver* h = hs[i];
			ctx[i] = cuba_hip_solver::SolveCtx(); ctx[i].batched = true; ctx[i].deferCoarse = &coarseJobs; ctx[i].deferLaunch = true;
			BatchEntry& e = hTab[1][a];
			e = BatchEntry();
			const bool finished = h->solveBegin(ctx[i]);
			e.g = h->g; e.st = h->st; e.sys = h->sys;
			if (finished) { okv[i] = ctx[i].result; inPcg[i] = 0; continue; }      // (an exact solve at once: the graph sits the batched launches out)
			inPcg[i] = 1; nPcg++;
			tol2 = ctx[i].tol2;
			e.maxIter = ctx[i].maxIter;
			e.gridSpmv = (h->g.Pf + h->sys.spmv_rows - 1) / h->sys.spmv_rows;
			if (ctx[i].deferred) { batch_fill_setup(h->g, h->st, h->sys, ctx[i].copySrc, ctx[i].copyDst, ctx[i].copyCount, e.t); e.t.fusedOn = 1; nFused++; }
			setupMax = std::max(setupMax, e.t.setupGrid);
			gridSpmvMax = std::max(gridSpmvMax, e.gridSpmv); ncMax = std::max(ncMax, h->sys.nc); ldsMax = std::max(ldsMax, batch_pcg2_lds_bytes(h->sys));
			target = std::max(target, (ctx[i].predicted + 3) / 4 * 4); budgetMax = std::max(budgetMax, ctx[i].budget);
		}
		if (nPcg > 0)
		{
			upload(1, m);
			if (nFused > 0) launch_batch_setup(dTab[1], m, setupMax, bs);
			HIP_TRY(hipEventRecord(lead->evSetup, bs));
			lead->launchCoarseJobs(coarseJobs, lead->evSetup);
			if (nFused > 0) launch_batch_first_precond(dTab[1], m, lead->sys, ncMax, ldsMax, tol2, bs);
			// ---- the iterations of all graphs
			int k0 = 0, looks = 0;
			std::vector<char> conv((size_t)n, 0), broke((size_t)n, 0);
			bool all = false;
			while (k0 < budgetMax && !all)
			{
				int todo = std::max(4, std::min(target, budgetMax) - k0);
				while (todo > 0)
				{
					const int c = todo;
					for (int k = 0; k < c; k++) launch_pcg_batch_iteration(dTab[1], m, lead->g, lead->sys, gridSpmvMax, ncMax, ldsMax, k, tol2, bs);
					launch_pcg_batch_advance(dTab[1], m, c, bs, tol2);
					for (int i : act) if (inPcg[i]) hs[i]->noteReport();
					k0 += c; todo -= c;
				}
				all = true;
				for (int i : act)
				{
					if (!inPcg[i]) continue;
					cuba_hip_solver* h = hs[i];
					h->waitReport();
					const volatile int* f = ctx[i].hInts;
					if (f[0] != 0) broke[i] = 1;
					else if (f[2] != 0 || f[1] < std::min(k0, ctx[i].maxIter)) conv[i] = 1;
					if (!broke[i] && !conv[i] && k0 < ctx[i].budget) all = false;
					h->cntPcgLooks++; h->cntHostLooks++;
				}
				target = k0 + ((looks == 0 && k0 <= 96) ? 4 : std::max(8, k0 / 8 / 4 * 4));
				looks++;
			}
			for (int i : act)
			{
				if (!inPcg[i]) continue;
				cuba_hip_solver::SolveCtx& sc = ctx[i];
				sc.k0 = std::min(k0, std::max(sc.budget, 4)); sc.looks = looks; sc.eagerIters = 0;
				// (a graph alone stops at ITS iteration budget and hands the solve over; in a batch its iterations run on while the others
				// need them -- harmless, the exact solve replaces the iterate -- but convergence counts only within the budget)
				sc.converged = conv[i] != 0 && sc.hInts[1] <= sc.budget;
				if (!sc.converged && !broke[i]) sc.itersDone = std::min((int)sc.hInts[1], sc.budget);
				okv[i] = hs[i]->retryWithFp64Inverse(broke[i] ? hs[i]->solveBrokeDown(sc) : hs[i]->solveEnd(sc));
			}
			batchSolves++;
		}
		else lead->launchCoarseJobs(coarseJobs, nullptr);
		// ---- phase C: trial tails, decisions, reports, conditional restores
		unsigned tailMax = 0, restoreMax = 0;
		for (int a = 0; a < m; a++)
		{
			const int i = act[a];
			cuba_hip_solver* h = hs[i];
			BatchEntry& e = hTab[2][a];
			e = BatchEntry();
			e.g = h->g; e.st = h->st; e.sys = h->sys;
			if (!h->lmAfterSolveHost(runs[i])) continue;          // (the run has ended: nothing of this trial is launched)
			batch_fill_tail(h->g, h->st, h->sys, h->d_backup.data(), runs[i].lm, h->d_state.data(), h->d_state.size(), e.t);
			e.t.backupDst = h->d_backup.data();
			if (!okv[i])
			{
				// (a failed reduced solve: the decision alone, at once; the batched restore launch follows it in the stream)
				launch_lm_decide_failed(h->sys, runs[i].lm, bs);
				e.t.tailGrid = 0; e.t.reportOn = 0;
			}
			tailMax = std::max(tailMax, e.t.tailGrid); restoreMax = std::max(restoreMax, e.t.restoreGrid);
			h->noteReport();
			runs[i].enq++;
		}
		upload(2, m);
		launch_batch_tail(dTab[2], m, tailMax, restoreMax, bs);
		(void)hipStreamQuery(bs);
	}
	for (int i = 0; i < n; i++) nDone[i] = hs[i]->lmRunEnd(runs[i]);
	return batchSolves;
}

// Several graphs, one launch chain (include/cuba_hip.h: cuba_hip_optimize_batch).  Every handle runs ITS OWN Levenberg-Marquardt loop --
// decisions on the device, per graph -- and everything of a trial but the PCG iterations on its own stream (linearise + Schur, set-up
// launch, coarse inverse, trial tail: they overlap across the streams); the iterations of all graphs go out as ONE chain of batched
// launches (blockIdx.y = graph, kernel arguments from a device table) on the first handle's stream, joined and forked by events.  Per
// graph the kernels, their arguments and their order are those of cuba_hip_optimize: results are bit-identical to the solo runs.  A graph
// whose solve has converged returns at once from the iterations the others still need (its device-side stop flag), a graph whose run has
// ended leaves the batch.  The host issues 2 launches per iteration for the whole batch instead of 2 per graph -- it is the host's launch
// rate, not the GPU, that bounds several handles driven side by side.
int cuba_hip_optimize_batch_impl(cuba_hip_solver** hs, int n, int niter, double* chi2, int* nDone)
{
	bool together = n > 1;
	for (int i = 0; i < n; i++) hs[i]->need();
	for (int i = 0; i < n && together; i++)
		together = hs[i]->batchable() && hs[i]->device == hs[0]->device && batch_kernel_class(hs[i]->g, hs[i]->sys) == batch_kernel_class(hs[0]->g, hs[0]->sys) &&
			hs[i]->pcgTol == hs[0]->pcgTol && (i == 0 || hs[i]->stream != hs[0]->stream);
	if (!together)
	{
		for (int i = 0; i < n; i++) nDone[i] = hs[i]->optimize(niter, chi2 ? chi2 + (size_t)i * niter : nullptr);
		return 0;
	}
	bool full = true;
	for (int i = 0; i < n && full; i++) full = hs[i]->fullyBatchable() && hs[i]->st.mixed == hs[0]->st.mixed;
	if (full) return cuba_hip_optimize_batch_full(hs, n, niter, chi2, nDone);
	cuba_hip_solver* lead = hs[0];
	hipStream_t bs = lead->stream;
	std::vector<cuba_hip_solver::LmRun> runs((size_t)n);
	std::vector<cuba_hip_solver::SolveCtx> ctx((size_t)n);
	std::vector<hipEvent_t>& evJoin = lead->batchEvents;
	while ((int)evJoin.size() < n + 1) { hipEvent_t e; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); evJoin.push_back(e); }
	hipEvent_t evFork = evJoin[n];
	if (!lead->h_batchTab) { HIP_TRY(hipHostMalloc((void**)&lead->h_batchTab, sizeof(BatchEntry) * CUBA_HIP_BATCH_MAX * 3, hipHostMallocDefault)); lead->batchTabEntries = 3 * CUBA_HIP_BATCH_MAX; }
	lead->d_batchTab.resize(sizeof(BatchEntry) * CUBA_HIP_BATCH_MAX * 3);
	BatchEntry* hTab = lead->h_batchTab;
	const BatchEntry* dTab = reinterpret_cast<const BatchEntry*>(lead->d_batchTab.data());
	lead->ensureOverlapObjects();
	std::vector<cuba_hip_solver::CoarseJob> coarseJobs;
	for (int i = 0; i < n; i++) hs[i]->lmRunBegin(runs[i], niter, chi2 ? chi2 + (size_t)i * niter : nullptr);
	std::vector<int> act, inPcg;
	std::vector<char> okv((size_t)n);
	int batchSolves = 0;
	for (;;)
	{
		act.clear();
		for (int i = 0; i < n; i++) if (hs[i]->lmBeforeTrial(runs[i])) act.push_back(i);
		if (act.empty()) break;
		// linearise + Schur and the front half of the reduced solve, each graph on its own stream
		inPcg.clear();
		for (int i : act)
		{
			hs[i]->schur(true);
			ctx[i] = cuba_hip_solver::SolveCtx(); ctx[i].batched = true; ctx[i].deferCoarse = &coarseJobs;
			if (hs[i]->solveBegin(ctx[i])) okv[i] = ctx[i].result; else inPcg.push_back(i);
		}
		lead->launchCoarseJobs(coarseJobs);
		if (!inPcg.empty())
		{
			const int m = (int)inPcg.size();
			int gridSpmvMax = 0, ncMax = 0; size_t ldsMax = 0; int target = 4, budgetMax = 0;
			for (int a = 0; a < m; a++)
			{
				cuba_hip_solver* h = hs[inPcg[a]];
				BatchEntry& e = hTab[a];
				e.g = h->g; e.st = h->st; e.sys = h->sys; e.maxIter = ctx[inPcg[a]].maxIter;
				e.gridSpmv = (h->g.Pf + h->sys.spmv_rows - 1) / h->sys.spmv_rows;
				gridSpmvMax = std::max(gridSpmvMax, e.gridSpmv); ncMax = std::max(ncMax, h->sys.nc); ldsMax = std::max(ldsMax, batch_pcg2_lds_bytes(h->sys));
				target = std::max(target, (ctx[inPcg[a]].predicted + 3) / 4 * 4); budgetMax = std::max(budgetMax, ctx[inPcg[a]].budget);
				if (h != lead) { HIP_TRY(hipEventRecord(evJoin[a], h->stream)); HIP_TRY(hipStreamWaitEvent(bs, evJoin[a], 0)); }
			}
			HIP_TRY(hipMemcpyAsync(lead->d_batchTab.data(), hTab, sizeof(BatchEntry) * m, hipMemcpyHostToDevice, bs));
			const Scalar tol2 = ctx[inPcg[0]].tol2;
			int k0 = 0, looks = 0;
			std::vector<char> conv((size_t)m, 0), broke((size_t)m, 0);
			bool all = false;
			while (k0 < budgetMax && !all)
			{
				int todo = std::max(4, std::min(target, budgetMax) - k0);
				while (todo > 0)
				{
					const int c = todo;
					// chunk-local iteration numbers + the advance / stop test / report launch, as a lone handle enqueues its batches
					for (int k = 0; k < c; k++) launch_pcg_batch_iteration(dTab, m, lead->g, lead->sys, gridSpmvMax, ncMax, ldsMax, k, tol2, bs);
					launch_pcg_batch_advance(dTab, m, c, bs, tol2);
					for (int a = 0; a < m; a++) hs[inPcg[a]]->noteReport();
					k0 += c; todo -= c;
				}
				all = true;
				for (int a = 0; a < m; a++)
				{
					cuba_hip_solver* h = hs[inPcg[a]];
					h->waitReport();
					const volatile int* f = ctx[inPcg[a]].hInts;
					if (f[0] != 0) broke[a] = 1;
					else if (f[2] != 0 || f[1] < std::min(k0, ctx[inPcg[a]].maxIter)) conv[a] = 1;
					if (!broke[a] && !conv[a] && k0 < ctx[inPcg[a]].budget) all = false;
					h->cntPcgLooks++; h->cntHostLooks++;
				}
				target = k0 + ((looks == 0 && k0 <= 96) ? 4 : std::max(8, k0 / 8 / 4 * 4));
				looks++;
			}
			// back to the graphs' own streams
			HIP_TRY(hipEventRecord(evFork, bs));
			for (int a = 0; a < m; a++)
			{
				const int i = inPcg[a];
				cuba_hip_solver* h = hs[i];
				if (h != lead) HIP_TRY(hipStreamWaitEvent(h->stream, evFork, 0));
				cuba_hip_solver::SolveCtx& sc = ctx[i];
				sc.k0 = std::min(k0, std::max(sc.budget, 4)); sc.looks = looks; sc.eagerIters = 0;
				// (a graph alone stops at ITS iteration budget and hands the solve over; in a batch its iterations run on while the others
				// need them -- harmless, the exact solve replaces the iterate -- but convergence counts only within the budget)
				sc.converged = conv[a] != 0 && sc.hInts[1] <= sc.budget;
				if (!sc.converged && !broke[a]) sc.itersDone = std::min((int)sc.hInts[1], sc.budget);
				okv[i] = h->retryWithFp64Inverse(broke[a] ? h->solveBrokeDown(sc) : h->solveEnd(sc));
			}
			batchSolves++;
		}
		for (int i : act) (void)hs[i]->lmAfterSolve(runs[i], okv[i] != 0);
	}
	for (int i = 0; i < n; i++) nDone[i] = hs[i]->lmRunEnd(runs[i]);
	return batchSolves;
}

int cuba_hip_solver::optimize(int niter, double* chi2Out)
{
	lap(nullptr);
	need();
	lap("optimize: structure ready");
	// the default: the decision of every trial on the device, the trial's tail fused into one pass over the edges.  The host loop below
	// serves the profiled run (every stage synchronises), landmark partitions (the sums need the other ranks) and the degenerate graphs.
	if (!profile && partHi < 0 && Pf > 0 && Lf > 0 && E > 0 && trial_tail_parts(g, st) <= d_parts.size())
		return optimizeDeviceDecision(niter, chi2Out);
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
			if (ok) { backSubstitute(); update(); }
			evaluateTrial(lam, ok, &Fhat, &scale);                  // chi2 at the trial estimate + gain-ratio denominator, one host look
			scale += 1e-3;
			rho = ok ? (F - Fhat) / scale : -1;
			if (rho > 0)
			{
				// (t * t * t, not pow: the expressions of lm_decide in ba_edge.hip, so that this loop and the device-resident decision
				// give the same damping bit for bit)
				const double t3 = 2 * rho - 1;
				const double a = 1 - t3 * t3 * t3;
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

void cuba_hip_solver::enqueueEvaluate(double lam, bool withScale)
{
	launch_residual_chi2(g, d_parts.data(), slotsDev, nullptr, stream);
	if (withScale) launch_pose_scale(g, sys, lam, slotsDev + 3 * NSLOT, stream);
	launch_pcg_report(sys, stream); noteReport();       // ticket behind the results (which the kernels wrote into the mapped host block)
}

void cuba_hip_solver::readEvaluate(bool withScale, double* Fhat, double* scale)
{
	waitReport(); cntHostLooks++;
	*Fhat = (double)slot(0);
	*scale = withScale ? (double)slot(NSLOT) + (double)slot(3 * NSLOT) : 0.0;   // landmark part (back_substitute) + pose part
}

void cuba_hip_solver::evaluateTrial(double lam, bool withScale, double* Fhat, double* scale)
{
	StageTimer tm(this, 2);
	enqueueEvaluate(lam, withScale);
	readEvaluate(withScale, Fhat, scale);
}

void cuba_hip_solver::timeKernels(int reps, double* msOut)
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
	if (!sys.upper) launch_hsc_expand(g, st, sys, stream);
	if (sys.agg > 0)
	{
		drainInversion();
		sys.acinv = launch_coarse_setup(g, st, sys, d_coarse[0].data(), d_coarse[1].data(), stream);
		if (fp32Inverse()) launch_coarse_to_fp32(sys.acinv, d_coarse32[0].data(), 6 * sys.cl * sys.nc, stream);
		else launch_coarse_finish(sys.acinv, sys.acinv, 6 * sys.cl * sys.nc, stream);
		coarseValid = false;
		launch_pcg2_fused(g, sys, 0, 0, 1 << 30, -1.0, 0, stream);
	}
	if (sys.upper) msOut[2] = timeit([&] { launch_pcg_upper_iteration(g, st, sys, 0, 1 << 30, -1.0, stream, 1); });
	else msOut[2] = timeit([&] { launch_pcg_spmv(g, st, sys, 0, 1 << 30, -1.0, stream); });
	if (sys.agg > 0)
	{
		// (upper-triangle iteration: row updates + preconditioner, the two launches that stand where the fused kernel does)
		if (sys.upper)
		{
			msOut[3] = timeit([&] { launch_pcg_upper_iteration(g, st, sys, 0, 1 << 30, -1.0, stream, 2); });
			msOut[5] = timeit([&] { launch_pcg_upper_iteration(g, st, sys, 0, 1 << 30, -1.0, stream, 4); });
		}
		else
		{
			msOut[3] = timeit([&] { launch_pcg2_fused(g, sys, 0, 1, 1 << 30, -1.0, 1, stream); });
			msOut[5] = 0;
		}
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

void cuba_hip_solver::chiSquares(double* out, bool wait)
{
	need();
	if (partHi >= 0) d_perEdge.zero(stream);          // (a landmark partition evaluates its own edges only: the others report 0)
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
