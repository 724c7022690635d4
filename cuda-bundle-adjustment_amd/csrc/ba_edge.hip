// ba_edge.hip -- the edge-streaming kernels of one Levenberg-Marquardt trial outside the linearisation: robust chi2
// (computeActiveErrors / computeChiSquares, /root/reference/src/cuda_block_solver.cu:733-786, 841-875), back-substitution
// (schurComplementPost :1029-1043), the gain-ratio denominator (computeScale :1070-1091), the SE3 / R^3 update
// (updatePoses / updateLandmarks :1045-1068) and the fused trial tail that does the last three in one pass over the edges.
// Lane = edge, wave = whole landmarks (edges sorted by landmark); every sum has a fixed order => bit-reproducible.

#include "ba_device.hpp"

namespace cubahip
{

// ---------------------------------------------------------------------------------------------------
// robust chi2 (and optional per-edge non-robust chi2).
// Replaces computeActiveErrorsKernel / computeChiSquaresKernel (cuda_block_solver.cu:733-786, 841-875):
// no errors/Xcs are stored -- later kernels recompute them from 40 B/edge instead of re-reading 48 B/edge.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void residual_chi2_body(const DeviceGraph& g, Scalar* parts, Scalar* per_edge, int bid, int nb)
{
	Scalar acc = 0;
	for (int e = g.e_begin + bid * 256 + threadIdx.x; e < g.e_end; e += nb * 256)
	{
		const int pe = g.e_pose[e];
		const bool stereo = (pe & STEREO_BIT) != 0;
		const int ip = pe & ~STEREO_BIT;
		const int il = g.e_lm[e];
		Scalar q[4], t[3], cam[5], Xw[3], meas[3], r[3], Xc[3];
		load_pose(g, ip, q, t, cam);
#pragma unroll
		for (int i = 0; i < 3; i++) Xw[i] = g.Xw[3 * (size_t)il + i];
		meas[0] = g.e_mu[e]; meas[1] = g.e_mv[e]; meas[2] = g.e_mr[e];
		const Scalar ee = g.e_w[e] * edge_residual(q, t, cam, Xw, meas, stereo, r, Xc);
		const int kind = stereo ? g.rk[1].kind : g.rk[0].kind;
		const Scalar delta = stereo ? g.rk[1].delta : g.rk[0].delta;
		acc += robust_rho(kind, delta, ee);
		if (per_edge) per_edge[e] = ee;
	}
	acc = wave_sum(acc);
	__shared__ Scalar part[4];
	if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x == 0) parts[bid] = part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(256) void residual_chi2_kernel(DeviceGraph g, Scalar* parts, Scalar* per_edge)
{
	residual_chi2_body(g, parts, per_edge, blockIdx.x, gridDim.x);
}

void launch_residual_chi2(const DeviceGraph& g, Scalar* parts, Scalar* slots, Scalar* per_edge, hipStream_t st)
{
	const int n = g.e_end - g.e_begin;
	const int grid = n > 0 ? min((n + 255) / 256, 2048) : 0;
	if (grid > 0) hipLaunchKernelGGL(residual_chi2_kernel, dim3(grid), dim3(256), 0, st, g, parts, per_edge);
	launch_reduce_parts(parts, grid, slots, st);
}

// ---------------------------------------------------------------------------------------------------
// back substitution xl = inv(Hll + lambda I) (bl - sum_e Hpl_e^T xp[pose(e)]) and the landmark part of
// sum x (lambda x + b).  Replaces schurComplementPostKernel (:1029-1043) + half of computeScaleKernel (:1070-1091).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void edge_hplT_x(const LaneEdge& le, const Scalar* xp, Scalar c[3])
{
	// Hpl^T x = JL^T w (JP x)
	const EdgeLin& L = le.lin;
	Scalar v[3];
#pragma unroll
	for (int m = 0; m < 3; m++)
	{
		Scalar s = 0;
#pragma unroll
		for (int r = 0; r < 6; r++) s += L.JP[m][r] * xp[r];
		v[m] = le.wr * s;
	}
#pragma unroll
	for (int k = 0; k < 3; k++) c[k] = L.JL[0][k] * v[0] + L.JL[1][k] * v[1] + L.JL[2][k] * v[2];
}

__device__ __forceinline__ Scalar finish_landmark(const DeviceSystem& sys, int il, const Scalar csum[3], Scalar lambda)
{
	const Scalar* ls = sys.lm_sys + 9 * (size_t)il;
	Scalar inv[6], bl[3], cl[3], xl[3];
#pragma unroll
	for (int k = 0; k < 6; k++) inv[k] = ls[k];
#pragma unroll
	for (int k = 0; k < 3; k++) { bl[k] = ls[6 + k]; cl[k] = bl[k] - csum[k]; }
	Scalar sc = 0;
#pragma unroll
	for (int i = 0; i < 3; i++)
	{
		xl[i] = inv[sym3_idx(i, 0)] * cl[0] + inv[sym3_idx(i, 1)] * cl[1] + inv[sym3_idx(i, 2)] * cl[2];
		sys.xl[3 * (size_t)il + i] = xl[i];
		sc += xl[i] * (lambda * xl[i] + bl[i]);
	}
	return sc;
}

__global__ __launch_bounds__(LIN_BLOCK) void back_substitute_kernel(DeviceGraph g, DeviceStructure st, DeviceSystem sys, Scalar lambda)
{
	__shared__ Scalar lds_all[(LIN_BLOCK / WAVE) * WAVE * 3];
	const int lane = threadIdx.x & 63;
	const int wv = threadIdx.x >> 6;
	const int wave = blockIdx.x * (LIN_BLOCK / WAVE) + wv;
	if (wave >= st.nWaves) return;
	Scalar* lds = lds_all + wv * WAVE * 3;
	const int lm0 = st.wave_lm[2 * wave], lm1 = st.wave_lm[2 * wave + 1];
	const int e0 = g.lm_ptr[lm0], e1 = g.lm_ptr[lm1];
	const int e = e0 + lane;
	const bool valid = e < e1;
	int il = lm0, seg0 = 0, seg1 = 0;
	Scalar c[3] = { 0, 0, 0 };
	if (valid)
	{
		il = g.e_lm[e];
		if (il < g.Lf)
		{
			seg0 = g.lm_ptr[il] - e0;
			seg1 = g.lm_ptr[il + 1] - e0;
			const int ip = g.e_pose[e] & ~STEREO_BIT;
			if (ip < g.Pf)
			{
				LaneEdge le;
				linearize_edge(g, e, le);
				Scalar xp[6];
#pragma unroll
				for (int r = 0; r < 6; r++) xp[r] = sys.xp[6 * (size_t)ip + r];
				edge_hplT_x(le, xp, c);
			}
		}
	}
	const bool lmFree = valid && il < g.Lf;
#pragma unroll
	for (int k = 0; k < 3; k++) lds[lane * 3 + k] = c[k];
	wave_lds_sync();
	Scalar sc = 0;
	if (lmFree && lane == seg0)
	{
		Scalar cs[3] = { 0, 0, 0 };
		for (int j = seg0; j < seg1; j++)
		{
			cs[0] += lds[j * 3 + 0]; cs[1] += lds[j * 3 + 1]; cs[2] += lds[j * 3 + 2];
		}
		sc = finish_landmark(sys, il, cs, lambda);
	}
	sc = wave_sum(sc);
	if (lane == 0) sys.parts[wave] = sc;     // one partial per wave, summed by reduce_parts_kernel
}

__global__ __launch_bounds__(256) void big_back_substitute_kernel(DeviceGraph g, DeviceStructure st, DeviceSystem sys, Scalar lambda)
{
	__shared__ Scalar red[4][3];
	const int il = st.big_lm[blockIdx.x];
	if (il >= g.Lf)
	{
		if (threadIdx.x == 0) sys.parts[st.nWaves + blockIdx.x] = 0;      // (a fixed landmark has no increment, but its partial is summed)
		return;
	}
	const int e0 = g.lm_ptr[il], e1 = g.lm_ptr[il + 1];
	Scalar acc[3] = { 0, 0, 0 };
	for (int e = e0 + threadIdx.x; e < e1; e += 256)
	{
		const int ip = g.e_pose[e] & ~STEREO_BIT;
		if (ip >= g.Pf) continue;
		LaneEdge le;
		linearize_edge(g, e, le);
		Scalar xp[6], c[3];
#pragma unroll
		for (int r = 0; r < 6; r++) xp[r] = sys.xp[6 * (size_t)ip + r];
		edge_hplT_x(le, xp, c);
		acc[0] += c[0]; acc[1] += c[1]; acc[2] += c[2];
	}
#pragma unroll
	for (int k = 0; k < 3; k++) acc[k] = wave_sum(acc[k]);
	if ((threadIdx.x & 63) == 0)
#pragma unroll
		for (int k = 0; k < 3; k++) red[threadIdx.x >> 6][k] = acc[k];
	__syncthreads();
	if (threadIdx.x == 0)
	{
		Scalar cs[3];
#pragma unroll
		for (int k = 0; k < 3; k++) cs[k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
		sys.parts[st.nWaves + blockIdx.x] = finish_landmark(sys, il, cs, lambda);
	}
}

static void launch_back_substitute_kernels(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, Scalar lambda, hipStream_t s)
{
	if (st.nWaves > 0)
	{
		const int grid = (st.nWaves + (LIN_BLOCK / WAVE) - 1) / (LIN_BLOCK / WAVE);
		hipLaunchKernelGGL(back_substitute_kernel, dim3(grid), dim3(LIN_BLOCK), 0, s, g, st, sys, lambda);
	}
	if (st.nBig > 0)
		hipLaunchKernelGGL(big_back_substitute_kernel, dim3(st.nBig), dim3(256), 0, s, g, st, sys, lambda);
}

void launch_back_substitute(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, Scalar lambda, hipStream_t s)
{
	if (g.Lf <= 0) return;
	launch_back_substitute_kernels(g, st, sys, lambda, s);
	launch_reduce_parts(sys.parts, st.nWaves + st.nBig, sys.slots + NSLOT, s);
}

// sum x (lambda x + b), pose part and (stage API only) landmark part.  Ref: computeScaleKernel :1070-1091.
__device__ __forceinline__ void pose_scale_body(const DeviceGraph& g, const DeviceSystem& sys, Scalar lambda, Scalar* parts, int bid, int nb)
{
	Scalar acc = 0;
	for (int i = bid * 256 + threadIdx.x; i < g.Pf * 6; i += nb * 256)
	{
		const Scalar x = sys.xp[i];
		acc += x * (lambda * x + sys.bp[i]);
	}
	acc = wave_sum(acc);
	if ((threadIdx.x & 63) == 0) parts[bid * 4 + (threadIdx.x >> 6)] = acc;
}

__global__ __launch_bounds__(256) void pose_scale_kernel(DeviceGraph g, DeviceSystem sys, Scalar lambda, Scalar* parts)
{
	pose_scale_body(g, sys, lambda, parts, blockIdx.x, gridDim.x);
}

// Second stage of the three sums of a trial (landmark part of the denominator from the back-substitution, chi2, pose part) and
// the report to the host in one launch: each sum is added exactly as reduce_parts_kernel adds it; the results go into the
// mapped host block, the ticket follows them.
// The Levenberg-Marquardt decision of one trial, taken on the device: gain ratio, acceptance, next damping (control flow and arithmetic of
// CudaBundleAdjustmentImpl::optimize, /root/reference/src/cuda_bundle_adjustment.cpp:816-851; no fused multiply-adds: the host loop of the
// stage API and the multi-GPU driver evaluate the same expressions and must get the same bits).  ok = 0: the reduced solve failed, rho = -1.
// `halt` marks what ends a run without the host being able to foresee it (a rejected trial with rho == 0, a damping that is no longer
// finite, the tenth rejection in a row): every later trial the host may already have enqueued is then rejected whatever it computed.
__device__ __forceinline__ void lm_decide(double* st, Scalar* lamOut, double* ring, int ok, double sumLm, double Fhat, double sumPose)
{
#pragma clang fp contract(off)
	const double F = st[0], lam = st[1], nu = st[2];
	const bool halt = st[3] != 0.0;
	const int trial = (int)st[4];
	const double scale = (ok ? sumLm + sumPose : 0.0) + 1e-3;
	const double rho = ok ? (F - Fhat) / scale : -1.0;
	const bool acc = !halt && rho > 0;
	double lamN = lam, nuN = nu, Fn = F, rej = st[6];
	bool haltN = halt;
	if (!halt)
	{
		if (acc)
		{
			const double t = 2 * rho - 1;
			const double a = 1 - t * t * t;
			lamN = lam * fmax(1. / 3, fmin(a, 2. / 3));
			nuN = 2; Fn = Fhat; rej = 0;
		}
		else
		{
			lamN = lam * nu; nuN = nu * 2; rej += 1;
			if (!(rho < 0) && rho <= 0) haltN = true;         // (rho == 0: the reference leaves its trial loop and then its iteration loop)
			if (rej >= st[7]) haltN = true;
			if (!(rho < 0)) rej = 0;                          // (a NaN gain ratio ends the iteration without ending the run: the host loop starts counting afresh)
		}
		if (!(fabs(lamN) <= 1.7e308)) haltN = true;           // (not finite)
	}
	double* rec = ring + (size_t)(trial % LM_RING) * LM_REC;
	// (rec[7]: the record's identity -- run nonce and trial number -- by which the host tells a record that has landed from what the slot held before)
	rec[0] = ok ? Fhat : 0.0; rec[1] = scale; rec[2] = rho; rec[3] = lamN; rec[4] = Fn; rec[5] = acc ? 1.0 : 0.0; rec[6] = haltN ? 1.0 : 0.0; rec[7] = st[8] + (double)(trial + 1);
	st[0] = Fn; st[1] = lamN; st[2] = nuN; st[3] = haltN ? 1.0 : 0.0; st[4] = (double)(trial + 1); st[5] = acc ? 1.0 : 0.0; st[6] = rej;
	lamOut[0] = (Scalar)lamN;
}

__device__ __forceinline__ void reduce_report_body(const DeviceSystem& sys, const Scalar* pA, int nA, Scalar* oA, const Scalar* pB, int nB, Scalar* oB,
	const Scalar* pC, int nC, Scalar* oC, double* lmState, Scalar* lmLam, double* lmRing)
{
	__shared__ Scalar sh[3][16];
	// the three sums side by side: thread shares first, then one barrier for all of them (each is added exactly as
	// reduce_parts_kernel adds it)
	const Scalar vA = wave_sum(parts_thread_sum(pA, nA)), vB = wave_sum(parts_thread_sum(pB, nB)), vC = wave_sum(parts_thread_sum(pC, nC));
	if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = vA; sh[1][threadIdx.x >> 6] = vB; sh[2][threadIdx.x >> 6] = vC; }
	__syncthreads();
	if (threadIdx.x < 64)
	{
		const bool in = threadIdx.x < 16;
		const Scalar tA = wave_sum(in ? sh[0][threadIdx.x] : Scalar(0)), tB = wave_sum(in ? sh[1][threadIdx.x] : Scalar(0)), tC = wave_sum(in ? sh[2][threadIdx.x] : Scalar(0));
		store_slot_group(oA, tA); store_slot_group(oB, tB); store_slot_group(oC, tC);
		if (lmState && threadIdx.x == 0) lm_decide(lmState, lmLam, lmRing, 1, (double)tA, (double)tB, (double)tC);
	}
	__threadfence_system();          // every writer's results before the ticket
	asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // (and landed: see publish_report)
	__syncthreads();
	if (threadIdx.x == 0 && sys.host_flags) publish_report(sys);
}


__global__ __launch_bounds__(1024) void reduce_report_kernel(DeviceSystem sys, const Scalar* pA, int nA, Scalar* oA, const Scalar* pB, int nB, Scalar* oB,
	const Scalar* pC, int nC, Scalar* oC, double* lmState, Scalar* lmLam, double* lmRing)
{
	reduce_report_body(sys, pA, nA, oA, pB, nB, oB, pC, nC, oC, lmState, lmLam, lmRing);
}

__global__ __launch_bounds__(1024) void reduce_report_batch_kernel(const BatchEntry* __restrict__ tab)
{
	const BatchEntry& e = tab[blockIdx.x];
	const BatchTrial& t = e.t;
	if (!t.reportOn) return;
	reduce_report_body(e.sys, t.scParts, t.nA, e.sys.slots + NSLOT, t.chiParts, t.nA, e.sys.slots, t.scaleParts, 4 * t.nScale, e.sys.slots + 3 * NSLOT, t.lmState, t.lmLam, t.lmRing);
}

__global__ __launch_bounds__(256) void landmark_scale_kernel(DeviceGraph g, DeviceSystem sys, Scalar lambda, Scalar* parts)
{
	Scalar acc = 0;
	for (int i = blockIdx.x * 256 + threadIdx.x; i < g.Lf * 3; i += gridDim.x * 256)
	{
		const Scalar x = sys.xl[i];
		acc += x * (lambda * x + sys.lm_sys[9 * (size_t)(i / 3) + 6 + (i % 3)]);
	}
	acc = wave_sum(acc);
	if ((threadIdx.x & 63) == 0) parts[blockIdx.x * 4 + (threadIdx.x >> 6)] = acc;
}

void launch_pose_scale(const DeviceGraph& g, const DeviceSystem& sys, Scalar lambda, Scalar* slots, hipStream_t s)
{
	const int grid = g.Pf > 0 ? min((g.Pf * 6 + 255) / 256, 256) : 0;
	if (grid > 0) hipLaunchKernelGGL(pose_scale_kernel, dim3(grid), dim3(256), 0, s, g, sys, lambda, sys.parts);
	launch_reduce_parts(sys.parts, grid * 4, slots, s);
}

void launch_landmark_scale(const DeviceGraph& g, const DeviceSystem& sys, Scalar lambda, Scalar* slots, hipStream_t s)
{
	const int grid = g.Lf > 0 ? min((g.Lf * 3 + 255) / 256, 1024) : 0;
	if (grid > 0) hipLaunchKernelGGL(landmark_scale_kernel, dim3(grid), dim3(256), 0, s, g, sys, lambda, sys.parts);
	launch_reduce_parts(sys.parts, grid * 4, slots, s);
}

// ---------------------------------------------------------------------------------------------------
// manifold update.  Ref: updatePosesKernel / updateLandmarksKernel :1045-1068.
// ---------------------------------------------------------------------------------------------------
// one launch: the first workgroups take the poses, the rest the landmark coordinates
__global__ __launch_bounds__(256) void update_state_kernel(DeviceGraph g, DeviceSystem sys, int poseBlocks)
{
	if ((int)blockIdx.x < poseBlocks)
	{
		const int i = blockIdx.x * 256 + threadIdx.x;
		if (i >= g.Pf) return;
		Scalar upd[6], q[4], t[3];
#pragma unroll
		for (int k = 0; k < 6; k++) upd[k] = sys.xp[6 * (size_t)i + k];
#pragma unroll
		for (int k = 0; k < 4; k++) q[k] = g.q[4 * (size_t)i + k];
#pragma unroll
		for (int k = 0; k < 3; k++) t[k] = g.t[3 * (size_t)i + k];
		pose_exp_update(upd, q, t);
#pragma unroll
		for (int k = 0; k < 4; k++) g.q[4 * (size_t)i + k] = q[k];
#pragma unroll
		for (int k = 0; k < 3; k++) g.t[3 * (size_t)i + k] = t[k];
		return;
	}
	const int i = (blockIdx.x - poseBlocks) * 256 + threadIdx.x;
	if (i < g.Lf * 3) g.Xw[i] += sys.xl[i];
}

void launch_update_state(const DeviceGraph& g, const DeviceSystem& sys, hipStream_t s)
{
	const int pb = (g.Pf + 255) / 256, lb = (g.Lf * 3 + 255) / 256;
	if (pb + lb > 0) hipLaunchKernelGGL(update_state_kernel, dim3(pb + lb), dim3(256), 0, s, g, sys, pb);
}

// ---------------------------------------------------------------------------------------------------
// Fused tail of an LM trial (optimize() only): back-substitution, update and evaluation of the trial in ONE pass over the edges.
// A landmark's wave computes xl from the PRE-update estimate (read from the state backup the landmark pass of this trial made,
// so the pose-update workgroups of the same launch may overwrite the live state meanwhile), stores xl and Xw + xl, and then every
// lane evaluates its own edge at the updated estimate: its pose is updated in registers by the very code the pose-update
// workgroups run (pose_exp_update on the same inputs), its landmark comes through LDS from the head lane.  Replaces
// schurComplementPostKernel + updatePosesKernel + updateLandmarksKernel + computeActiveErrorsKernel + computeScaleKernel
// (cuda_block_solver.cu:1029-1091, 733-786) for one trial: the edge stream is read twice per trial instead of three times.
// Roles by workgroup index: [0, nLmGroups) landmark waves, then poseBlocks pose-update workgroups, then nScale workgroups for the
// pose part of the gain-ratio denominator.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ Scalar finish_landmark_x(const DeviceSystem& sys, int il, const Scalar csum[3], Scalar lambda, Scalar xl[3])
{
	const Scalar* ls = sys.lm_sys + 9 * (size_t)il;
	Scalar inv[6], bl[3], cl[3];
#pragma unroll
	for (int k = 0; k < 6; k++) inv[k] = ls[k];
#pragma unroll
	for (int k = 0; k < 3; k++) { bl[k] = ls[6 + k]; cl[k] = bl[k] - csum[k]; }
	Scalar sc = 0;
#pragma unroll
	for (int i = 0; i < 3; i++)
	{
		xl[i] = inv[sym3_idx(i, 0)] * cl[0] + inv[sym3_idx(i, 1)] * cl[1] + inv[sym3_idx(i, 2)] * cl[2];
		sys.xl[3 * (size_t)il + i] = xl[i];
		sc += xl[i] * (lambda * xl[i] + bl[i]);
	}
	return sc;
}

// robust chi2 term of an edge at the UPDATED estimate: (q0, t0) is the pre-update pose, upd its increment (poseFree = false: fixed pose)
__device__ __forceinline__ Scalar updated_edge_rho(const DeviceGraph& g, const Scalar q0[4], const Scalar t0[3], const Scalar cam[5], const Scalar upd[6], bool poseFree,
	const Scalar Xn[3], const Scalar meas[3], Scalar w, bool stereo)
{
	Scalar q[4] = { q0[0], q0[1], q0[2], q0[3] }, t[3] = { t0[0], t0[1], t0[2] }, r[3], Xc[3];
	if (poseFree) pose_exp_update(upd, q, t);
	const Scalar ee = w * edge_residual(q, t, cam, Xn, meas, stereo, r, Xc);
	return robust_rho(stereo ? g.rk[1].kind : g.rk[0].kind, stereo ? g.rk[1].delta : g.rk[0].delta, ee);
}

__device__ __forceinline__ void update_pose_rows(const DeviceGraph& g, const DeviceSystem& sys, const Scalar* __restrict__ old, int i)
{
	Scalar upd[6], q[4], t[3];
#pragma unroll
	for (int k = 0; k < 6; k++) upd[k] = sys.xp[6 * (size_t)i + k];
#pragma unroll
	for (int k = 0; k < 4; k++) q[k] = old[4 * (size_t)i + k];
#pragma unroll
	for (int k = 0; k < 3; k++) t[k] = old[4 * (size_t)g.Pt + 3 * (size_t)i + k];
	pose_exp_update(upd, q, t);
#pragma unroll
	for (int k = 0; k < 4; k++) g.q[4 * (size_t)i + k] = q[k];
#pragma unroll
	for (int k = 0; k < 3; k++) g.t[3 * (size_t)i + k] = t[k];
}

__device__ __forceinline__ void trial_tail_body(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, Scalar lambda,
	const Scalar* __restrict__ old, Scalar* __restrict__ scParts, Scalar* __restrict__ chiParts, int nLmGroups, int poseBlocks, Scalar* __restrict__ scaleParts, int nScale)
{
	lambda = launch_lambda(sys, lambda);
	if ((int)blockIdx.x >= nLmGroups)
	{
		const int b = blockIdx.x - nLmGroups;
		if (b < poseBlocks)
		{
			const int i = b * LIN_BLOCK + threadIdx.x;
			if (i < g.Pf) update_pose_rows(g, sys, old, i);
		}
		else pose_scale_body(g, sys, lambda, scaleParts, b - poseBlocks, nScale);
		return;
	}
	__shared__ Scalar lds_all[(LIN_BLOCK / WAVE) * WAVE * 3];
	__shared__ Scalar wpart[2][LIN_BLOCK / WAVE];
	const int lane = threadIdx.x & 63;
	const int wv = threadIdx.x >> 6;
	const int wave = blockIdx.x * (LIN_BLOCK / WAVE) + wv;
	const bool waveOn = wave < st.nWaves;
	Scalar* lds = lds_all + wv * WAVE * 3;
	const Scalar* qo = old; const Scalar* to = old + 4 * (size_t)g.Pt; const Scalar* Xo = old + 7 * (size_t)g.Pt;
	const int lm0 = waveOn ? st.wave_lm[2 * wave] : 0, lm1 = waveOn ? st.wave_lm[2 * wave + 1] : 0;
	const int e0 = waveOn ? g.lm_ptr[lm0] : 0, e1 = waveOn ? g.lm_ptr[lm1] : 0;
	const int e = e0 + lane;
	const bool valid = waveOn && e < e1;
	int il = lm0, ip = 0, seg0 = 0, seg1 = 0;
	bool stereo = false, poseFree = false;
	Scalar q[4] = { 0, 0, 0, 1 }, t[3] = { 0, 0, 0 }, cam[5] = { 1, 1, 0, 0, 0 }, Xw[3] = { 0, 0, 1 }, meas[3] = { 0, 0, 0 }, xp[6] = { 0, 0, 0, 0, 0, 0 }, w = 0;
	Scalar c[3] = { 0, 0, 0 };
	if (valid)
	{
		const int pe = g.e_pose[e];
		stereo = (pe & STEREO_BIT) != 0;
		ip = pe & ~STEREO_BIT;
		il = g.e_lm[e];
		poseFree = ip < g.Pf;
#pragma unroll
		for (int i = 0; i < 4; i++) q[i] = qo[4 * (size_t)ip + i];
#pragma unroll
		for (int i = 0; i < 3; i++) t[i] = to[3 * (size_t)ip + i];
#pragma unroll
		for (int i = 0; i < 5; i++) cam[i] = g.cam[5 * (size_t)ip + i];
#pragma unroll
		for (int i = 0; i < 3; i++) Xw[i] = Xo[3 * (size_t)il + i];
		meas[0] = g.e_mu[e]; meas[1] = g.e_mv[e]; meas[2] = g.e_mr[e];
		w = g.e_w[e];
		if (poseFree)
		{
#pragma unroll
			for (int r = 0; r < 6; r++) xp[r] = sys.xp[6 * (size_t)ip + r];
		}
		if (il < g.Lf)
		{
			seg0 = g.lm_ptr[il] - e0;
			seg1 = g.lm_ptr[il + 1] - e0;
			if (poseFree)
			{
				// Hpl^T xp at the linearisation point (the pre-update estimate), exactly as back_substitute_kernel forms it
				LaneEdge le;
				Scalar Xc[3];
				const Scalar ss = edge_residual(q, t, cam, Xw, meas, stereo, le.lin.r, Xc);
				le.wr = w * robust_weight(stereo ? g.rk[1].kind : g.rk[0].kind, stereo ? g.rk[1].delta : g.rk[0].delta, w * ss);
				const Rot3 R = quat_to_rot(q[0], q[1], q[2], q[3]);
				edge_jacobians(Xc, R, cam, stereo, le.lin);
				edge_hplT_x(le, xp, c);
			}
		}
	}
	const bool lmFree = valid && il < g.Lf;
#pragma unroll
	for (int k = 0; k < 3; k++) lds[lane * 3 + k] = c[k];
	wave_lds_sync();
	Scalar sc = 0, xl[3] = { 0, 0, 0 };
	if (lmFree && lane == seg0)
	{
		Scalar cs[3] = { 0, 0, 0 };
		for (int j = seg0; j < seg1; j++)
		{
			cs[0] += lds[j * 3 + 0]; cs[1] += lds[j * 3 + 1]; cs[2] += lds[j * 3 + 2];
		}
		sc = finish_landmark_x(sys, il, cs, lambda, xl);
#pragma unroll
		for (int k = 0; k < 3; k++) g.Xw[3 * (size_t)il + k] = Xw[k] + xl[k];     // (= update_landmarks: Xw += xl)
	}
	wave_lds_sync();                     // every segment sum has been read: the head lanes may reuse their own LDS slots
	if (lmFree && lane == seg0)
	{
#pragma unroll
		for (int k = 0; k < 3; k++) lds[lane * 3 + k] = xl[k];
	}
	wave_lds_sync();
	Scalar rho = 0;
	if (valid)
	{
		Scalar Xn[3] = { Xw[0], Xw[1], Xw[2] };
		if (lmFree)
		{
#pragma unroll
			for (int k = 0; k < 3; k++) Xn[k] = Xw[k] + lds[seg0 * 3 + k];
		}
		rho = updated_edge_rho(g, q, t, cam, xp, poseFree, Xn, meas, w, stereo);
	}
	sc = wave_sum(sc); rho = wave_sum(rho);
	if (lane == 0) { wpart[0][wv] = sc; wpart[1][wv] = rho; }
	__syncthreads();
	if (threadIdx.x == 0)
	{
		scParts[blockIdx.x] = (wpart[0][0] + wpart[0][1]) + (wpart[0][2] + wpart[0][3]);
		chiParts[blockIdx.x] = (wpart[1][0] + wpart[1][1]) + (wpart[1][2] + wpart[1][3]);
	}
}


__global__ __launch_bounds__(LIN_BLOCK) void trial_tail_kernel(DeviceGraph g, DeviceStructure st, DeviceSystem sys, Scalar lambda,
	const Scalar* __restrict__ old, Scalar* __restrict__ scParts, Scalar* __restrict__ chiParts, int nLmGroups, int poseBlocks, Scalar* __restrict__ scaleParts, int nScale)
{
	trial_tail_body(g, st, sys, lambda, old, scParts, chiParts, nLmGroups, poseBlocks, scaleParts, nScale);
}

__global__ __launch_bounds__(LIN_BLOCK) void trial_tail_batch_kernel(const BatchEntry* __restrict__ tab)
{
	const BatchEntry& e = tab[blockIdx.y];
	const BatchTrial& t = e.t;
	if (blockIdx.x >= t.tailGrid) return;
	trial_tail_body(e.g, e.st, e.sys, Scalar(-1), t.old, t.scParts, t.chiParts, t.nLm, t.poseBlocks, t.scaleParts, t.nScale);
}

// landmarks with more than 64 observations: one workgroup each (free or fixed: their edges are evaluated either way)
__global__ __launch_bounds__(256) void big_trial_tail_kernel(DeviceGraph g, DeviceStructure st, DeviceSystem sys, Scalar lambda,
	const Scalar* __restrict__ old, Scalar* __restrict__ scParts, Scalar* __restrict__ chiParts)
{
	__shared__ Scalar red[4][3];
	__shared__ Scalar xsh[3];
	__shared__ Scalar rsh[4];
	lambda = launch_lambda(sys, lambda);
	DeviceGraph go = g;                   // the pre-update estimate
	go.q = const_cast<Scalar*>(old); go.t = go.q + 4 * (size_t)g.Pt; go.Xw = go.q + 7 * (size_t)g.Pt;
	const int il = st.big_lm[blockIdx.x];
	const int e0 = g.lm_ptr[il], e1 = g.lm_ptr[il + 1];
	const bool lmFree = il < g.Lf;
	Scalar acc[3] = { 0, 0, 0 };
	if (lmFree)
	{
		for (int e = e0 + threadIdx.x; e < e1; e += 256)
		{
			const int ip = g.e_pose[e] & ~STEREO_BIT;
			if (ip >= g.Pf) continue;
			LaneEdge le;
			linearize_edge(go, e, le);
			Scalar xp[6], c[3];
#pragma unroll
			for (int r = 0; r < 6; r++) xp[r] = sys.xp[6 * (size_t)ip + r];
			edge_hplT_x(le, xp, c);
			acc[0] += c[0]; acc[1] += c[1]; acc[2] += c[2];
		}
	}
#pragma unroll
	for (int k = 0; k < 3; k++) acc[k] = wave_sum(acc[k]);
	if ((threadIdx.x & 63) == 0)
#pragma unroll
		for (int k = 0; k < 3; k++) red[threadIdx.x >> 6][k] = acc[k];
	__syncthreads();
	if (threadIdx.x == 0)
	{
		Scalar xl[3] = { 0, 0, 0 }, sc = 0;
		if (lmFree)
		{
			Scalar cs[3];
#pragma unroll
			for (int k = 0; k < 3; k++) cs[k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
			sc = finish_landmark_x(sys, il, cs, lambda, xl);
#pragma unroll
			for (int k = 0; k < 3; k++) g.Xw[3 * (size_t)il + k] = go.Xw[3 * (size_t)il + k] + xl[k];
		}
		scParts[blockIdx.x] = sc;
#pragma unroll
		for (int k = 0; k < 3; k++) xsh[k] = xl[k];
	}
	__syncthreads();
	Scalar Xn[3];
#pragma unroll
	for (int k = 0; k < 3; k++) Xn[k] = go.Xw[3 * (size_t)il + k] + xsh[k];
	Scalar rho = 0;
	for (int e = e0 + threadIdx.x; e < e1; e += 256)
	{
		const int pe = g.e_pose[e];
		const bool stereo = (pe & STEREO_BIT) != 0;
		const int ip = pe & ~STEREO_BIT;
		Scalar q[4], t[3], cam[5], meas[3], xp[6];
		load_pose(go, ip, q, t, cam);
		meas[0] = g.e_mu[e]; meas[1] = g.e_mv[e]; meas[2] = g.e_mr[e];
		const bool poseFree = ip < g.Pf;
#pragma unroll
		for (int r = 0; r < 6; r++) xp[r] = poseFree ? sys.xp[6 * (size_t)ip + r] : Scalar(0);
		rho += updated_edge_rho(g, q, t, cam, xp, poseFree, Xn, meas, g.e_w[e], stereo);
	}
	rho = wave_sum(rho);
	if ((threadIdx.x & 63) == 0) rsh[threadIdx.x >> 6] = rho;
	__syncthreads();
	if (threadIdx.x == 0) chiParts[blockIdx.x] = (rsh[0] + rsh[1]) + (rsh[2] + rsh[3]);
}

size_t trial_tail_parts(const DeviceGraph& g, const DeviceStructure& st)
{
	const size_t nA = ((size_t)(st.nWaves + LIN_BLOCK / WAVE - 1) / (LIN_BLOCK / WAVE) + st.nBig + 63) / 64 * 64;
	return 2 * nA + 4 * 256 + 64;
}

void launch_trial_tail_fused(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, Scalar lambda, const Scalar* old, hipStream_t s,
	const LmDevice* decide)
{
	const int nLm = (st.nWaves + LIN_BLOCK / WAVE - 1) / (LIN_BLOCK / WAVE);
	const int nA = nLm + st.nBig;
	Scalar* scParts = sys.parts;
	Scalar* chiParts = sys.parts + (size_t)(nA + 63) / 64 * 64;
	Scalar* scaleParts = chiParts + (size_t)(nA + 63) / 64 * 64;
	const int poseBlocks = (g.Pf + LIN_BLOCK - 1) / LIN_BLOCK;
	const int nScale = g.Pf > 0 ? min((g.Pf * 6 + 255) / 256, 256) : 0;
	if (nLm + poseBlocks + nScale > 0)
		hipLaunchKernelGGL(trial_tail_kernel, dim3(nLm + poseBlocks + nScale), dim3(LIN_BLOCK), 0, s, g, st, sys, lambda, old, scParts, chiParts, nLm, poseBlocks, scaleParts, nScale);
	if (st.nBig > 0) hipLaunchKernelGGL(big_trial_tail_kernel, dim3(st.nBig), dim3(256), 0, s, g, st, sys, lambda, old, scParts + nLm, chiParts + nLm);
	hipLaunchKernelGGL(reduce_report_kernel, dim3(1), dim3(1024), 0, s, sys, scParts, nA, sys.slots + NSLOT, chiParts, nA, sys.slots, scaleParts, 4 * nScale, sys.slots + 3 * NSLOT,
		decide ? decide->state : (double*)nullptr, decide ? decide->lam : (Scalar*)nullptr, decide ? decide->ring : (double*)nullptr);
}

// launch_trial_tail_fused + launch_restore_if_rejected of one graph as entries of the batched launches (no landmark with more than 64 observations)
void batch_fill_tail(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, const Scalar* old, const LmDevice& lm, Scalar* state, size_t stateCount, BatchTrial& t)
{
	const int nLm = (st.nWaves + LIN_BLOCK / WAVE - 1) / (LIN_BLOCK / WAVE);
	const int nA = nLm + st.nBig;
	t.scParts = sys.parts;
	t.chiParts = sys.parts + (size_t)(nA + 63) / 64 * 64;
	t.scaleParts = t.chiParts + (size_t)(nA + 63) / 64 * 64;
	t.poseBlocks = (g.Pf + LIN_BLOCK - 1) / LIN_BLOCK;
	t.nScale = g.Pf > 0 ? min((g.Pf * 6 + 255) / 256, 256) : 0;
	t.nLm = nLm; t.nA = nA; t.old = old;
	t.tailGrid = (unsigned)(nLm + t.poseBlocks + t.nScale);
	t.lmState = lm.state; t.lmLam = lm.lam; t.lmRing = lm.ring; t.reportOn = 1;
	t.state = state; t.stateCount = stateCount;
	t.restoreGrid = (unsigned)std::min<size_t>(512, (stateCount + 255) / 256);
}

// decision of a trial whose reduced solve failed + the report, one thread
__global__ void lm_decide_failed_kernel(DeviceSystem sys, double* lmState, Scalar* lmLam, double* lmRing)
{
	if (threadIdx.x != 0) return;
	lm_decide(lmState, lmLam, lmRing, 0, 0.0, 0.0, 0.0);
	__threadfence_system();
	if (sys.host_flags) publish_report(sys);
}

void launch_lm_decide_failed(const DeviceSystem& sys, const LmDevice& lm, hipStream_t s)
{
	hipLaunchKernelGGL(lm_decide_failed_kernel, dim3(1), dim3(64), 0, s, sys, lm.state, lm.lam, lm.ring);
}

// the reference's pop() when -- and only when -- the decision before it was a rejection
__global__ __launch_bounds__(256) void restore_if_rejected_batch_kernel(const BatchEntry* __restrict__ tab)
{
	const BatchTrial& t = tab[blockIdx.y].t;
	if (blockIdx.x >= t.restoreGrid || t.lmState[5] != 0.0) return;
	const size_t stride = (size_t)t.restoreGrid * 256;
	const Scalar* __restrict__ backup = t.backupDst;
	Scalar* __restrict__ state = t.state;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < t.stateCount; i += stride) state[i] = backup[i];
}

__global__ __launch_bounds__(256) void restore_if_rejected_kernel(Scalar* __restrict__ state, const Scalar* __restrict__ backup, size_t count, const double* lmState)
{
	if (lmState[5] != 0.0) return;
	const size_t stride = (size_t)gridDim.x * 256;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) state[i] = backup[i];
}

void launch_restore_if_rejected(Scalar* state, const Scalar* backup, size_t count, const LmDevice& lm, hipStream_t s)
{
	const unsigned grid = (unsigned)std::min<size_t>(512, (count + 255) / 256);
	if (grid) hipLaunchKernelGGL(restore_if_rejected_kernel, dim3(grid), dim3(256), 0, s, state, backup, count, lm.state);
}

void launch_batch_tail(const BatchEntry* tab, int n, unsigned tailGridMax, unsigned restoreGridMax, hipStream_t s)
{
	if (tailGridMax) hipLaunchKernelGGL(trial_tail_batch_kernel, dim3(tailGridMax, n), dim3(LIN_BLOCK), 0, s, tab);
	hipLaunchKernelGGL(reduce_report_batch_kernel, dim3(n), dim3(1024), 0, s, tab);
	if (restoreGridMax) hipLaunchKernelGGL(restore_if_rejected_batch_kernel, dim3(restoreGridMax, n), dim3(256), 0, s, tab);
}

}  // namespace cubahip
