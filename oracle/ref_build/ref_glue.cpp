// ref_glue.cpp -- TEST INFRASTRUCTURE ONLY.  C entry point that drives the REFERENCE's own device functions
// (namespace cuba::gpu, /root/reference/src/cuda_block_solver.h:29-94, compiled from the reference sources in
// place by oracle/ref_build/Makefile) through one Levenberg-Marquardt trial, in exactly the order
// CudaBlockSolver uses them (/root/reference/src/cuda_bundle_adjustment.cpp:263-500).  The reduced-system
// solve (cuSOLVER in the reference, not available) is replaced by an increment xp supplied by the caller.
// Used by tests/test_ref_kernels.py to pin BOTH the CPU oracle and our HIP path against reference outputs.
#include <cstdint>
#include <cstring>
#include <vector>

#include "cuda_block_solver.h"   // the reference's header (found through -I/root/reference/src)

using namespace cuba;

// Same-node stage baseline (round-2 verdict item 8): with ref_set_timing(reps > 0) every stage below is run `reps` times between
// two HIP events and its average wall time on the device queue -- the reference's own blocking read-backs included, that is how
// CudaBlockSolver calls these functions -- is left in ref_get_stage_ms().  Stage outputs of a timing run are NOT parity data
// (the accumulating stages have then accumulated `reps` times); scripts/ref_stage_times.py is the only caller.
namespace
{
int g_reps = 0;
double g_stage_ms[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
template <class F> void stage(int idx, F&& fn)
{
	if (g_reps <= 0) { fn(); return; }
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	fn();                                   // warm
	hipDeviceSynchronize();
	hipEventRecord(e0, 0);
	for (int r = 0; r < g_reps; r++) fn();
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0; hipEventElapsedTime(&ms, e0, e1);
	g_stage_ms[idx] = (double)ms / g_reps;
	hipEventDestroy(e0); hipEventDestroy(e1);
}
}  // namespace

extern "C" void ref_set_timing(int reps) { g_reps = reps; }
// [0] computeActiveErrors x2  [1] fillZero + constructQuadraticForm x2  [2] maxDiagonal x2  [3] addLambda x2
// [4] computeBschure  [5] computeHschure  [6] schurComplementPost  [7] computeScale + updatePoses + updateLandmarks
extern "C" void ref_get_stage_ms(double* out8) { for (int i = 0; i < 8; i++) out8[i] = g_stage_ms[i]; }

extern "C" int ref_run_trial(
	int Pt, int Pf, int Lt, int Lf, const double* q, const double* t, const double* cam, const double* Xw,
	int E2, int E3, const int* eP, const int* eL, const double* meas3, const double* omega,
	const int* rk_type, const double* rk_delta,
	int nblk, const int* hscRowPtr, const int* hscColInd,
	double lambda, const double* xp_in,
	// outputs
	double* chi2, double* Hpp, double* bp, double* Hll, double* bl, double* maxdiag,
	double* bsc, double* hsc, double* invHll, double* xl, double* scale,
	double* q_out, double* t_out, double* Xw_out, double* chi2_after, double* chi_per_edge)
{
	const int E = E2 + E3;
	// ---- uploads, laid out like CudaBlockSolver::buildStructure (:319-354) -------------------------------
	GpuVec1d d_solution; d_solution.resize((size_t)Pt * 7 + (size_t)Lt * 3);
	GpuVec4d d_qs; GpuVec3d d_ts, d_Xws;
	d_qs.map(Pt, d_solution.data());
	d_ts.map(Pt, d_qs.data() + Pt);
	d_Xws.map(Lt, d_ts.data() + Pt);
	d_qs.upload((const Vec4d*)q); d_ts.upload((const Vec3d*)t); d_Xws.upload((const Vec3d*)Xw);
	GpuVec5d d_cameras; d_cameras.assign(Pt, cam);

	std::vector<Vec2d> m2(E2); std::vector<Vec3d> m3(E3);
	for (int i = 0; i < E2; i++) { m2[i][0] = meas3[3 * i]; m2[i][1] = meas3[3 * i + 1]; }
	for (int i = 0; i < E3; i++) for (int k = 0; k < 3; k++) m3[i][k] = meas3[3 * (E2 + i) + k];
	std::vector<Vec2i> pl(E); std::vector<uint8_t> flags(E); std::vector<Vec3i> blockpos;
	for (int i = 0; i < E; i++)
	{
		pl[i][0] = eP[i]; pl[i][1] = eL[i];
		const bool fp = eP[i] >= Pf, fl = eL[i] >= Lf;
		flags[i] = (uint8_t)((fp ? EDGE_FLAG_FIXED_P : 0) | (fl ? EDGE_FLAG_FIXED_L : 0));
		if (!fp && !fl) { Vec3i b; b[0] = eP[i]; b[1] = eL[i]; b[2] = i; blockpos.push_back(b); }
	}
	const int nHpl = (int)blockpos.size();
	GpuVec2d d_meas2, d_err2; GpuVec3d d_meas3, d_err3, d_Xcs2, d_Xcs3;
	GpuVec1d d_om2, d_om3; GpuVec2i d_pl2, d_pl3; GpuVec1b d_fl2, d_fl3;
	d_meas2.assign(E2, m2.data()); d_meas3.assign(E3, m3.data());
	d_err2.resize(E2); d_err3.resize(E3); d_Xcs2.resize(E2); d_Xcs3.resize(E3);
	d_om2.assign(E2, omega); d_om3.assign(E3, omega + E2);
	d_pl2.assign(E2, pl.data()); d_pl3.assign(E3, pl.data() + E2);
	d_fl2.assign(E2, flags.data()); d_fl3.assign(E3, flags.data() + E2);
	DeviceBuffer<Scalar> d_chi(1);
	const RobustKernel k2(rk_type[0], rk_delta[0]), k3(rk_type[1], rk_delta[1]);

	GpuVec1d d_x, d_b; d_x.resize((size_t)Pf * 6 + (size_t)Lf * 3); d_b.resize(d_x.size());
	GpuPx1BlockVec d_xp, d_bp, d_bsc, d_HppBak; GpuLx1BlockVec d_xl, d_bl, d_HllBak;
	GpuPxPBlockVec d_Hpp; GpuLxLBlockVec d_Hll, d_invHll; GpuPxLBlockVec d_HplInvHll;
	d_xp.map(Pf, d_x.data()); d_bp.map(Pf, d_b.data());
	d_xl.map(Lf, d_x.data() + (size_t)Pf * 6); d_bl.map(Lf, d_b.data() + (size_t)Pf * 6);
	d_Hpp.resize(Pf); d_Hll.resize(Lf); d_HppBak.resize(Pf); d_HllBak.resize(Lf);
	d_bsc.resize(Pf); d_invHll.resize(Lf); d_HplInvHll.resize(nHpl);

	// Hpl structure on the device (gpu::buildHplStructure) and the Schur product list for the caller's Hsc pattern
	GpuHplBlockMat d_Hpl; d_Hpl.resize(Pf, Lf); d_Hpl.resizeNonZeros(nHpl);
	GpuVec3i d_blockpos; d_blockpos.assign(nHpl, blockpos.data());
	GpuVec1i d_nnzPerCol, d_edge2Hpl, d_e2h2, d_e2h3;
	d_nnzPerCol.resize(Lf + 1); d_edge2Hpl.resize(E);
	gpu::buildHplStructure(d_blockpos, d_Hpl, d_edge2Hpl, d_nnzPerCol);
	d_e2h2.map(E2, d_edge2Hpl.data()); d_e2h3.map(E3, d_edge2Hpl.data() + E2);
	GpuHscBlockMat d_Hsc; d_Hsc.resize(Pf, Pf); d_Hsc.resizeNonZeros(nblk);
	d_Hsc.upload(nullptr, hscRowPtr, hscColInd);
	// number of block products = sum over landmarks of n(n+1)/2 (src/sparse_block_matrix.cpp:69-100)
	std::vector<int> nper(Lf, 0);
	for (const auto& b : blockpos) nper[b[1]]++;
	long long nmul = 0;
	for (int l = 0; l < Lf; l++) nmul += (long long)nper[l] * (nper[l] + 1) / 2;
	GpuVec3i d_mulIds; d_mulIds.resize((size_t)nmul);
	gpu::findHschureMulBlockIndices(d_Hpl, d_Hsc, d_mulIds);

	// ---- one trial, stage by stage ------------------------------------------------------------------
	Scalar c2 = 0, c3 = 0;
	stage(0, [&] {
		c2 = gpu::computeActiveErrors(d_qs, d_ts, d_cameras, d_Xws, d_meas2, d_om2, d_pl2, k2, d_err2, d_Xcs2, d_chi);
		c3 = gpu::computeActiveErrors(d_qs, d_ts, d_cameras, d_Xws, d_meas3, d_om3, d_pl3, k3, d_err3, d_Xcs3, d_chi);
	});
	*chi2 = c2 + c3;
	stage(1, [&] {
		d_Hpp.fillZero(); d_Hll.fillZero(); d_bp.fillZero(); d_bl.fillZero();
		gpu::constructQuadraticForm(d_Xcs2, d_qs, d_cameras, d_err2, d_om2, d_pl2, d_e2h2, d_fl2, k2, d_Hpp, d_bp, d_Hll, d_bl, d_Hpl);
		gpu::constructQuadraticForm(d_Xcs3, d_qs, d_cameras, d_err3, d_om3, d_pl3, d_e2h3, d_fl3, k3, d_Hpp, d_bp, d_Hll, d_bl, d_Hpl);
	});
	cudaMemcpy(Hpp, d_Hpp.values(), sizeof(double) * 36 * Pf, cudaMemcpyDeviceToHost);
	cudaMemcpy(bp, d_bp.values(), sizeof(double) * 6 * Pf, cudaMemcpyDeviceToHost);
	cudaMemcpy(Hll, d_Hll.values(), sizeof(double) * 9 * Lf, cudaMemcpyDeviceToHost);
	cudaMemcpy(bl, d_bl.values(), sizeof(double) * 3 * Lf, cudaMemcpyDeviceToHost);
	{
		DeviceBuffer<Scalar> d_buffer(16);
		Scalar a = 0, b = 0;
		stage(2, [&] { a = gpu::maxDiagonal(d_Hpp, d_buffer); b = gpu::maxDiagonal(d_Hll, d_buffer); });
		*maxdiag = a > b ? a : b;
	}
	if (g_reps > 0)
	{
		// timing only: add + restore so that the diagonals end up damped exactly once
		stage(3, [&] { gpu::addLambda(d_Hpp, lambda, d_HppBak); gpu::addLambda(d_Hll, lambda, d_HllBak); gpu::restoreDiagonal(d_Hpp, d_HppBak); gpu::restoreDiagonal(d_Hll, d_HllBak); });
	}
	gpu::addLambda(d_Hpp, lambda, d_HppBak);
	gpu::addLambda(d_Hll, lambda, d_HllBak);
	stage(4, [&] { gpu::computeBschure(d_bp, d_Hpl, d_Hll, d_bl, d_bsc, d_invHll, d_HplInvHll); });
	stage(5, [&] { gpu::computeHschure(d_Hpp, d_HplInvHll, d_Hpl, d_mulIds, d_Hsc); });
	cudaMemcpy(bsc, d_bsc.values(), sizeof(double) * 6 * Pf, cudaMemcpyDeviceToHost);
	cudaMemcpy(hsc, d_Hsc.values(), sizeof(double) * 36 * nblk, cudaMemcpyDeviceToHost);
	cudaMemcpy(invHll, d_invHll.values(), sizeof(double) * 9 * Lf, cudaMemcpyDeviceToHost);
	cudaMemcpy(d_xp.values(), xp_in, sizeof(double) * 6 * Pf, cudaMemcpyHostToDevice);   // stands in for cuSOLVER
	stage(6, [&] { gpu::schurComplementPost(d_invHll, d_bl, d_Hpl, d_xp, d_xl); });
	cudaMemcpy(xl, d_xl.values(), sizeof(double) * 3 * Lf, cudaMemcpyDeviceToHost);
	stage(7, [&] {
		gpu::computeScale(d_x, d_b, d_chi, lambda);
		d_chi.download(scale);
		// updatePoses covers xp.size() == Pf free poses, updateLandmarks the Lf free landmarks
		gpu::updatePoses(d_xp, d_qs, d_ts);
		gpu::updateLandmarks(d_xl, d_Xws);
	});
	d_qs.download((Vec4d*)q_out); d_ts.download((Vec3d*)t_out); d_Xws.download((Vec3d*)Xw_out);
	const Scalar a2 = gpu::computeActiveErrors(d_qs, d_ts, d_cameras, d_Xws, d_meas2, d_om2, d_pl2, k2, d_err2, d_Xcs2, d_chi);
	const Scalar a3 = gpu::computeActiveErrors(d_qs, d_ts, d_cameras, d_Xws, d_meas3, d_om3, d_pl3, k3, d_err3, d_Xcs3, d_chi);
	*chi2_after = a2 + a3;
	GpuVec1d d_chiSqs, d_cs2, d_cs3; d_chiSqs.resize(E);
	d_cs2.map(E2, d_chiSqs.data()); d_cs3.map(E3, d_chiSqs.data() + E2);
	gpu::computeChiSquares(d_qs, d_ts, d_cameras, d_Xws, d_meas2, d_om2, d_pl2, d_cs2);
	gpu::computeChiSquares(d_qs, d_ts, d_cameras, d_Xws, d_meas3, d_om3, d_pl3, d_cs3);
	d_chiSqs.download(chi_per_edge);
	gpu::waitForKernelCompletion();
	return (int)cudaGetLastError();
}
