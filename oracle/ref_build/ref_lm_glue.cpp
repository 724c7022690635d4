// ref_lm_glue.cpp -- TEST INFRASTRUCTURE ONLY (oracle/_ref).  C entry point that runs the REFERENCE's own optimiser end to
// end: cuba::CudaBundleAdjustment::create() -> addPoseVertex / addLandmarkVertex / add*Edge -> setRobustKernels ->
// initialize() -> optimize(n), i.e. CudaBundleAdjustmentImpl + CudaBlockSolver + the 21 kernels of
// /root/reference/src (cuda_bundle_adjustment.cpp, sparse_block_matrix.cpp, cuda_block_solver.cu compiled where they lie),
// with only the closed-source cuSOLVER step replaced (ref_linear_solver.cpp).  tests/test_ref_lm.py requires the CPU
// oracle's and the HIP path's LM TRAJECTORIES (chi2 per iteration, final estimates, per-edge chi2) to follow it -- the
// LM-level pin "from the reference itself" next to the stage-level pin of ref_glue.cpp.
#include <chrono>
#include <cstdint>
#include <memory>
#include <vector>

#include "cuda_bundle_adjustment.h"     // the reference's public header (-I/root/reference/include)

// the reference's own stage timers (CudaBundleAdjustment::timeProfile, keys "0: Initialize Optimizer" .. "7: Update Solution", seconds) of
// the LAST initialize() + optimize() of the last ref_lm_run, then the host wall of that initialize() and of that optimize()
static double g_lastProfile[10];
extern "C" void ref_lm_last_profile(double out[10]) { for (int i = 0; i < 10; i++) out[i] = g_lastProfile[i]; }

extern "C" int ref_lm_run(
	int P, const int* pose_id, const uint8_t* pose_fixed, const double* q, const double* t, const double* cam5,
	int L, const int* lm_id, const uint8_t* lm_fixed, const double* Xw,
	int E2, const int* m_vp, const int* m_vl, const double* m_meas, const double* m_info,
	int E3, const int* s_vp, const int* s_vl, const double* s_meas, const double* s_info,
	const int* rk_type, const double* rk_delta, int niterations, int nruns,
	double* chi2_out, int* n_done, double* q_out, double* t_out, double* Xw_out, double* chi_mono, double* chi_stereo)
{
	using namespace cuba;
	std::vector<std::unique_ptr<PoseVertex>> poses;
	std::vector<std::unique_ptr<LandmarkVertex>> lms;
	std::vector<std::unique_ptr<MonoEdge>> mono;
	std::vector<std::unique_ptr<StereoEdge>> stereo;
	auto ba = CudaBundleAdjustment::create();
	for (int i = 0; i < P; i++)
	{
		CameraParams c; c.fx = cam5[5 * i]; c.fy = cam5[5 * i + 1]; c.cx = cam5[5 * i + 2]; c.cy = cam5[5 * i + 3]; c.bf = cam5[5 * i + 4];
		poses.push_back(std::make_unique<PoseVertex>(pose_id[i], Eigen::Quaterniond(q + 4 * i), Array<double, 3>(t + 3 * i), c, pose_fixed[i] != 0));
		ba->addPoseVertex(poses.back().get());
	}
	for (int i = 0; i < L; i++)
	{
		lms.push_back(std::make_unique<LandmarkVertex>(lm_id[i], Array<double, 3>(Xw + 3 * i), lm_fixed[i] != 0));
		ba->addLandmarkVertex(lms.back().get());
	}
	for (int i = 0; i < E2; i++)
	{
		mono.push_back(std::make_unique<MonoEdge>(Array<double, 2>(m_meas + 2 * i), m_info[i], ba->poseVertex(m_vp[i]), ba->landmarkVertex(m_vl[i])));
		ba->addMonocularEdge(mono.back().get());
	}
	for (int i = 0; i < E3; i++)
	{
		stereo.push_back(std::make_unique<StereoEdge>(Array<double, 3>(s_meas + 3 * i), s_info[i], ba->poseVertex(s_vp[i]), ba->landmarkVertex(s_vl[i])));
		ba->addStereoEdge(stereo.back().get());
	}
	ba->setRobustKernels((RobustKernelType)rk_type[0], rk_delta[0], EdgeType::MONOCULAR);
	ba->setRobustKernels((RobustKernelType)rk_type[1], rk_delta[1], EdgeType::STEREO);
	// nruns > 1: the samples' protocol -- initialize() + optimize() again from the estimates the previous run wrote back
	for (int run = 0; run < nruns; run++)
	{
		const auto t0 = std::chrono::steady_clock::now();
		ba->initialize();
		const auto t1 = std::chrono::steady_clock::now();
		ba->optimize(niterations);
		const auto t2 = std::chrono::steady_clock::now();
		g_lastProfile[8] = std::chrono::duration<double>(t1 - t0).count();
		g_lastProfile[9] = std::chrono::duration<double>(t2 - t1).count();
	}
	{
		int k = 0;
		for (const auto& item : ba->timeProfile()) if (k < 8) g_lastProfile[k++] = item.second;      // (a std::map: keys sort by their leading digit)
	}
	const auto& stats = ba->batchStatistics();
	*n_done = (int)stats.size();
	for (size_t i = 0; i < stats.size(); i++) chi2_out[i] = stats[i].chi2;
	for (int i = 0; i < P; i++)
	{
		for (int k = 0; k < 4; k++) q_out[4 * i + k] = poses[i]->q.coeffs()[k];
		for (int k = 0; k < 3; k++) t_out[3 * i + k] = poses[i]->t[k];
	}
	for (int i = 0; i < L; i++) for (int k = 0; k < 3; k++) Xw_out[3 * i + k] = lms[i]->Xw[k];
	for (int i = 0; i < E2; i++) chi_mono[i] = ba->chiSquared(mono[i].get());
	for (int i = 0; i < E3; i++) chi_stereo[i] = ba->chiSquared(stereo[i].get());
	return 0;
}
