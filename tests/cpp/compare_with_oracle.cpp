// compare_with_oracle.cpp -- TEST INFRASTRUCTURE (links oracle/libba_oracle.so; lives under tests/ for that reason).
//
// Counterpart of the reference's samples/sample_comparison_with_g2o.cpp (protocol :67-79 and :303-307, output
// format :81-136) with the CPU oracle in g2o's seat: the same graph goes into an exact-solve CPU Levenberg-
// Marquardt (oracle/ba_oracle.cpp) and into the library under test through the cuba::CudaBundleAdjustment API;
// both are warmed up with one iteration, then initialize + optimize(10) is timed on each side, and the chi2 table
// and the RMSE of the final estimates are printed in the README's format (README.md:161-192 of the reference).
// g2o itself is not installable in this image; a build with g2o would replace class CpuSide below and nothing else.
//
//   usage: compare_with_oracle graph.json [iterations=10] [chi2_rel_tol=1e-6] [rot_tol=1e-8] [trans_tol=1e-6] [lm_tol=1e-6]
//   exit code 0 = every printed difference is inside its tolerance, 2 = a tolerance is exceeded, 1 = usage / IO error
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <vector>

#include <opencv2/core.hpp>   // mini_opencv stand-in (JSON reader) unless real OpenCV is on the include path

#include <cuda_bundle_adjustment.h>

extern "C" {   // oracle/ba_oracle.cpp
void* orc_create(int Pt, int Pf, int Lt, int Lf, const double* q, const double* t, const double* cam, const double* Xw,
	int E, const int* eP, const int* eL, const uint8_t* eDim, const double* meas3, const double* omega);
void orc_destroy(void* h);
void orc_set_robust_kernel(void* h, int edgeType, int kind, double delta);
int orc_optimize(void* h, int niter, double* chi2, double* lambdas, int* trials);
void orc_get_state(void* h, double* q, double* t, double* Xw);
}

namespace
{
struct Store
{
	std::vector<std::unique_ptr<cuba::PoseVertex>> poses;
	std::vector<std::unique_ptr<cuba::LandmarkVertex>> landmarks;
	std::vector<std::unique_ptr<cuba::MonoEdge>> mono;
	std::vector<std::unique_ptr<cuba::StereoEdge>> stereo;
};

template <int N>
cuba::Array<double, N> readVec(const cv::FileNode& node)
{
	cuba::Array<double, N> a;
	int k = 0;
	for (const auto& v : node) { if (k >= N) break; a[k++] = double(v); }
	return a;
}

// The CPU side: its own copy of the estimates in solver order (the indices the library assigned in initialize():
// free vertices first, vertices without edges left out, edges with two fixed ends dropped --
// src/cuda_bundle_adjustment.cpp:142-243), optimised by the oracle.
class CpuSide
{
public:
	CpuSide(const Store& s, double deltaMono, double deltaStereo) : dm_(deltaMono), ds_(deltaStereo)
	{
		for (const auto& v : s.poses) if (v->iP >= 0) { Pt_++; Pf_ += !v->fixed; }
		for (const auto& v : s.landmarks) if (v->iL >= 0) { Lt_++; Lf_ += !v->fixed; }
		q_.resize(4 * (size_t)Pt_); t_.resize(3 * (size_t)Pt_); cam_.resize(5 * (size_t)Pt_); X_.resize(3 * (size_t)Lt_);
		for (const auto& v : s.poses)
		{
			if (v->iP < 0) continue;
			const double* c = v->q.coeffs().data();          // x, y, z, w
			for (int k = 0; k < 4; k++) q_[4 * (size_t)v->iP + k] = c[k];
			for (int k = 0; k < 3; k++) t_[3 * (size_t)v->iP + k] = v->t[k];
			const double cam[5] = { v->camera.fx, v->camera.fy, v->camera.cx, v->camera.cy, v->camera.bf };
			for (int k = 0; k < 5; k++) cam_[5 * (size_t)v->iP + k] = cam[k];
		}
		for (const auto& v : s.landmarks)
			if (v->iL >= 0) for (int k = 0; k < 3; k++) X_[3 * (size_t)v->iL + k] = v->Xw[k];
		auto addEdge = [&](const cuba::PoseVertex* vp, const cuba::LandmarkVertex* vl, int dim, const double* m, double info) {
			if (vp->fixed && vl->fixed) return;
			eP_.push_back(vp->iP); eL_.push_back(vl->iL); eDim_.push_back((uint8_t)dim);
			for (int k = 0; k < 3; k++) meas_.push_back(k < dim ? m[k] : 0.0);
			omega_.push_back(info);
		};
		for (const auto& e : s.mono) addEdge(e->vertexP, e->vertexL, 2, e->measurement.data(), e->information);
		for (const auto& e : s.stereo) addEdge(e->vertexP, e->vertexL, 3, e->measurement.data(), e->information);
	}

	// "initializeOptimization() + optimize(n)": a fresh solver over the current estimates, like the library's initialize()
	std::vector<double> optimize(int n)
	{
		void* h = orc_create(Pt_, Pf_, Lt_, Lf_, q_.data(), t_.data(), cam_.data(), X_.data(), (int)eP_.size(), eP_.data(), eL_.data(),
			eDim_.data(), meas_.data(), omega_.data());
		orc_set_robust_kernel(h, 0, 1, dm_);
		orc_set_robust_kernel(h, 1, 1, ds_);
		std::vector<double> chi2(n), lam(n); std::vector<int> trials(n);
		const int done = orc_optimize(h, n, chi2.data(), lam.data(), trials.data());
		chi2.resize(done);
		orc_get_state(h, q_.data(), t_.data(), X_.data());
		orc_destroy(h);
		return chi2;
	}

	const double* q(int iP) const { return &q_[4 * (size_t)iP]; }
	const double* t(int iP) const { return &t_[3 * (size_t)iP]; }
	const double* X(int iL) const { return &X_[3 * (size_t)iL]; }

private:
	double dm_, ds_;
	int Pt_ = 0, Pf_ = 0, Lt_ = 0, Lf_ = 0;
	std::vector<double> q_, t_, cam_, X_, meas_, omega_;
	std::vector<int> eP_, eL_;
	std::vector<uint8_t> eDim_;
};
}  // namespace

int main(int argc, char** argv)
{
	if (argc < 2) { std::printf("usage: %s graph.json [iterations=10] [chi2_rel_tol] [rot_tol] [trans_tol] [landmark_tol]\n", argv[0]); return 1; }
	const int iterations = argc > 2 ? std::atoi(argv[2]) : 10;
	const double tolChi = argc > 3 ? std::atof(argv[3]) : 1e-6, tolR = argc > 4 ? std::atof(argv[4]) : 1e-8;
	const double tolT = argc > 5 ? std::atof(argv[5]) : 1e-6, tolL = argc > 6 ? std::atof(argv[6]) : 1e-6;
	const double deltaMono = std::sqrt(5.991), deltaStereo = std::sqrt(7.815);     // ref :195-200

	std::cout << "Reading Graph... " << std::flush;
	cv::FileStorage fs(argv[1], cv::FileStorage::READ);
	if (!fs.isOpened()) { std::fprintf(stderr, "cannot open %s\n", argv[1]); return 1; }
	cuba::CameraParams cam;
	cam.fx = fs["fx"]; cam.fy = fs["fy"]; cam.cx = fs["cx"]; cam.cy = fs["cy"]; cam.bf = fs["bf"];
	Store store;
	auto gpu = cuba::CudaBundleAdjustment::create();
	for (const auto& n : fs["pose_vertices"])
	{
		const Eigen::Quaterniond q(readVec<4>(n["q"]));
		store.poses.push_back(std::make_unique<cuba::PoseVertex>(int(n["id"]), q, readVec<3>(n["t"]), cam, int(n["fixed"]) != 0));
		gpu->addPoseVertex(store.poses.back().get());
	}
	for (const auto& n : fs["landmark_vertices"])
	{
		store.landmarks.push_back(std::make_unique<cuba::LandmarkVertex>(int(n["id"]), readVec<3>(n["Xw"]), int(n["fixed"]) != 0));
		gpu->addLandmarkVertex(store.landmarks.back().get());
	}
	for (const auto& n : fs["monocular_edges"])
	{
		store.mono.push_back(std::make_unique<cuba::MonoEdge>(readVec<2>(n["measurement"]), double(n["information"]),
			gpu->poseVertex(int(n["vertexP"])), gpu->landmarkVertex(int(n["vertexL"]))));
		gpu->addMonocularEdge(store.mono.back().get());
	}
	for (const auto& n : fs["stereo_edges"])
	{
		store.stereo.push_back(std::make_unique<cuba::StereoEdge>(readVec<3>(n["measurement"]), double(n["information"]),
			gpu->poseVertex(int(n["vertexP"])), gpu->landmarkVertex(int(n["vertexL"]))));
		gpu->addStereoEdge(store.stereo.back().get());
	}
	gpu->setRobustKernels(cuba::RobustKernelType::HUBER, deltaMono, cuba::EdgeType::MONOCULAR);
	gpu->setRobustKernels(cuba::RobustKernelType::HUBER, deltaStereo, cuba::EdgeType::STEREO);

	// "warm-up" (ref :303-307): one iteration on each side, from the same initial estimates
	gpu->initialize();                                  // assigns the solver indices the CPU side is built from
	CpuSide cpu(store, deltaMono, deltaStereo);         // copies the initial estimates
	cpu.optimize(1);
	gpu->optimize(1);
	std::cout << "Done." << std::endl << std::endl;

	std::cout << "=== Graph size : " << std::endl;
	std::cout << "num poses      : " << gpu->nposes() << std::endl;
	std::cout << "num landmarks  : " << gpu->nlandmarks() << std::endl;
	std::cout << "num edges      : " << gpu->nedges() << std::endl << std::endl;

	std::cout << "Running BA with CPU... " << std::flush;
	const auto t0 = std::chrono::steady_clock::now();
	const std::vector<double> statsCPU = cpu.optimize(iterations);
	const auto t1 = std::chrono::steady_clock::now();
	std::cout << "Done." << std::endl << std::endl;

	std::cout << "Running BA with GPU... " << std::flush;
	const auto t2 = std::chrono::steady_clock::now();
	gpu->initialize();
	gpu->optimize(iterations);
	const auto t3 = std::chrono::steady_clock::now();
	std::cout << "Done." << std::endl << std::endl;

	std::cout << "=== Processing time : " << std::endl;
	std::printf("CPU : %9.4f [sec]   (oracle, exact sparse Cholesky, 1 thread)\n", std::chrono::duration<double>(t1 - t0).count());
	std::printf("GPU : %9.4f [sec]\n\n", std::chrono::duration<double>(t3 - t2).count());

	std::cout << "=== Objective function value : " << std::endl;
	const auto& statsGPU = gpu->batchStatistics();
	const size_t nit = std::max(statsCPU.size(), statsGPU.size());
	std::printf("%10s|%14s|%14s\n", "iteration", "chi2 CPU", "chi2 GPU");
	double worstChi = statsCPU.size() == statsGPU.size() ? 0.0 : 1.0;
	for (size_t i = 0; i < nit; i++)
	{
		std::printf("%10zu|", i + 1);
		if (i < statsCPU.size()) std::printf("%14.1f|", statsCPU[i]); else std::printf("%14s|", "N/A");
		if (i < statsGPU.size()) std::printf("%14.1f", statsGPU[i].chi2); else std::printf("%14s", "N/A");
		std::puts("");
		if (i < statsCPU.size() && i < statsGPU.size())
			worstChi = std::max(worstChi, std::fabs(statsGPU[i].chi2 - statsCPU[i]) / statsCPU[i]);
	}
	std::cout << std::endl;

	double sqR = 0, sqT = 0, sqP = 0; size_t nP = 0, nL = 0;
	for (const auto& v : store.poses)
	{
		if (v->iP < 0) continue;
		const double* qg = v->q.coeffs().data();
		for (int k = 0; k < 4; k++) sqR += (cpu.q(v->iP)[k] - qg[k]) * (cpu.q(v->iP)[k] - qg[k]);
		for (int k = 0; k < 3; k++) sqT += (cpu.t(v->iP)[k] - v->t[k]) * (cpu.t(v->iP)[k] - v->t[k]);
		nP++;
	}
	for (const auto& v : store.landmarks)
	{
		if (v->iL < 0) continue;
		for (int k = 0; k < 3; k++) sqP += (cpu.X(v->iL)[k] - v->Xw[k]) * (cpu.X(v->iL)[k] - v->Xw[k]);
		nL++;
	}
	const double rmseR = std::sqrt(sqR / std::max<size_t>(nP, 1)), rmseT = std::sqrt(sqT / std::max<size_t>(nP, 1)), rmseL = std::sqrt(sqP / std::max<size_t>(nL, 1));
	std::cout << "=== RMSE between CPU estimates and GPU estimates : " << std::endl;
	std::printf("Rotation    : %.2e\n", rmseR);
	std::printf("Translation : %.2e\n", rmseT);
	std::printf("Landmark    : %.2e\n", rmseL);
	std::printf("max relative chi2 difference : %.2e\n", worstChi);
	const bool ok = worstChi <= tolChi && rmseR <= tolR && rmseT <= tolT && rmseL <= tolL;
	std::printf("tolerances (chi2 %.0e, rotation %.0e, translation %.0e, landmark %.0e) : %s\n", tolChi, tolR, tolT, tolL, ok ? "PASS" : "FAIL");
	return ok ? 0 : 2;
}
