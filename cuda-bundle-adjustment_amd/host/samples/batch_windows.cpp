// batch_windows.cpp -- several independent bundle-adjustment problems optimised at once through the C++ API (cuba::optimizeBatch, this
// library's extension over the reference's interface: include/cuda_bundle_adjustment.h:34-125 has independent objects, optimised one
// optimize() call at a time).  The usage pattern: the local-BA windows of several agents, or of ORB-SLAM's local-mapping and
// loop-closing threads, handed to the GPU together.  The same graph file is loaded N times into N objects -- every copy with its own vertex
// and edge objects, window k with the initial landmark estimates moved by k millimetres so that the runs differ --, then the windows are
// optimised (a) one optimize() after the other and (b), from the same starts, by one optimizeBatch() call.  Prints per window the chi2 of
// both ways (they must agree bit for bit: tests/test_host_cpp.py) and the two wall times.
//
//   usage: batch_windows graph.json [windows=4] [iterations=10]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include <opencv2/core.hpp>
#include <cuda_bundle_adjustment.h>

template <int N>
static cuba::Array<double, N> readVec(const cv::FileNode& node)
{
	cuba::Array<double, N> a;
	int k = 0;
	for (const auto& v : node) { if (k >= N) break; a[k++] = double(v); }
	return a;
}

struct Window
{
	std::vector<std::unique_ptr<cuba::PoseVertex>> poses;
	std::vector<std::unique_ptr<cuba::LandmarkVertex>> landmarks;
	std::vector<std::unique_ptr<cuba::MonoEdge>> mono;
	std::vector<std::unique_ptr<cuba::StereoEdge>> stereo;
	cuba::CudaBundleAdjustment::Ptr ba;
	std::vector<cuba::Array<double, 4>> q0; std::vector<cuba::Array<double, 3>> t0, X0;

	void load(const cv::FileStorage& fs, double shift)
	{
		cuba::CameraParams cam;
		cam.fx = fs["fx"]; cam.fy = fs["fy"]; cam.cx = fs["cx"]; cam.cy = fs["cy"]; cam.bf = fs["bf"];
		ba = cuba::CudaBundleAdjustment::create();
		for (const auto& n : fs["pose_vertices"])
		{
			poses.push_back(std::make_unique<cuba::PoseVertex>(int(n["id"]), Eigen::Quaterniond(readVec<4>(n["q"])), readVec<3>(n["t"]), cam, int(n["fixed"]) != 0));
			ba->addPoseVertex(poses.back().get());
		}
		for (const auto& n : fs["landmark_vertices"])
		{
			auto X = readVec<3>(n["Xw"]);
			X[0] += shift;
			landmarks.push_back(std::make_unique<cuba::LandmarkVertex>(int(n["id"]), X, int(n["fixed"]) != 0));
			ba->addLandmarkVertex(landmarks.back().get());
		}
		for (const auto& n : fs["monocular_edges"])
		{
			mono.push_back(std::make_unique<cuba::MonoEdge>(readVec<2>(n["measurement"]), double(n["information"]),
				ba->poseVertex(int(n["vertexP"])), ba->landmarkVertex(int(n["vertexL"]))));
			ba->addMonocularEdge(mono.back().get());
		}
		for (const auto& n : fs["stereo_edges"])
		{
			stereo.push_back(std::make_unique<cuba::StereoEdge>(readVec<3>(n["measurement"]), double(n["information"]),
				ba->poseVertex(int(n["vertexP"])), ba->landmarkVertex(int(n["vertexL"]))));
			ba->addStereoEdge(stereo.back().get());
		}
		ba->setRobustKernels(cuba::RobustKernelType::HUBER, std::sqrt(5.991), cuba::EdgeType::MONOCULAR);
		ba->setRobustKernels(cuba::RobustKernelType::HUBER, std::sqrt(7.815), cuba::EdgeType::STEREO);
		for (auto& p : poses) { cuba::Array<double, 4> q; for (int k = 0; k < 4; k++) q[k] = p->q.coeffs().data()[k]; q0.push_back(q); t0.push_back(p->t); }
		for (auto& l : landmarks) X0.push_back(l->Xw);
	}
	void reset()
	{
		for (size_t i = 0; i < poses.size(); i++) { for (int k = 0; k < 4; k++) poses[i]->q.coeffs().data()[k] = q0[i][k]; poses[i]->t = t0[i]; }
		for (size_t i = 0; i < landmarks.size(); i++) landmarks[i]->Xw = X0[i];
	}
};

int main(int argc, char** argv)
{
	if (argc < 2) { std::printf("usage: %s graph.json [windows=4] [iterations=10]\n", argv[0]); return 0; }
	const int nw = argc > 2 ? std::atoi(argv[2]) : 4, iters = argc > 3 ? std::atoi(argv[3]) : 10;
	cv::FileStorage fs(argv[1], cv::FileStorage::READ);
	if (!fs.isOpened()) { std::fprintf(stderr, "cannot open %s\n", argv[1]); return 1; }
	std::vector<Window> w((size_t)nw);
	for (int k = 0; k < nw; k++) w[k].load(fs, 1e-3 * k);
	using Clock = std::chrono::steady_clock;
	// warm-up of every object (structure analysis, first launches), as the reference's samples do
	for (auto& x : w) { x.ba->initialize(); x.ba->optimize(1); }
	// (a) one after the other
	for (auto& x : w) x.reset();
	const auto t0 = Clock::now();
	for (auto& x : w) { x.ba->initialize(); x.ba->optimize(iters); }
	const double solo = std::chrono::duration<double>(Clock::now() - t0).count();
	std::vector<std::vector<double>> chiSolo;
	for (auto& x : w) { std::vector<double> c; for (const auto& s : x.ba->batchStatistics()) c.push_back(s.chi2); chiSolo.push_back(c); }
	// (b) together
	for (auto& x : w) x.reset();
	std::vector<cuba::CudaBundleAdjustment*> objs;
	for (auto& x : w) objs.push_back(x.ba.get());
	const auto t1 = Clock::now();
	for (auto& x : w) x.ba->initialize();
	cuba::optimizeBatch(objs.data(), nw, iters);
	const double batch = std::chrono::duration<double>(Clock::now() - t1).count();
	bool same = true;
	for (int k = 0; k < nw; k++)
	{
		const auto& st = w[k].ba->batchStatistics();
		std::printf("window %d: %zu iterations, chi2 solo %.10e batch %.10e\n", k, st.size(), chiSolo[k].empty() ? 0.0 : chiSolo[k].back(), st.empty() ? 0.0 : st.back().chi2);
		same = same && st.size() == chiSolo[k].size();
		for (size_t i = 0; same && i < st.size(); i++) same = st[i].chi2 == chiSolo[k][i];
	}
	std::printf("bit-identical: %s\n", same ? "yes" : "NO");
	std::printf("wall initialize() + optimize(%d) of %d windows: one after the other %.3f ms, optimizeBatch %.3f ms\n", iters, nw, 1e3 * solo, 1e3 * batch);
	return same ? 0 : 2;
}
