// local_ba_flow.cpp -- the usage pattern the reference's README names as its real-world integration (ORB-SLAM2's
// local bundle adjustment): many poses held fixed, a first optimisation with robust kernels, per-edge chi-squared
// queries to flag outliers (cuba::CudaBundleAdjustment::chiSquared, ref src/cuda_bundle_adjustment.cpp:878-881),
// removal of the flagged edges (removeEdge, :731-764), re-initialize() and a second optimisation without them.
// Prints the objective of both stages and the number of removed edges; tests/test_host_cpp.py checks the numbers
// against the same flow driven through the C ABI from Python.
//
//   usage: local_ba_flow graph.json [fix_every=3] [iters1=5] [iters2=10]
#include <cmath>
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

int main(int argc, char** argv)
{
	if (argc < 2) { std::printf("usage: %s graph.json [fix_every=3] [iters1=5] [iters2=10]\n", argv[0]); return 0; }
	const int fixEvery = argc > 2 ? std::atoi(argv[2]) : 3;
	const int iters1 = argc > 3 ? std::atoi(argv[3]) : 5, iters2 = argc > 4 ? std::atoi(argv[4]) : 10;
	cv::FileStorage fs(argv[1], cv::FileStorage::READ);
	if (!fs.isOpened()) { std::fprintf(stderr, "cannot open %s\n", argv[1]); return 1; }
	cuba::CameraParams cam;
	cam.fx = fs["fx"]; cam.fy = fs["fy"]; cam.cx = fs["cx"]; cam.cy = fs["cy"]; cam.bf = fs["bf"];

	std::vector<std::unique_ptr<cuba::PoseVertex>> poses;
	std::vector<std::unique_ptr<cuba::LandmarkVertex>> landmarks;
	std::vector<std::unique_ptr<cuba::MonoEdge>> mono;
	std::vector<std::unique_ptr<cuba::StereoEdge>> stereo;
	auto ba = cuba::CudaBundleAdjustment::create();
	for (const auto& n : fs["pose_vertices"])
	{
		const int id = n["id"];
		const bool fixed = int(n["fixed"]) != 0 || (fixEvery > 0 && id % fixEvery == 0);   // "covisible but not local" keyframes
		poses.push_back(std::make_unique<cuba::PoseVertex>(id, Eigen::Quaterniond(readVec<4>(n["q"])), readVec<3>(n["t"]), cam, fixed));
		ba->addPoseVertex(poses.back().get());
	}
	for (const auto& n : fs["landmark_vertices"])
	{
		landmarks.push_back(std::make_unique<cuba::LandmarkVertex>(int(n["id"]), readVec<3>(n["Xw"]), int(n["fixed"]) != 0));
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

	// stage 1: robust optimisation
	ba->initialize();
	ba->optimize(iters1);
	for (const auto& s : ba->batchStatistics()) std::printf("stage1 iter: %2d, chi2: %.6f\n", s.iteration + 1, s.chi2);

	// outlier rejection on the per-edge chi-squared (ORB-SLAM2's thresholds), then the edges leave the graph
	size_t removed = 0;
	for (auto& e : mono) if (ba->chiSquared(e.get()) > 5.991) { ba->removeEdge(e.get()); removed++; }
	for (auto& e : stereo) if (ba->chiSquared(e.get()) > 7.815) { ba->removeEdge(e.get()); removed++; }
	std::printf("removed %zu of %zu edges, %zu left\n", removed, mono.size() + stereo.size(), ba->nedges());

	// stage 2: plain least squares on the inliers
	ba->setRobustKernels(cuba::RobustKernelType::NONE, 0, cuba::EdgeType::MONOCULAR);
	ba->setRobustKernels(cuba::RobustKernelType::NONE, 0, cuba::EdgeType::STEREO);
	ba->initialize();
	ba->optimize(iters2);
	for (const auto& s : ba->batchStatistics()) std::printf("stage2 iter: %2d, chi2: %.6f\n", s.iteration + 1, s.chi2);

	// stage 3: the same graph with new values -- every measurement moves by a quarter pixel, every information halves --
	// and no vertex or edge added or removed: initialize() and the device library keep what depends on the topology
	for (auto& e : mono) { e->measurement[0] += 0.25; e->measurement[1] -= 0.25; e->information *= 0.5; }
	for (auto& e : stereo) { e->measurement[0] += 0.25; e->measurement[1] -= 0.25; e->measurement[2] += 0.25; e->information *= 0.5; }
	ba->initialize();
	ba->optimize(3);
	for (const auto& s : ba->batchStatistics()) std::printf("stage3 iter: %2d, chi2: %.6f\n", s.iteration + 1, s.chi2);

	// stage 4: initialize() again with NOTHING changed but the estimates stage 3 left -- the library is told that edges and edge values
	// are those it already holds (cuba_hip_hint_unchanged) and uploads the estimates only; then one measurement changes, which the
	// next initialize() must notice
	ba->initialize();
	ba->optimize(2);
	for (const auto& s : ba->batchStatistics()) std::printf("stage4 iter: %2d, chi2: %.6f\n", s.iteration + 1, s.chi2);
	if (!mono.empty()) mono.front()->measurement[0] += 40.0; else stereo.front()->measurement[0] += 40.0;
	ba->initialize();
	ba->optimize(1);
	for (const auto& s : ba->batchStatistics()) std::printf("stage5 iter: %2d, chi2: %.6f\n", s.iteration + 1, s.chi2);
	return 0;
}
