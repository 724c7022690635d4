// ba_from_file.cpp -- reads a bundle-adjustment graph (the JSON schema of the reference's datasets),
// runs initialize() + optimize(N) through the cuba::CudaBundleAdjustment API and prints timing, the
// per-stage profile and the objective per iteration.  Counterpart of the reference's
// samples/sample_ba_from_file.cpp (same protocol: one warm-up initialize()+optimize(1), then the timed
// run), written against our dependency-free JSON reader.
//
//   usage: sample_ba_from_file graph.json [iterations=10] [huber=1]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include <opencv2/core.hpp>   // mini_opencv stand-in (JSON reader) unless real OpenCV is on the include path

#include <cuda_bundle_adjustment.h>

namespace
{
struct Store   // owns the vertices and edges: the optimiser never deletes them
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
}  // namespace

int main(int argc, char** argv)
{
	if (argc < 2) { std::printf("usage: %s graph.json [iterations=10] [huber=1]\n", argv[0]); return 0; }
	const int iterations = argc > 2 ? std::atoi(argv[2]) : 10;
	const bool huber = argc > 3 ? std::atoi(argv[3]) != 0 : true;

	cv::FileStorage fs(argv[1], cv::FileStorage::READ);
	if (!fs.isOpened()) { std::fprintf(stderr, "cannot open %s\n", argv[1]); return 1; }
	cuba::CameraParams cam;
	cam.fx = fs["fx"]; cam.fy = fs["fy"]; cam.cx = fs["cx"]; cam.cy = fs["cy"]; cam.bf = fs["bf"];

	Store store;
	auto ba = cuba::CudaBundleAdjustment::create();
	for (const auto& n : fs["pose_vertices"])
	{
		const Eigen::Quaterniond q(readVec<4>(n["q"]));
		store.poses.push_back(std::make_unique<cuba::PoseVertex>(int(n["id"]), q, readVec<3>(n["t"]), cam, int(n["fixed"]) != 0));
		ba->addPoseVertex(store.poses.back().get());
	}
	for (const auto& n : fs["landmark_vertices"])
	{
		store.landmarks.push_back(std::make_unique<cuba::LandmarkVertex>(int(n["id"]), readVec<3>(n["Xw"]), int(n["fixed"]) != 0));
		ba->addLandmarkVertex(store.landmarks.back().get());
	}
	for (const auto& n : fs["monocular_edges"])
	{
		store.mono.push_back(std::make_unique<cuba::MonoEdge>(readVec<2>(n["measurement"]), double(n["information"]),
			ba->poseVertex(int(n["vertexP"])), ba->landmarkVertex(int(n["vertexL"]))));
		ba->addMonocularEdge(store.mono.back().get());
	}
	for (const auto& n : fs["stereo_edges"])
	{
		store.stereo.push_back(std::make_unique<cuba::StereoEdge>(readVec<3>(n["measurement"]), double(n["information"]),
			ba->poseVertex(int(n["vertexP"])), ba->landmarkVertex(int(n["vertexL"]))));
		ba->addStereoEdge(store.stereo.back().get());
	}
	if (huber)
	{
		ba->setRobustKernels(cuba::RobustKernelType::HUBER, std::sqrt(5.991), cuba::EdgeType::MONOCULAR);
		ba->setRobustKernels(cuba::RobustKernelType::HUBER, std::sqrt(7.815), cuba::EdgeType::STEREO);
	}
	std::printf("poses %zu  landmarks %zu  edges %zu\n", ba->nposes(), ba->nlandmarks(), ba->nedges());

	ba->initialize();        // warm-up, as in the reference's sample (it moves the estimates by one LM step)
	ba->optimize(1);

	const auto t0 = std::chrono::steady_clock::now();
	ba->initialize();
	ba->optimize(iterations);
	const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

	std::printf("BA total : %.4f [sec]  (%.2f M edge-iterations/s)\n", sec,
		1e-6 * double(ba->nedges()) * double(ba->batchStatistics().size()) / sec);
	for (const auto& kv : ba->timeProfile()) std::printf("%-30s : %8.1f[msec]\n", kv.first.c_str(), 1e3 * kv.second);
	for (const auto& s : ba->batchStatistics()) std::printf("iter: %2d, chi2: %.6f\n", s.iteration + 1, s.chi2);
	return 0;
}
