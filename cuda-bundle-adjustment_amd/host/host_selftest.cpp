// host_selftest.cpp -- GPU-free check of the host layer: graph editing, ownership and the index
// assignment of initialize() (ids in order, vertices without edges skipped, free before fixed), i.e. the
// behaviour of CudaBundleAdjustmentImpl / CudaBlockSolver::initialize
// (/root/reference/src/cuda_bundle_adjustment.cpp:115-261, :681-764).  Exit code 0 = pass.
#include <cstdio>
#include <stdexcept>
#include <vector>

#include <cuda_bundle_adjustment.h>

#define REQUIRE(c) do { if (!(c)) { std::printf("FAILED: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main()
{
	using namespace cuba;
	CameraParams cam; cam.fx = cam.fy = 500; cam.cx = 320; cam.cy = 240; cam.bf = 50;
	const Eigen::Quaterniond q0;
	PoseVertex p7(7, q0, Array<double, 3>(0, 0, 0), cam), p3(3, q0, Array<double, 3>(1, 0, 0), cam, true),
		p5(5, q0, Array<double, 3>(2, 0, 0), cam), p9(9, q0, Array<double, 3>(3, 0, 0), cam);
	LandmarkVertex l20(20, Array<double, 3>(0, 0, 10)), l10(10, Array<double, 3>(1, 0, 10), true), l30(30, Array<double, 3>(2, 0, 10)),
		l40(40, Array<double, 3>(3, 0, 10));
	MonoEdge m1(Array<double, 2>(1, 2), 1.0, &p7, &l20), m2(Array<double, 2>(3, 4), 2.0, &p3, &l10);
	StereoEdge s1(Array<double, 3>(1, 2, 3), 3.0, &p3, &l20), s2(Array<double, 3>(4, 5, 6), 4.0, &p5, &l30), s3(Array<double, 3>(7, 8, 9), 5.0, &p7, &l30);

	auto ba = CudaBundleAdjustment::create();
	for (auto* p : { &p7, &p3, &p5, &p9 }) ba->addPoseVertex(p);
	for (auto* l : { &l20, &l10, &l30, &l40 }) ba->addLandmarkVertex(l);
	ba->addMonocularEdge(&m1); ba->addMonocularEdge(&m2);
	ba->addStereoEdge(&s1); ba->addStereoEdge(&s2); ba->addStereoEdge(&s3);
	REQUIRE(ba->nposes() == 4 && ba->nlandmarks() == 4 && ba->nedges() == 5);
	REQUIRE(ba->poseVertex(5) == &p5 && ba->landmarkVertex(30) == &l30);
	REQUIRE(p7.edges.size() == 2 && l30.edges.size() == 2 && p9.edges.empty());
	bool threw = false;
	try { ba->poseVertex(1234); } catch (const std::out_of_range&) { threw = true; }
	REQUIRE(threw);

	ba->initialize();
	// free poses in id order (5, 7), then the fixed one (3); pose 9 has no edges and keeps iP = -1
	REQUIRE(p5.iP == 0 && p7.iP == 1 && p3.iP == 2 && p9.iP == -1);
	REQUIRE(l20.iL == 0 && l30.iL == 1 && l10.iL == 2 && l40.iL == -1);
	REQUIRE(ba->chiSquared(&m2) == 0.0);     // inactive edge (both ends fixed)
	REQUIRE(ba->batchStatistics().empty());

	// re-initialisation: nothing changed -> same indices (the cached path); a vertex that becomes fixed without any edge
	// or vertex being added or removed must still be re-indexed (free before fixed)
	ba->initialize();
	REQUIRE(p5.iP == 0 && p7.iP == 1 && p3.iP == 2 && l20.iL == 0 && l30.iL == 1 && l10.iL == 2);
	p5.fixed = true;
	ba->initialize();
	REQUIRE(p7.iP == 0 && p3.iP == 1 && p5.iP == 2);
	l30.fixed = true;                        // s2 (p5 - l30) now has both ends fixed: inactive, reports 0
	ba->initialize();
	REQUIRE(l20.iL == 0 && l10.iL == 1 && l30.iL == 2 && ba->chiSquared(&s2) == 0.0);
	p5.fixed = false; l30.fixed = false;
	ba->initialize();
	REQUIRE(p5.iP == 0 && p7.iP == 1 && p3.iP == 2 && l20.iL == 0 && l30.iL == 1 && l10.iL == 2);

	ba->removeEdge(&s3);
	REQUIRE(ba->nedges() == 4 && p7.edges.size() == 1 && l30.edges.size() == 1);
	ba->removePoseVertex(&p7);               // removes its remaining edge m1 too (no iteration-while-erasing UB)
	REQUIRE(ba->nposes() == 3 && ba->nedges() == 3 && l20.edges.size() == 1);
	ba->initialize();
	REQUIRE(p5.iP == 0 && p3.iP == 1 && l20.iL == 0 && l30.iL == 1);
	ba->clear();
	REQUIRE(ba->nposes() == 0 && ba->nedges() == 0);
	threw = false;
	try { ba->optimize(1); } catch (const std::runtime_error&) { threw = true; }   // initialize() is required again
	REQUIRE(threw);
	// an application compiled against another Eigen (different vertex layout) is refused instead of corrupting memory
	{
		size_t layout[8] = { sizeof(cuba::PoseVertex) + 16, 16, 32, 64, sizeof(cuba::LandmarkVertex), 24, sizeof(cuba::MonoEdge), sizeof(cuba::StereoEdge) };
		threw = false;
		try { auto bad = cuba::CudaBundleAdjustment::createChecked(layout, 8); } catch (const std::runtime_error&) { threw = true; }
		REQUIRE(threw);
	}
	std::printf("host_selftest: all checks passed\n");
	return 0;
}
