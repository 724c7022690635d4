// bundle_adjustment.hpp -- the cuba::CudaBundleAdjustment interface, MI355X build.
//
// Same abstract class as /root/reference/include/cuda_bundle_adjustment.h:34-125 (every method keeps
// its name, signature and documented behaviour); the implementation behind create() is a host C++
// layer over the C ABI of include/cuba_hip.h, which runs the hot path as HIP kernels on gfx950.
//
// Behaviour kept from the reference:
//   * the optimiser never deletes vertices or edges -- the caller owns them (ref .h:30-31);
//   * poseVertex(id) / landmarkVertex(id) throw std::out_of_range for unknown ids (ref .cpp:707-715);
//   * initialize() must be called after the graph changed and before optimize();
//   * optimize() may be called repeatedly (warm start); results are written back into the vertices;
//   * chiSquared(e) is 0 for edges that were inactive (both ends fixed) or unknown (ref .cpp:878-881);
//   * timeProfile() uses the reference's eight key strings (ref .cpp:545-562).
// Difference: device / numerical failures are reported by exceptions (std::runtime_error) from
// initialize()/optimize() instead of being printed and ignored (ref src/macro.h:22-27).
#pragma once

#include <cstddef>

#include "ba_types.hpp"

namespace cuba
{

class CudaBundleAdjustment
{
public:
	using Ptr = UniquePtr<CudaBundleAdjustment>;

	// Inline on purpose: the application's own view of the vertex / edge layouts (they embed Eigen types, whose size and
	// alignment depend on the Eigen the APPLICATION was compiled with) travels into the library, which refuses to start
	// when it was built against a different layout -- e.g. the library against the in-repo Eigen stand-in and the
	// application against real Eigen3, whose Quaterniond is 16-byte aligned.  Throws std::runtime_error on a mismatch.
	static Ptr create()
	{
		const size_t layout[8] = { sizeof(PoseVertex), alignof(PoseVertex), offsetof(PoseVertex, t), offsetof(PoseVertex, camera),
			sizeof(LandmarkVertex), offsetof(LandmarkVertex, fixed), sizeof(MonoEdge), sizeof(StereoEdge) };
		return createChecked(layout, 8);
	}
	static Ptr createChecked(const size_t* layout, int n);

	virtual void addPoseVertex(PoseVertex* v) = 0;
	virtual void addLandmarkVertex(LandmarkVertex* v) = 0;
	virtual void addMonocularEdge(MonoEdge* e) = 0;
	virtual void addStereoEdge(StereoEdge* e) = 0;

	virtual PoseVertex* poseVertex(int id) const = 0;
	virtual LandmarkVertex* landmarkVertex(int id) const = 0;

	virtual void removePoseVertex(PoseVertex* v) = 0;
	virtual void removeLandmarkVertex(LandmarkVertex* v) = 0;
	virtual void removeEdge(BaseEdge* e) = 0;

	virtual size_t nposes() const = 0;
	virtual size_t nlandmarks() const = 0;
	virtual size_t nedges() const = 0;

	// robust kernel for all edges of one type; default is NONE
	virtual void setRobustKernels(RobustKernelType kernelType, double delta, EdgeType edgeType) = 0;

	virtual void initialize() = 0;
	virtual void optimize(int niterations) = 0;   // Levenberg-Marquardt iterations
	virtual void clear() = 0;

	virtual const BatchStatistics& batchStatistics() const = 0;
	virtual const TimeProfile& timeProfile() const = 0;
	virtual double chiSquared(const BaseEdge* e) const = 0;

	virtual ~CudaBundleAdjustment();
};

// Extension of this library (the reference has no counterpart): optimize(niterations) of several independent objects -- ORB-SLAM's
// local-BA windows, the graphs of several agents -- as ONE device launch chain.  Every object ends exactly where its own optimize()
// would have ended (estimates in its vertices, batchStatistics(), chiSquared()), bit for bit; small graphs gain most (eight
// KITTI-07-sized graphs: 3.2 x one graph's throughput on one MI355X).  All objects must have been initialize()d.
void optimizeBatch(CudaBundleAdjustment* const* objects, int n, int niterations);

}  // namespace cuba
