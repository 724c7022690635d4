// ba_types.hpp -- user-facing graph types of the cuba:: bundle-adjustment API (MI355X build).
//
// Source-compatible with the type header of fixstars/cuda-bundle-adjustment
// (/root/reference/include/cuda_bundle_adjustment_types.h:30-247): same names, members, constructors
// and ownership rules, so code written against the reference (e.g. its samples) compiles unchanged.
// The caller allocates and owns every vertex and edge; the optimiser keeps raw pointers, fills
// PoseVertex::iP / LandmarkVertex::iL and the per-vertex edge sets, and writes the optimised
// q / t / Xw back into the vertices.
#pragma once

#include <map>
#include <memory>
#include <string>
#include <unordered_set>
#include <vector>

#include <Eigen/Core>
#include <Eigen/Geometry>

namespace cuba
{

// ---- aliases (types.h:36-43) ---------------------------------------------------------------------
template <class T, int N> using Array = Eigen::Matrix<T, N, 1>;
template <class T> using Set = std::unordered_set<T>;
template <class T> using UniquePtr = std::unique_ptr<T>;

// ---- pinhole / rectified-stereo intrinsics (types.h:51-62) ---------------------------------------
struct CameraParams
{
	double fx = 0;   // focal length x [px]
	double fy = 0;   // focal length y [px]
	double cx = 0;   // principal point x [px]
	double cy = 0;   // principal point y [px]
	double bf = 0;   // stereo baseline times fx
	CameraParams() = default;
};

struct PoseVertex;
struct LandmarkVertex;

// ---- edges (types.h:73-148) ----------------------------------------------------------------------
struct BaseEdge
{
	virtual PoseVertex* poseVertex() const = 0;
	virtual LandmarkVertex* landmarkVertex() const = 0;
	virtual int dim() const = 0;            // 2 = monocular, 3 = stereo
	virtual ~BaseEdge() = default;
};

template <int DIM>
struct Edge : BaseEdge
{
	using Measurement = Array<double, DIM>;
	using Information = double;             // isotropic information (a scalar, as in the reference)

	Edge() : measurement(Measurement()), information(Information()), vertexP(nullptr), vertexL(nullptr) {}
	Edge(const Measurement& m, Information I, PoseVertex* vertexP, LandmarkVertex* vertexL)
		: measurement(m), information(I), vertexP(vertexP), vertexL(vertexL) {}

	PoseVertex* poseVertex() const override { return vertexP; }
	LandmarkVertex* landmarkVertex() const override { return vertexL; }
	int dim() const override { return DIM; }

	Measurement measurement;
	Information information;
	PoseVertex* vertexP;
	LandmarkVertex* vertexL;
};

using MonoEdge = Edge<2>;
using StereoEdge = Edge<3>;

enum class EdgeType { MONOCULAR = 0, STEREO = 1, COUNT = 2 };

// ---- vertices (types.h:156-208) -------------------------------------------------------------------
struct PoseVertex
{
	using Quaternion = Eigen::Quaterniond;
	using Rotation = Quaternion;
	using Translation = Array<double, 3>;

	PoseVertex() : q(Rotation()), t(Translation()), fixed(false), id(-1), iP(-1) {}
	PoseVertex(int id, const Rotation& q, const Translation& t, const CameraParams& camera, bool fixed = false)
		: q(q), t(t), camera(camera), fixed(fixed), id(id), iP(-1) {}

	Rotation q;              // world -> camera rotation
	Translation t;           // world -> camera translation
	CameraParams camera;
	bool fixed;              // held constant during optimisation
	int id;
	int iP;                  // solver index, assigned by initialize()
	Set<BaseEdge*> edges;    // incident edges, maintained by add*/remove*
};

struct LandmarkVertex
{
	using Point3D = Array<double, 3>;

	LandmarkVertex() : Xw(Point3D()), fixed(false), id(-1), iL(-1) {}
	LandmarkVertex(int id, const Point3D& Xw, bool fixed = false) : Xw(Xw), fixed(fixed), id(id), iL(-1) {}

	Point3D Xw;
	bool fixed;
	int id;
	int iL;                  // solver index, assigned by initialize()
	Set<BaseEdge*> edges;
};

// ---- robust kernels / statistics (types.h:213-236) -------------------------------------------------
enum class RobustKernelType { NONE = 0, HUBER = 1, TUKEY = 2 };

struct BatchInfo
{
	int iteration;
	double chi2;             // robust objective after the iteration
};

using BatchStatistics = std::vector<BatchInfo>;
using TimeProfile = std::map<std::string, double>;

// short names used inside the reference's sources (types.h:242-245)
using VertexP = PoseVertex;
using VertexL = LandmarkVertex;
using Edge2D = MonoEdge;
using Edge3D = StereoEdge;

}  // namespace cuba
