// edit_fuzz.cpp -- randomised sweep over the C++ API's incremental paths (not one of the reference's samples): one long-lived
// cuba::CudaBundleAdjustment object is edited at random between optimisations -- measurements changed, `fixed` flags toggled, edges removed and
// added back, landmarks removed with their edges, estimates moved by the caller, nothing changed at all -- and after every edit its
// initialize() + optimize(k) is compared, estimate by estimate and bit for bit, with a FRESH object built from the same vertices, edges and
// estimates.  What the long-lived object keeps between calls (id-ordered vertex lists, the previous flattening, the same-topology and
// unchanged-values promises to the device library, the device-side structure) must never show in a result.  Run with CUBA_HIP_HEURISTICS=0:
// the run-to-run memories of the device library (first coarse inverse, iteration counts) are what a fresh object cannot have.
//
//   usage: edit_fuzz graph.json [rounds=30] [seed=1]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <random>
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

struct MPose { int id; double q[4], t[3]; bool fixed; };
struct MLm { int id; double X[3]; bool fixed; bool alive; };
struct MEdge { int dim, ip, il; double m[3], info; bool alive; };      // ip / il: indices into the model's vertex arrays

struct Built
{
	std::unique_ptr<cuba::CudaBundleAdjustment> ba;
	std::vector<std::unique_ptr<cuba::PoseVertex>> poses;
	std::vector<std::unique_ptr<cuba::LandmarkVertex>> lms;
	std::vector<std::unique_ptr<cuba::BaseEdge>> edges;      // by model edge index (null: never added)
};

static void addEdgeTo(Built& b, const std::vector<MEdge>& E, int k)
{
	const MEdge& e = E[k];
	if (e.dim == 2)
	{
		cuba::Array<double, 2> m; m[0] = e.m[0]; m[1] = e.m[1];
		auto p = std::make_unique<cuba::MonoEdge>(m, e.info, b.poses[e.ip].get(), b.lms[e.il].get());
		b.ba->addMonocularEdge(p.get()); b.edges[k] = std::move(p);
	}
	else
	{
		cuba::Array<double, 3> m; m[0] = e.m[0]; m[1] = e.m[1]; m[2] = e.m[2];
		auto p = std::make_unique<cuba::StereoEdge>(m, e.info, b.poses[e.ip].get(), b.lms[e.il].get());
		b.ba->addStereoEdge(p.get()); b.edges[k] = std::move(p);
	}
}

static Built build(const std::vector<MPose>& P, const std::vector<MLm>& L, const std::vector<MEdge>& E, const std::vector<int>& order, const cuba::CameraParams& cam)
{
	Built b;
	b.ba = cuba::CudaBundleAdjustment::create();
	for (const MPose& p : P)
	{
		cuba::Array<double, 4> q; for (int k = 0; k < 4; k++) q[k] = p.q[k];
		cuba::Array<double, 3> t; for (int k = 0; k < 3; k++) t[k] = p.t[k];
		b.poses.push_back(std::make_unique<cuba::PoseVertex>(p.id, Eigen::Quaterniond(q), t, cam, p.fixed));
		b.ba->addPoseVertex(b.poses.back().get());
	}
	for (const MLm& l : L)
	{
		cuba::Array<double, 3> X; for (int k = 0; k < 3; k++) X[k] = l.X[k];
		b.lms.push_back(std::make_unique<cuba::LandmarkVertex>(l.id, X, l.fixed));
		if (l.alive) b.ba->addLandmarkVertex(b.lms.back().get());
	}
	b.edges.resize(E.size());
	for (int k : order) if (E[k].alive) addEdgeTo(b, E, k);
	b.ba->setRobustKernels(cuba::RobustKernelType::HUBER, std::sqrt(5.991), cuba::EdgeType::MONOCULAR);
	b.ba->setRobustKernels(cuba::RobustKernelType::HUBER, std::sqrt(7.815), cuba::EdgeType::STEREO);
	return b;
}

int main(int argc, char** argv)
{
	if (argc < 2) { std::printf("usage: %s graph.json [rounds=30] [seed=1]\n", argv[0]); return 0; }
	const int rounds = argc > 2 ? std::atoi(argv[2]) : 30;
	std::mt19937_64 rng(argc > 3 ? std::atoll(argv[3]) : 1);
	cv::FileStorage fs(argv[1], cv::FileStorage::READ);
	if (!fs.isOpened()) { std::fprintf(stderr, "cannot open %s\n", argv[1]); return 1; }
	cuba::CameraParams cam;
	cam.fx = fs["fx"]; cam.fy = fs["fy"]; cam.cx = fs["cx"]; cam.cy = fs["cy"]; cam.bf = fs["bf"];
	std::vector<MPose> P; std::vector<MLm> L; std::vector<MEdge> E;
	std::vector<int> poseOfId, lmOfId;
	auto slot = [](std::vector<int>& v, int id) -> int& { if ((int)v.size() <= id) v.resize(id + 1, -1); return v[id]; };
	for (const auto& n : fs["pose_vertices"])
	{
		MPose p; p.id = n["id"]; p.fixed = int(n["fixed"]) != 0;
		const auto q = readVec<4>(n["q"]); const auto t = readVec<3>(n["t"]);
		for (int k = 0; k < 4; k++) p.q[k] = q[k];
		for (int k = 0; k < 3; k++) p.t[k] = t[k];
		slot(poseOfId, p.id) = (int)P.size(); P.push_back(p);
	}
	for (const auto& n : fs["landmark_vertices"])
	{
		MLm l; l.id = n["id"]; l.fixed = int(n["fixed"]) != 0; l.alive = true;
		const auto X = readVec<3>(n["Xw"]);
		for (int k = 0; k < 3; k++) l.X[k] = X[k];
		slot(lmOfId, l.id) = (int)L.size(); L.push_back(l);
	}
	for (const auto& n : fs["monocular_edges"])
	{
		MEdge e; e.dim = 2; e.ip = poseOfId[int(n["vertexP"])]; e.il = lmOfId[int(n["vertexL"])]; e.info = double(n["information"]); e.alive = true;
		const auto m = readVec<2>(n["measurement"]); e.m[0] = m[0]; e.m[1] = m[1]; e.m[2] = 0; E.push_back(e);
	}
	for (const auto& n : fs["stereo_edges"])
	{
		MEdge e; e.dim = 3; e.ip = poseOfId[int(n["vertexP"])]; e.il = lmOfId[int(n["vertexL"])]; e.info = double(n["information"]); e.alive = true;
		const auto m = readVec<3>(n["measurement"]); for (int k = 0; k < 3; k++) e.m[k] = m[k]; E.push_back(e);
	}
	std::vector<int> order(E.size());
	for (size_t k = 0; k < E.size(); k++) order[k] = (int)k;
	Built A = build(P, L, E, order, cam);
	std::uniform_real_distribution<double> U(0.0, 1.0);
	std::normal_distribution<double> N01(0.0, 1.0);
	auto pick = [&](size_t n) { return (size_t)(U(rng) * n) % n; };
	int failures = 0;
	const char* names[] = { "nothing", "measurements", "fixed flags", "remove edges", "add edges back", "remove a landmark", "move estimates", "remove + add in one go" };
	for (int rd = 0; rd < rounds; rd++)
	{
		const int action = rd == 0 ? 0 : (int)pick(8);
		size_t touched = 0;
		auto removeEdgeK = [&](size_t k) { E[k].alive = false; A.ba->removeEdge(A.edges[k].get()); touched++; };
		auto addEdgeK = [&](size_t k) {
			E[k].alive = true;
			// (the long-lived object appends a returning edge to its insertion order: the model's order follows)
			for (size_t i = 0; i < order.size(); i++) if (order[i] == (int)k) { order.erase(order.begin() + i); break; }
			order.push_back((int)k);
			if (E[k].dim == 2) A.ba->addMonocularEdge(static_cast<cuba::MonoEdge*>(A.edges[k].get())); else A.ba->addStereoEdge(static_cast<cuba::StereoEdge*>(A.edges[k].get()));
			touched++;
		};
		if (action == 1)
			for (size_t k = 0; k < E.size(); k++) if (E[k].alive && U(rng) < 0.05)
			{
				for (int c = 0; c < E[k].dim; c++) E[k].m[c] += 0.3 * N01(rng);
				if (E[k].dim == 2) { auto* e = static_cast<cuba::MonoEdge*>(A.edges[k].get()); e->measurement[0] = E[k].m[0]; e->measurement[1] = E[k].m[1]; }
				else { auto* e = static_cast<cuba::StereoEdge*>(A.edges[k].get()); for (int c = 0; c < 3; c++) e->measurement[c] = E[k].m[c]; }
				touched++;
			}
		if (action == 2)
		{
			for (int n = 0; n < 3; n++) { const size_t i = pick(P.size()); P[i].fixed = !P[i].fixed; touched++; }
			size_t nfix = 0; for (const MPose& p : P) nfix += p.fixed;
			if (nfix == 0) P[0].fixed = true;
			if (nfix == P.size()) P[P.size() / 2].fixed = false;
			for (size_t i = 0; i < P.size(); i++) A.poses[i]->fixed = P[i].fixed;
			for (int n = 0; n < 20; n++) { const size_t i = pick(L.size()); L[i].fixed = !L[i].fixed; A.lms[i]->fixed = L[i].fixed; }
		}
		if (action == 3 || action == 7)
			for (size_t k = 0; k < E.size(); k++) if (E[k].alive && U(rng) < 0.03) removeEdgeK(k);
		if (action == 4 || action == 7)
			for (size_t k = 0; k < E.size(); k++) if (!E[k].alive && L[E[k].il].alive && U(rng) < 0.5) addEdgeK(k);
		if (action == 5)
		{
			const size_t i = pick(L.size());
			if (L[i].alive)
			{
				L[i].alive = false;
				for (size_t k = 0; k < E.size(); k++) if (E[k].il == (int)i) E[k].alive = false;
				A.ba->removeLandmarkVertex(A.lms[i].get()); touched++;
			}
		}
		if (action == 6)
			for (size_t i = 0; i < L.size(); i++) if (U(rng) < 0.2)
			{
				for (int c = 0; c < 3; c++) { L[i].X[c] += 0.02 * N01(rng); A.lms[i]->Xw[c] = L[i].X[c]; }
				touched++;
			}
		const int iters = 1 + (int)pick(4);
		// the fresh object first sees exactly what the long-lived one is about to see
		Built B = build(P, L, E, order, cam);
		A.ba->initialize(); A.ba->optimize(iters);
		B.ba->initialize(); B.ba->optimize(iters);
		size_t diffs = 0;
		for (size_t i = 0; i < P.size(); i++)
			diffs += std::memcmp(A.poses[i]->q.coeffs().data(), B.poses[i]->q.coeffs().data(), 32) != 0 || std::memcmp(A.poses[i]->t.data(), B.poses[i]->t.data(), 24) != 0;
		for (size_t i = 0; i < L.size(); i++) diffs += std::memcmp(A.lms[i]->Xw.data(), B.lms[i]->Xw.data(), 24) != 0;
		const auto& sa = A.ba->batchStatistics(); const auto& sb = B.ba->batchStatistics();
		bool chiSame = sa.size() == sb.size();
		for (size_t i = 0; chiSame && i < sa.size(); i++) chiSame = sa[i].chi2 == sb[i].chi2;
		// per-edge chi2 of a few edges
		size_t chiEdgeDiffs = 0;
		for (int n = 0; n < 50; n++) { const size_t k = pick(E.size()); if (E[k].alive) chiEdgeDiffs += A.ba->chiSquared(A.edges[k].get()) != B.ba->chiSquared(B.edges[k].get()); }
		const bool ok = diffs == 0 && chiSame && chiEdgeDiffs == 0 && A.ba->nedges() == B.ba->nedges();
		if (!ok) failures++;
		std::printf("round %2d  %-22s touched %6zu  iterations %d  edges %zu  chi2 %.6f  %s", rd, names[action], touched, iters, A.ba->nedges(), sa.empty() ? 0.0 : sa.back().chi2,
			ok ? "identical\n" : "DIFFERENT\n");
		if (!ok) std::printf("          vertices that differ %zu, chi2 trajectories equal %d, per-edge chi2 differences %zu, edges %zu / %zu\n", diffs, (int)chiSame, chiEdgeDiffs, A.ba->nedges(), B.ba->nedges());
		// the model continues from the long-lived object's estimates
		for (size_t i = 0; i < P.size(); i++) { for (int k = 0; k < 4; k++) P[i].q[k] = A.poses[i]->q.coeffs().data()[k]; for (int k = 0; k < 3; k++) P[i].t[k] = A.poses[i]->t.data()[k]; }
		for (size_t i = 0; i < L.size(); i++) for (int k = 0; k < 3; k++) L[i].X[k] = A.lms[i]->Xw.data()[k];
	}
	std::printf("%d rounds, %d failures\n", rounds, failures);
	return failures ? 1 : 0;
}
