// ba_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement (fp64, single thread, plain loops) of the bundle-adjustment hot path of
// fixstars/cuda-bundle-adjustment, used ONLY as the checker for the HIP path (tests/,
// __graft_entry__.smoke(), and bench.py's cpu_baseline leg).  Nothing under
// cuda-bundle-adjustment_amd/ may include, link or call this file.
//
// PARITY PIN: the reference ships no unit tests / golden vectors for this path, and its only known answers
// (README.md:141-150,176-191 chi2 tables) need samples/ba_input.7z, which is absent from the checkout.  The pin
// is therefore "outputs of the reference itself run here": oracle/_ref = the reference's own device layer
// (src/cuda_block_solver.cu compiled in place through a CUDA->HIP name shim, oracle/ref_build/) runs one LM
// trial on the MI355X and tests/test_ref_kernels.py requires this file to reproduce every stage output
// (chi2, Hpp/bp/Hll/bl, max diagonal, bsc, Hsc, Hll^-1, xl, scale, updated estimates, per-edge chi2).
// The LM CONTROLLER is pinned the same way: oracle/_ref/libcuba_ref_lm.so is the reference's whole optimiser
// (src/cuda_bundle_adjustment.cpp: CudaBundleAdjustmentImpl::optimize + CudaBlockSolver, src/sparse_block_matrix.cpp and
// src/cuda_block_solver.cu, compiled in place) and tests/test_ref_lm.py requires this file's optimize() to reproduce its
// chi2 trajectories (<= 1e-9 relative), iteration counts, final estimates and per-edge chi2, including the pose-only /
// landmark-only modes (the reference's gpu::solveDiagonalSystem) and runs with rejected trials.
// One part of the path remains UNPINNED against the reference because it cannot run here:
//   * the reduced-system solve -- NVIDIA cuSOLVER sparse Cholesky (src/cuda_linear_solver.cpp:147-232, closed
//     source), restated as an exact sparse block Cholesky with min-degree ordering (any exact SPD solve is
//     equivalent up to rounding; pinned against numpy dense solves instead; in libcuba_ref_lm.so a dense host
//     Cholesky stands in for it).
// Further independent pins in tests/: finite-difference Jacobians, mpmath exp-map, dense full-system solves.
//
// Every function cites the reference lines (under /root/reference/) it follows.
// Conventions: pose = unit quaternion (x,y,z,w) + t, world->camera;
// residual r = proj - meas; small matrices column-major; H dx = b with b = J^T W r.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <set>
#include <vector>

// Optional OpenMP build (libba_oracle_omp.so, -fopenmp -DORACLE_OMP): the edge / landmark loops run on all host
// cores the way g2o's G2O_OPENMP build runs its linearisation, with atomic accumulation into shared blocks; the
// sparse factorisation stays sequential (as g2o's linear solvers are).  Used ONLY for bench.py's all-core CPU
// baseline -- every parity check uses the single-thread library, whose summation order is fixed.
#ifdef ORACLE_OMP
#include <omp.h>
#define ORC_PRAGMA(x) _Pragma(#x)
#define ORC_PARALLEL_FOR ORC_PRAGMA(omp parallel for schedule(static))
#define ORC_PARALLEL_FOR_DYN ORC_PRAGMA(omp parallel for schedule(static))   /* contiguous landmark ranges per thread: neighbouring landmarks share Hsc blocks */
#define ORC_PARALLEL_SUM(v) ORC_PRAGMA(omp parallel for schedule(static) reduction(+ : v))
#define ORC_ATOMIC ORC_PRAGMA(omp atomic)
#else
#define ORC_PARALLEL_FOR
#define ORC_PARALLEL_FOR_DYN
#define ORC_PARALLEL_SUM(v)
#define ORC_ATOMIC
#endif

namespace {

constexpr int PD = 6;  // pose block dimension     (src/constants.h:26)
constexpr int LD = 3;  // landmark block dimension (src/constants.h:27)

// ------------------------------------------------------------------------------------------------
// Per-edge math
// ------------------------------------------------------------------------------------------------

// q (x) v rotation via two cross products.  Follows src/cuda_block_solver.cu:238-260.
void rotate(const double* q, const double* v, double* out)
{
	double a[3] = { q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0] };
	for (int i = 0; i < 3; i++) a[i] += a[i];
	double b[3] = { q[1] * a[2] - q[2] * a[1], q[2] * a[0] - q[0] * a[2], q[0] * a[1] - q[1] * a[0] };
	for (int i = 0; i < 3; i++) out[i] = v[i] + q[3] * a[i] + b[i];
}

// World -> camera.  src/cuda_block_solver.cu:262-268.
void world_to_camera(const double* q, const double* t, const double* Xw, double* Xc)
{
	rotate(q, Xw, Xc);
	for (int i = 0; i < 3; i++) Xc[i] += t[i];
}

// Pinhole (mdim=2) / rectified stereo (mdim=3) projection.  src/cuda_block_solver.cu:275-290.
// cam = (fx, fy, cx, cy, bf).  No cheirality test (reference has none).
void camera_to_image(const double* Xc, const double* cam, int mdim, double* p)
{
	const double invZ = 1 / Xc[2];
	p[0] = cam[0] * invZ * Xc[0] + cam[2];
	p[1] = cam[1] * invZ * Xc[1] + cam[3];
	if (mdim == 3) p[2] = p[0] - cam[4] * invZ;
}

// Eigen-style quaternion -> rotation matrix (column-major 3x3).  src/cuda_block_solver.cu:292-321.
void quat_to_rot(const double* q, double* R)
{
	const double x = q[0], y = q[1], z = q[2], w = q[3];
	const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
	const double twx = tx * w, twy = ty * w, twz = tz * w;
	const double txx = tx * x, txy = ty * x, txz = tz * x;
	const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
#define RR(i, j) R[(j) * 3 + (i)]
	RR(0, 0) = 1 - (tyy + tzz); RR(0, 1) = txy - twz;       RR(0, 2) = txz + twy;
	RR(1, 0) = txy + twz;       RR(1, 1) = 1 - (txx + tzz); RR(1, 2) = tyz - twx;
	RR(2, 0) = txz - twy;       RR(2, 1) = tyz + twx;       RR(2, 2) = 1 - (txx + tyy);
#undef RR
}

// Analytic Jacobians of the (g2o-convention) error w.r.t. the left se(3) perturbation [rot; trans]
// (JP, mdim x 6) and the landmark (JL, mdim x 3); both column-major with mdim rows.
// Mono: src/cuda_block_solver.cu:329-366.  Stereo: :368-415.
void jacobians(const double* Xc, const double* q, const double* cam, int mdim, double* JP, double* JL)
{
	const double X = Xc[0], Y = Xc[1], Z = Xc[2];
	const double fu = cam[0], fv = cam[1], bf = cam[4];
	double R[9];
	quat_to_rot(q, R);
#define RR(i, j) R[(j) * 3 + (i)]
#define P(i, j) JP[(j) * mdim + (i)]
#define L(i, j) JL[(j) * mdim + (i)]
	if (mdim == 2)
	{
		const double invZ = 1 / Z;
		const double x = invZ * X, y = invZ * Y;
		const double fu_invZ = fu * invZ, fv_invZ = fv * invZ;
		for (int j = 0; j < 3; j++)
		{
			L(0, j) = -fu_invZ * (RR(0, j) - x * RR(2, j));
			L(1, j) = -fv_invZ * (RR(1, j) - y * RR(2, j));
		}
		P(0, 0) = +fu * x * y;      P(0, 1) = -fu * (1 + x * x); P(0, 2) = +fu * y;
		P(0, 3) = -fu_invZ;         P(0, 4) = 0;                 P(0, 5) = +fu_invZ * x;
		P(1, 0) = +fv * (1 + y * y); P(1, 1) = -fv * x * y;      P(1, 2) = -fv * x;
		P(1, 3) = 0;                P(1, 4) = -fv_invZ;          P(1, 5) = +fv_invZ * y;
	}
	else
	{
		const double invZ = 1 / Z, invZZ = invZ * invZ;
		for (int j = 0; j < 3; j++)
		{
			L(0, j) = -fu * RR(0, j) * invZ + fu * X * RR(2, j) * invZZ;
			L(1, j) = -fv * RR(1, j) * invZ + fv * Y * RR(2, j) * invZZ;
			L(2, j) = L(0, j) - bf * RR(2, j) * invZZ;
		}
		P(0, 0) = X * Y * invZZ * fu;        P(0, 1) = -(1 + (X * X * invZZ)) * fu; P(0, 2) = Y * invZ * fu;
		P(0, 3) = -1 * invZ * fu;            P(0, 4) = 0;                           P(0, 5) = X * invZZ * fu;
		P(1, 0) = (1 + Y * Y * invZZ) * fv;  P(1, 1) = -X * Y * invZZ * fv;         P(1, 2) = -X * invZ * fv;
		P(1, 3) = 0;                         P(1, 4) = -1 * invZ * fv;              P(1, 5) = Y * invZZ * fv;
		P(2, 0) = P(0, 0) - bf * Y * invZZ;  P(2, 1) = P(0, 1) + bf * X * invZZ;    P(2, 2) = P(0, 2);
		P(2, 3) = P(0, 3);                   P(2, 4) = 0;                           P(2, 5) = P(0, 5) - bf * invZZ;
	}
#undef RR
#undef P
#undef L
}

// Robust kernels rho(e) and rho'(e) on the squared, information-weighted error e.
// src/cuda_block_solver.cu:666-727 (NONE=0, HUBER=1, TUKEY=2).
double robustify(int type, double delta, double e)
{
	const double d2 = delta * delta;
	if (type == 1) return e <= d2 ? e : (2 * std::sqrt(e) * delta - d2);
	if (type == 2)
	{
		const double maxv = (1.0 / 3) * d2;
		const double u = 1 - e / d2;
		return e <= d2 ? maxv * (1 - u * u * u) : maxv;
	}
	return e;
}

double robust_weight(int type, double delta, double e)
{
	const double d2 = delta * delta;
	if (type == 1) return e <= d2 ? 1 : (delta / std::sqrt(e));
	if (type == 2)
	{
		const double u = 1 - e / d2;
		return e <= d2 ? u * u : 0;
	}
	return 1;
}

// Closed-form inverse of a symmetric 3x3 (adjugate / determinant, no pivoting).
// src/cuda_block_solver.cu:417-452.  Column-major in and out.
void sym3x3_inverse(const double* A, double* B)
{
	const double A00 = A[0], A01 = A[3], A11 = A[4], A02 = A[2], A12 = A[7], A22 = A[8];
	const double det = A00 * A11 * A22 + A01 * A12 * A02 + A02 * A01 * A12
		- A00 * A12 * A12 - A02 * A11 * A02 - A01 * A01 * A22;
	const double id = 1 / det;
	const double B00 = id * (A11 * A22 - A12 * A12);
	const double B01 = id * (A02 * A12 - A01 * A22);
	const double B11 = id * (A00 * A22 - A02 * A02);
	const double B02 = id * (A01 * A12 - A02 * A11);
	const double B12 = id * (A02 * A01 - A00 * A12);
	const double B22 = id * (A00 * A11 - A01 * A01);
	B[0] = B00; B[3] = B01; B[6] = B02;
	B[1] = B01; B[4] = B11; B[7] = B12;
	B[2] = B02; B[5] = B12; B[8] = B22;
}

// Rotation matrix (col-major) -> quaternion (x,y,z,w), Eigen's branches.
// src/cuda_block_solver.cu:492-521.
void rot_to_quat(const double* R, double* q)
{
#define RR(i, j) R[(j) * 3 + (i)]
	double t = RR(0, 0) + RR(1, 1) + RR(2, 2);
	if (t > 0)
	{
		t = std::sqrt(t + 1);
		q[3] = 0.5 * t;
		t = 0.5 / t;
		q[0] = (RR(2, 1) - RR(1, 2)) * t;
		q[1] = (RR(0, 2) - RR(2, 0)) * t;
		q[2] = (RR(1, 0) - RR(0, 1)) * t;
	}
	else
	{
		int i = 0;
		if (RR(1, 1) > RR(0, 0)) i = 1;
		if (RR(2, 2) > RR(i, i)) i = 2;
		const int j = (i + 1) % 3, k = (j + 1) % 3;
		t = std::sqrt(RR(i, i) - RR(j, j) - RR(k, k) + 1);
		q[i] = 0.5 * t;
		t = 0.5 / t;
		q[3] = (RR(k, j) - RR(j, k)) * t;
		q[j] = (RR(j, i) + RR(i, j)) * t;
		q[k] = (RR(k, i) + RR(i, k)) * t;
	}
#undef RR
}

// SE3 exponential of update = [omega; upsilon] -> (q_exp, t_exp).
// src/cuda_block_solver.cu:454-490 (skew1/skew2/addOmega) and :551-579 (updateExp).
void se3_exp(const double* upd, double* qe, double* te)
{
	const double wx = upd[0], wy = upd[1], wz = upd[2];
	const double theta = std::sqrt(wx * wx + wy * wy + wz * wz);
	// O1 = [w]x, O2 = [w]x^2, column-major
	const double O1[9] = { 0, wz, -wy, -wz, 0, wx, wy, -wx, 0 };
	const double xx = wx * wx, yy = wy * wy, zz = wz * wz, xy = wx * wy, yz = wy * wz, zx = wz * wx;
	const double O2[9] = { -yy - zz, xy, zx, xy, -zz - xx, yz, zx, yz, -xx - yy };
	double a1, a2, a3;
	if (theta < 0.00001) { a1 = 1.0; a2 = 0.5; a3 = 1.0 / 6; }
	else
	{
		a1 = std::sin(theta) / theta;
		a2 = (1 - std::cos(theta)) / (theta * theta);
		a3 = (theta - std::sin(theta)) / (theta * theta * theta);
	}
	double R[9], V[9];
	for (int k = 0; k < 9; k++)
	{
		const double I = (k % 4 == 0) ? 1.0 : 0.0;
		R[k] = I + a1 * O1[k] + a2 * O2[k];
		V[k] = I + a2 * O1[k] + a3 * O2[k];
	}
	rot_to_quat(R, qe);
	for (int i = 0; i < 3; i++) te[i] = V[i] * upd[3] + V[3 + i] * upd[4] + V[6 + i] * upd[5];
}

// Left-multiplicative pose update: T <- exp(upd) * T, quaternion renormalised with w >= 0.
// src/cuda_block_solver.cu:523-539 (quaternion product / normalise) and :581-592 (updatePose).
void pose_update(const double* upd, double* q, double* t)
{
	double qe[4], te[3], u[3];
	se3_exp(upd, qe, te);
	rotate(qe, t, u);
	for (int i = 0; i < 3; i++) t[i] = te[i] + u[i];
	const double* a = qe; const double* b = q;
	double c[4];
	c[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
	c[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
	c[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
	c[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
	double invn = 1 / std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3]);
	if (c[3] < 0) invn = -invn;
	for (int i = 0; i < 4; i++) q[i] = invn * c[i];
}

// ------------------------------------------------------------------------------------------------
// Exact sparse block Cholesky of the reduced pose system (stands in for cuSOLVER csrchol + METIS,
// src/cuda_linear_solver.cpp:147-232,347).  6x6 blocks, min-degree ordering on the block graph,
// right-looking numeric factorisation.  Pattern input: upper-triangular BSR (row i, cols j >= i).
// ------------------------------------------------------------------------------------------------
struct BlockCholesky
{
	int n = 0;                           // block rows
	std::vector<int> perm, iperm;        // perm[k] = original index eliminated k-th
	std::vector<std::vector<int>> rows;  // rows[k]: sorted permuted row ids (> k) of L column k
	std::vector<int> colPtr;             // block offset of column k (diag first, then rows[k])
	std::vector<double> L;               // 36 doubles per block, column-major

	void analyze(int nb, const int* rowPtr, const int* colInd)
	{
		n = nb;
		std::vector<std::set<int>> adj(n);
		for (int i = 0; i < n; i++)
			for (int k = rowPtr[i]; k < rowPtr[i + 1]; k++)
			{
				const int j = colInd[k];
				if (j != i) { adj[i].insert(j); adj[j].insert(i); }
			}
		// greedy minimum (exact external) degree with explicit elimination graph
		perm.assign(n, 0); iperm.assign(n, 0);
		std::vector<char> done(n, 0);
		std::set<std::pair<int, int>> heap;
		for (int i = 0; i < n; i++) heap.insert({ (int)adj[i].size(), i });
		std::vector<std::vector<int>> structOrig(n);
		for (int k = 0; k < n; k++)
		{
			const int v = heap.begin()->second;
			heap.erase(heap.begin());
			done[v] = 1; perm[k] = v; iperm[v] = k;
			std::vector<int> nb_(adj[v].begin(), adj[v].end());
			structOrig[k] = nb_;
			for (int a : nb_)
			{
				heap.erase({ (int)adj[a].size(), a });
				adj[a].erase(v);
			}
			for (size_t x = 0; x < nb_.size(); x++)
				for (size_t y = x + 1; y < nb_.size(); y++)
				{
					adj[nb_[x]].insert(nb_[y]);
					adj[nb_[y]].insert(nb_[x]);
				}
			for (int a : nb_) heap.insert({ (int)adj[a].size(), a });
			adj[v].clear();
		}
		rows.assign(n, {});
		colPtr.assign(n + 1, 0);
		for (int k = 0; k < n; k++)
		{
			for (int a : structOrig[k]) rows[k].push_back(iperm[a]);
			std::sort(rows[k].begin(), rows[k].end());
			colPtr[k + 1] = colPtr[k] + 1 + (int)rows[k].size();
		}
		L.assign((size_t)colPtr[n] * 36, 0.0);
	}

	double* block(int col, int row)  // row >= col, permuted ids; must exist
	{
		if (row == col) return &L[(size_t)colPtr[col] * 36];
		const auto& r = rows[col];
		const int pos = int(std::lower_bound(r.begin(), r.end(), row) - r.begin());
		return &L[(size_t)(colPtr[col] + 1 + pos) * 36];
	}

	// values: upper-triangular BSR blocks (col-major 6x6) in the analysed pattern. Returns false if not SPD.
	bool factorize(const int* rowPtr, const int* colInd, const double* values)
	{
		std::fill(L.begin(), L.end(), 0.0);
		for (int i = 0; i < n; i++)
			for (int k = rowPtr[i]; k < rowPtr[i + 1]; k++)
			{
				const int j = colInd[k];
				const int pi = iperm[i], pj = iperm[j];
				const double* src = values + (size_t)k * 36;
				// A(i,j) block with j >= i (upper). Lower-triangular storage needs A(max,min).
				if (pi == pj)
				{
					double* dst = block(pi, pi);
					for (int e = 0; e < 36; e++) dst[e] = src[e];
				}
				else if (pi > pj)
				{
					// store A(pi,pj) = A(i,j) as is (row pi, col pj)
					double* dst = block(pj, pi);
					for (int e = 0; e < 36; e++) dst[e] = src[e];
				}
				else
				{
					// need A(pj,pi) = A(i,j)^T
					double* dst = block(pi, pj);
					for (int c = 0; c < 6; c++) for (int r = 0; r < 6; r++) dst[c * 6 + r] = src[r * 6 + c];
				}
			}
		for (int k = 0; k < n; k++)
		{
			double* D = block(k, k);
			// dense 6x6 Cholesky in place (lower), column-major
			for (int c = 0; c < 6; c++)
			{
				double d = D[c * 6 + c];
				for (int p = 0; p < c; p++) d -= D[p * 6 + c] * D[p * 6 + c];
				if (!(d > 0) || !std::isfinite(d)) return false;
				d = std::sqrt(d);
				D[c * 6 + c] = d;
				for (int r = c + 1; r < 6; r++)
				{
					double s = D[c * 6 + r];
					for (int p = 0; p < c; p++) s -= D[p * 6 + r] * D[p * 6 + c];
					D[c * 6 + r] = s / d;
				}
				for (int r = 0; r < c; r++) D[c * 6 + r] = 0;  // zero strict upper
			}
			const auto& rk = rows[k];
			const int m = (int)rk.size();
			double* Lk = &L[(size_t)(colPtr[k] + 1) * 36];
			// L(i,k) = A(i,k) * D^-T  (solve X D^T = A row-wise)
			for (int b = 0; b < m; b++)
			{
				double* X = Lk + (size_t)b * 36;
				for (int c = 0; c < 6; c++)
					for (int r = 0; r < 6; r++)
					{
						double s = X[c * 6 + r];
						for (int p = 0; p < c; p++) s -= X[p * 6 + r] * D[p * 6 + c];
						X[c * 6 + r] = s / D[c * 6 + c];
					}
			}
			// trailing update: A(i,j) -= L(i,k) L(j,k)^T for i >= j in rows[k]
			for (int bj = 0; bj < m; bj++)
			{
				const int j = rk[bj];
				const double* Lj = Lk + (size_t)bj * 36;
				for (int bi = bj; bi < m; bi++)
				{
					const int i = rk[bi];
					const double* Li = Lk + (size_t)bi * 36;
					double* T = block(j, i);
					for (int c = 0; c < 6; c++)
						for (int r = 0; r < 6; r++)
						{
							double s = 0;
							for (int p = 0; p < 6; p++) s += Li[p * 6 + r] * Lj[p * 6 + c];
							T[c * 6 + r] -= s;
						}
				}
			}
		}
		return true;
	}

	void solve(const double* b, double* x) const
	{
		std::vector<double> y((size_t)n * 6);
		for (int k = 0; k < n; k++) for (int r = 0; r < 6; r++) y[(size_t)k * 6 + r] = b[(size_t)perm[k] * 6 + r];
		// forward: L y = b
		for (int k = 0; k < n; k++)
		{
			const double* D = &L[(size_t)colPtr[k] * 36];
			double* yk = &y[(size_t)k * 6];
			for (int r = 0; r < 6; r++)
			{
				double s = yk[r];
				for (int p = 0; p < r; p++) s -= D[p * 6 + r] * yk[p];
				yk[r] = s / D[r * 6 + r];
			}
			const auto& rk = rows[k];
			for (size_t bi = 0; bi < rk.size(); bi++)
			{
				const double* Li = &L[(size_t)(colPtr[k] + 1 + bi) * 36];
				double* yi = &y[(size_t)rk[bi] * 6];
				for (int r = 0; r < 6; r++)
				{
					double s = 0;
					for (int p = 0; p < 6; p++) s += Li[p * 6 + r] * yk[p];
					yi[r] -= s;
				}
			}
		}
		// backward: L^T x = y
		for (int k = n - 1; k >= 0; k--)
		{
			const double* D = &L[(size_t)colPtr[k] * 36];
			double* yk = &y[(size_t)k * 6];
			const auto& rk = rows[k];
			for (size_t bi = 0; bi < rk.size(); bi++)
			{
				const double* Li = &L[(size_t)(colPtr[k] + 1 + bi) * 36];
				const double* yi = &y[(size_t)rk[bi] * 6];
				for (int c = 0; c < 6; c++)
				{
					double s = 0;
					for (int p = 0; p < 6; p++) s += Li[c * 6 + p] * yi[p];
					yk[c] -= s;
				}
			}
			for (int r = 5; r >= 0; r--)
			{
				double s = yk[r];
				for (int p = r + 1; p < 6; p++) s -= D[r * 6 + p] * yk[p];
				yk[r] = s / D[r * 6 + r];
			}
		}
		for (int k = 0; k < n; k++) for (int r = 0; r < 6; r++) x[(size_t)perm[k] * 6 + r] = y[(size_t)k * 6 + r];
	}
};

// ------------------------------------------------------------------------------------------------
// Problem = what CudaBlockSolver holds after initialize() (src/cuda_bundle_adjustment.cpp:115-261):
// poses [free | fixed], landmarks [free | fixed], active edges with (iP, iL) solver indices.
// ------------------------------------------------------------------------------------------------
struct Problem
{
	int Pt = 0, Pf = 0, Lt = 0, Lf = 0, E = 0;
	std::vector<double> q, t, cam, Xw;           // 4Pt, 3Pt, 5Pt, 3Lt
	std::vector<double> qBak, tBak, XwBak;       // push()/pop() (:502-510)
	std::vector<int> eP, eL;
	std::vector<uint8_t> eDim;
	std::vector<double> meas, omega;             // 3E (mono uses first 2), E
	int rkType[2] = { 0, 0 };
	double rkDelta[2] = { 0, 0 };

	// linear system
	std::vector<double> err, Xc;                 // 3E each (stored like d_errors / d_Xcs)
	std::vector<double> Hpp, bp, Hll, bl, Hpl;   // 36Pf, 6Pf, 9Lf, 3Lf, 18E (zero when a side is fixed)
	std::vector<double> HppDiagBak, HllDiagBak;  // addLambda backups (:906-931)
	std::vector<double> invHll, HplInvHll;       // 9Lf, 18E
	std::vector<double> bsc, xp, xl;             // 6Pf, 6Pf, 3Lf
	// Hsc pattern: upper-triangular BSR with every diagonal block present (Appendix B #3 fixed)
	std::vector<int> hscRowPtr, hscColInd;
	std::vector<double> hscVal;
	std::vector<std::vector<int>> lmEdges;       // per free landmark: edges with a free pose, sorted by pose
#ifdef ORACLE_OMP
	std::vector<std::vector<int>> lmAll;         // per landmark: all its edges, ascending edge index (OpenMP build only)
	std::vector<double> ompHpp, ompBp;           // per-thread pose-side accumulators
#endif
	bool structureBuilt = false;
	BlockCholesky chol;
	double lambda = 0;

	int rkOf(int e) const { return eDim[e] == 2 ? 0 : 1; }

	int hscFind(int i, int j) const  // block (i,j), j >= i
	{
		const int* b = &hscColInd[hscRowPtr[i]];
		const int* e = &hscColInd[hscRowPtr[i + 1]];
		return int(std::lower_bound(b, e, j) - &hscColInd[0]);
	}

	// Symbolic structure of Hsc from landmark co-visibility.
	// Follows src/sparse_block_matrix.cpp:55-133 (without the dense P x P byte map).
	void buildStructure()
	{
		lmEdges.assign(Lf, {});
		for (int e = 0; e < E; e++)
			if (eP[e] < Pf && eL[e] < Lf) lmEdges[eL[e]].push_back(e);
#ifdef ORACLE_OMP
		lmAll.assign(Lt, {});
		for (int e = 0; e < E; e++) lmAll[eL[e]].push_back(e);
#endif
		std::vector<std::set<int>> cols(Pf);
		for (int i = 0; i < Pf; i++) cols[i].insert(i);
		for (int l = 0; l < Lf; l++)
		{
			auto& v = lmEdges[l];
			std::stable_sort(v.begin(), v.end(), [&](int a, int b) { return eP[a] < eP[b]; });
			for (size_t a = 0; a < v.size(); a++)
				for (size_t b = a; b < v.size(); b++)
					cols[eP[v[a]]].insert(eP[v[b]]);
		}
		hscRowPtr.assign(Pf + 1, 0);
		hscColInd.clear();
		for (int i = 0; i < Pf; i++)
		{
			for (int j : cols[i]) hscColInd.push_back(j);
			hscRowPtr[i + 1] = (int)hscColInd.size();
		}
		hscVal.assign(hscColInd.size() * 36, 0.0);
		if (Pf > 0) chol.analyze(Pf, hscRowPtr.data(), hscColInd.data());
		Hpp.assign((size_t)Pf * 36, 0); bp.assign((size_t)Pf * 6, 0);
		Hll.assign((size_t)Lf * 9, 0); bl.assign((size_t)Lf * 3, 0);
		Hpl.assign((size_t)E * 18, 0); HplInvHll.assign((size_t)E * 18, 0);
		invHll.assign((size_t)Lf * 9, 0);
		bsc.assign((size_t)Pf * 6, 0); xp.assign((size_t)Pf * 6, 0); xl.assign((size_t)Lf * 3, 0);
		err.assign((size_t)E * 3, 0); Xc.assign((size_t)E * 3, 0);
		structureBuilt = true;
	}

	// Residuals, camera-frame points and total robust chi2.
	// Follows computeActiveErrorsKernel, src/cuda_block_solver.cu:733-786.
	double computeErrors()
	{
		if (!structureBuilt) buildStructure();
		double chi = 0;
		ORC_PARALLEL_SUM(chi)
		for (int e = 0; e < E; e++)
		{
			const int iP = eP[e], iL = eL[e], md = eDim[e];
			double p[3] = { 0, 0, 0 };
			world_to_camera(&q[4 * iP], &t[3 * iP], &Xw[3 * iL], &Xc[3 * e]);
			camera_to_image(&Xc[3 * e], &cam[5 * iP], md, p);
			double s = 0;
			for (int i = 0; i < 3; i++)
			{
				err[3 * e + i] = i < md ? p[i] - meas[3 * e + i] : 0.0;
				s += err[3 * e + i] * err[3 * e + i];
			}
			chi += robustify(rkType[rkOf(e)], rkDelta[rkOf(e)], omega[e] * s);
		}
		return chi;
	}

	// Non-robust per-edge chi2.  computeChiSquaresKernel, src/cuda_block_solver.cu:841-875.
	void chiSquares(double* out) const
	{
		for (int e = 0; e < E; e++)
		{
			const int iP = eP[e], iL = eL[e], md = eDim[e];
			double xc[3], p[3] = { 0, 0, 0 };
			world_to_camera(&q[4 * iP], &t[3 * iP], &Xw[3 * iL], xc);
			camera_to_image(xc, &cam[5 * iP], md, p);
			double s = 0;
			for (int i = 0; i < md; i++) s += (p[i] - meas[3 * e + i]) * (p[i] - meas[3 * e + i]);
			out[e] = omega[e] * s;
		}
	}

	// Hpp/bp/Hll/bl/Hpl from the stored errors and Xcs (IRLS weight w' = w * rho'(w |r|^2)).
	// Follows CudaBlockSolver::buildSystem (src/cuda_bundle_adjustment.cpp:384-410) and
	// constructQuadraticFormKernel (src/cuda_block_solver.cu:788-839).
	void buildSystem()
	{
		std::fill(Hpp.begin(), Hpp.end(), 0.0); std::fill(bp.begin(), bp.end(), 0.0);
		std::fill(Hll.begin(), Hll.end(), 0.0); std::fill(bl.begin(), bl.end(), 0.0);
#ifndef ORACLE_OMP
		std::fill(Hpl.begin(), Hpl.end(), 0.0);
		for (int e = 0; e < E; e++) accumulateEdge(e, Hpp.data(), bp.data());
#else
		// All-core variant: landmarks are spread over the threads (every landmark's Hll / bl / Hpl blocks have a single
		// writer, edges of a landmark are taken in ascending edge order like the sequential loop), the pose-side sums go
		// to per-thread copies of Hpp / bp that are added up afterwards.  Hpl blocks of edges with a fixed end are never
		// written and stay zero from buildStructure().
		const int T = omp_get_max_threads();
		const size_t nH = Hpp.size(), nb = bp.size();
		if (ompHpp.size() != (size_t)T * nH) { ompHpp.assign((size_t)T * nH, 0.0); ompBp.assign((size_t)T * nb, 0.0); }
		#pragma omp parallel
		{
			const int tid = omp_get_thread_num();
			double* H = &ompHpp[(size_t)tid * nH]; double* b = &ompBp[(size_t)tid * nb];
			std::fill(H, H + nH, 0.0); std::fill(b, b + nb, 0.0);
			#pragma omp for schedule(dynamic, 64)
			for (int l = 0; l < Lt; l++)
				for (int e : lmAll[l]) accumulateEdge(e, H, b);
			#pragma omp for schedule(static)
			for (long i = 0; i < (long)nH; i++) { double s = 0; for (int t2 = 0; t2 < T; t2++) s += ompHpp[(size_t)t2 * nH + i]; Hpp[i] = s; }
			#pragma omp for schedule(static)
			for (long i = 0; i < (long)nb; i++) { double s = 0; for (int t2 = 0; t2 < T; t2++) s += ompBp[(size_t)t2 * nb + i]; bp[i] = s; }
		}
#endif
	}

	// one edge's contribution to the normal equations (body of constructQuadraticFormKernel, :788-839); the pose-side
	// sums go to the arrays passed in (the shared ones, or a thread's private copy in the OpenMP build)
	void accumulateEdge(int e, double* HppOut, double* bpOut)
	{
		const int iP = eP[e], iL = eL[e], md = eDim[e];
		const double* r = &err[3 * e];
		const double ee = (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) * omega[e];
		const double w = omega[e] * robust_weight(rkType[rkOf(e)], rkDelta[rkOf(e)], ee);
		double JP[18], JL[9];
		jacobians(&Xc[3 * e], &q[4 * iP], &cam[5 * iP], md, JP, JL);
		const bool freeP = iP < Pf, freeL = iL < Lf;
		if (freeP)
		{
			double* H = &HppOut[(size_t)iP * 36];
			for (int c = 0; c < 6; c++)
			{
				for (int rr = 0; rr < 6; rr++)
				{
					double s = 0;
					for (int m = 0; m < md; m++) s += JP[rr * md + m] * JP[c * md + m];
					H[c * 6 + rr] += w * s;
				}
				double s = 0;
				for (int m = 0; m < md; m++) s += JP[c * md + m] * r[m];
				bpOut[(size_t)iP * 6 + c] += w * s;
			}
		}
		if (freeL)
		{
			double* H = &Hll[(size_t)iL * 9];
			for (int c = 0; c < 3; c++)
			{
				for (int rr = 0; rr < 3; rr++)
				{
					double s = 0;
					for (int m = 0; m < md; m++) s += JL[rr * md + m] * JL[c * md + m];
					H[c * 3 + rr] += w * s;
				}
				double s = 0;
				for (int m = 0; m < md; m++) s += JL[c * md + m] * r[m];
				bl[(size_t)iL * 3 + c] += w * s;
			}
		}
		if (freeP && freeL)
		{
			double* H = &Hpl[(size_t)e * 18];  // 6x3 col-major
			for (int c = 0; c < 3; c++)
				for (int rr = 0; rr < 6; rr++)
				{
					double s = 0;
					for (int m = 0; m < md; m++) s += JP[rr * md + m] * JL[c * md + m];
					H[c * 6 + rr] = w * s;
				}
		}
	}

	// max over the diagonals of Hpp and Hll, starting from 0.
	// src/cuda_bundle_adjustment.cpp:412-418, src/cuda_block_solver.cu:877-904.
	double maxDiagonal() const
	{
		double m = 0;
		for (int i = 0; i < Pf; i++) for (int k = 0; k < 6; k++) m = std::max(m, Hpp[(size_t)i * 36 + k * 7]);
		for (int i = 0; i < Lf; i++) for (int k = 0; k < 3; k++) m = std::max(m, Hll[(size_t)i * 9 + k * 4]);
		return m;
	}

	// src/cuda_bundle_adjustment.cpp:420-430, src/cuda_block_solver.cu:906-931.
	void setLambda(double lam)
	{
		lambda = lam;
		HppDiagBak.resize((size_t)Pf * 6); HllDiagBak.resize((size_t)Lf * 3);
		for (int i = 0; i < Pf; i++) for (int k = 0; k < 6; k++)
		{
			HppDiagBak[(size_t)i * 6 + k] = Hpp[(size_t)i * 36 + k * 7];
			Hpp[(size_t)i * 36 + k * 7] += lam;
		}
		for (int i = 0; i < Lf; i++) for (int k = 0; k < 3; k++)
		{
			HllDiagBak[(size_t)i * 3 + k] = Hll[(size_t)i * 9 + k * 4];
			Hll[(size_t)i * 9 + k * 4] += lam;
		}
	}

	void restoreDiagonal()
	{
		for (int i = 0; i < Pf; i++) for (int k = 0; k < 6; k++) Hpp[(size_t)i * 36 + k * 7] = HppDiagBak[(size_t)i * 6 + k];
		for (int i = 0; i < Lf; i++) for (int k = 0; k < 3; k++) Hll[(size_t)i * 9 + k * 4] = HllDiagBak[(size_t)i * 3 + k];
	}

	// Schur complement: invHll, Hpl*invHll, bsc = bp - sum Hpl invHll bl,
	// Hsc = Hpp - sum_l sum_{i<=j} (Hpl_i invHll) Hpl_j^T (upper-triangular BSR).
	// Follows computeBschureKernel / initializeHschurKernel / computeHschureKernel,
	// src/cuda_block_solver.cu:933-977.
	void schur()
	{
		bsc = bp;
		std::fill(hscVal.begin(), hscVal.end(), 0.0);
		for (int i = 0; i < Pf; i++)
			std::memcpy(&hscVal[(size_t)hscFind(i, i) * 36], &Hpp[(size_t)i * 36], 36 * sizeof(double));
		ORC_PARALLEL_FOR_DYN
		for (int l = 0; l < Lf; l++)
		{
			double* iH = &invHll[(size_t)l * 9];
			sym3x3_inverse(&Hll[(size_t)l * 9], iH);
			const auto& v = lmEdges[l];
			for (int e : v)
			{
				const double* A = &Hpl[(size_t)e * 18];
				double* W = &HplInvHll[(size_t)e * 18];
				for (int c = 0; c < 3; c++)
					for (int r = 0; r < 6; r++)
						W[c * 6 + r] = A[0 * 6 + r] * iH[c * 3 + 0] + A[1 * 6 + r] * iH[c * 3 + 1] + A[2 * 6 + r] * iH[c * 3 + 2];
				for (int r = 0; r < 6; r++)
				{
					const double d = W[0 * 6 + r] * bl[(size_t)l * 3 + 0] + W[1 * 6 + r] * bl[(size_t)l * 3 + 1] + W[2 * 6 + r] * bl[(size_t)l * 3 + 2];
					ORC_ATOMIC
					bsc[(size_t)eP[e] * 6 + r] -= d;
				}
			}
			for (size_t a = 0; a < v.size(); a++)
				for (size_t b = a; b < v.size(); b++)
				{
					const double* W = &HplInvHll[(size_t)v[a] * 18];
					const double* B = &Hpl[(size_t)v[b] * 18];
					double* T = &hscVal[(size_t)hscFind(eP[v[a]], eP[v[b]]) * 36];
					for (int c = 0; c < 6; c++)
						for (int r = 0; r < 6; r++)
						{
							const double d = W[0 * 6 + r] * B[0 * 6 + c] + W[1 * 6 + r] * B[1 * 6 + c] + W[2 * 6 + r] * B[2 * 6 + c];
							ORC_ATOMIC
							T[c * 6 + r] -= d;
						}
				}
		}
	}

	// Back-substitution xl = invHll (bl - Hpl^T xp).  schurComplementPostKernel, :1029-1043.
	void backSubstitute()
	{
		ORC_PARALLEL_FOR_DYN
		for (int l = 0; l < Lf; l++)
		{
			double cl[3] = { bl[(size_t)l * 3], bl[(size_t)l * 3 + 1], bl[(size_t)l * 3 + 2] };
			for (int e : lmEdges[l])
			{
				const double* A = &Hpl[(size_t)e * 18];
				const double* x = &xp[(size_t)eP[e] * 6];
				for (int c = 0; c < 3; c++)
				{
					double s = 0;
					for (int r = 0; r < 6; r++) s += A[c * 6 + r] * x[r];
					cl[c] -= s;
				}
			}
			const double* iH = &invHll[(size_t)l * 9];
			for (int r = 0; r < 3; r++)
				xl[(size_t)l * 3 + r] = iH[0 * 3 + r] * cl[0] + iH[1 * 3 + r] * cl[1] + iH[2 * 3 + r] * cl[2];
		}
	}

	// CudaBlockSolver::solve, src/cuda_bundle_adjustment.cpp:432-481 (three modes).
	bool solve()
	{
		if (Pf > 0 && Lf > 0)
		{
			schur();
			if (!chol.factorize(hscRowPtr.data(), hscColInd.data(), hscVal.data())) return false;
			chol.solve(bsc.data(), xp.data());
			backSubstitute();
		}
		else if (Pf > 0)
		{
			// pose-only: independent 6x6 SPD systems (reference: 3+3 Schur split, :617-664; same solution)
			for (int i = 0; i < Pf; i++)
			{
				int rp[2] = { 0, 1 }, ci[1] = { 0 };
				BlockCholesky c; c.analyze(1, rp, ci);
				if (!c.factorize(rp, ci, &Hpp[(size_t)i * 36])) return false;
				c.solve(&bp[(size_t)i * 6], &xp[(size_t)i * 6]);
			}
		}
		else
		{
			// landmark-only: 3x3 closed form, :610-615
			for (int l = 0; l < Lf; l++)
			{
				double iH[9];
				sym3x3_inverse(&Hll[(size_t)l * 9], iH);
				for (int r = 0; r < 3; r++)
					xl[(size_t)l * 3 + r] = iH[r] * bl[(size_t)l * 3] + iH[3 + r] * bl[(size_t)l * 3 + 1] + iH[6 + r] * bl[(size_t)l * 3 + 2];
			}
		}
		return true;
	}

	// updatePosesKernel / updateLandmarksKernel, src/cuda_block_solver.cu:1045-1068.
	void update()
	{
		for (int i = 0; i < Pf; i++) pose_update(&xp[(size_t)i * 6], &q[(size_t)i * 4], &t[(size_t)i * 3]);
		ORC_PARALLEL_FOR
		for (int i = 0; i < Lf * 3; i++) Xw[i] += xl[i];
	}

	// sum x (lambda x + b) over [xp; xl].  computeScaleKernel, :1070-1091.
	double computeScale(double lam) const
	{
		double s = 0;
		for (size_t i = 0; i < xp.size(); i++) s += xp[i] * (lam * xp[i] + bp[i]);
		ORC_PARALLEL_SUM(s)
		for (long i = 0; i < (long)xl.size(); i++) s += xl[i] * (lam * xl[i] + bl[i]);
		return s;
	}

	void push() { qBak = q; tBak = t; XwBak = Xw; }
	void pop() { q = qBak; t = tBak; Xw = XwBak; }

	// Levenberg-Marquardt driver.  Follows CudaBundleAdjustmentImpl::optimize,
	// src/cuda_bundle_adjustment.cpp:793-857 (tau = 1e-5, maxq = 10, g2o's rho / lambda rules).
	int optimize(int niter, double* chi2Out, double* lambdaOut, int* trialsOut)
	{
		const int maxq = 10;
		const double tau = 1e-5;
		double nu = 2, lam = 0, F = 0;
		int done = 0;
		for (int it = 0; it < niter; it++)
		{
			if (it == 0 && !structureBuilt) buildStructure();
			F = computeErrors();
			buildSystem();
			if (it == 0) lam = tau * maxDiagonal();
			int qn = 0;
			double rho = -1;
			for (; qn < maxq && rho < 0; qn++)
			{
				push();
				setLambda(lam);
				const bool ok = solve();
				update();
				const double Fhat = computeErrors();
				const double scale = computeScale(lam) + 1e-3;
				rho = ok ? (F - Fhat) / scale : -1;
				if (rho > 0)
				{
					const double a = 1 - std::pow(2 * rho - 1, 3);
					lam *= std::max(1. / 3, std::min(a, 2. / 3));
					nu = 2;
					F = Fhat;
					break;
				}
				else
				{
					lam *= nu;
					nu *= 2;
					restoreDiagonal();
					pop();
				}
			}
			if (chi2Out) chi2Out[it] = F;
			if (lambdaOut) lambdaOut[it] = lam;
			if (trialsOut) trialsOut[it] = std::min(qn + 1, maxq);
			done = it + 1;
			if (qn == maxq || rho <= 0 || !std::isfinite(lam)) break;
		}
		return done;
	}
};

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI (ctypes-friendly).  All matrices column-major, quaternions (x,y,z,w).
// ------------------------------------------------------------------------------------------------
extern "C" {

// threads the OpenMP build will use (1 in the plain build); orc_set_threads(0) = all cores
int orc_max_threads(void)
{
#ifdef ORACLE_OMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}
int orc_set_threads(int n)
{
#ifdef ORACLE_OMP
	omp_set_num_threads(n > 0 ? n : omp_get_num_procs());
	return omp_get_max_threads();
#else
	(void)n;
	return 1;
#endif
}

void orc_project(const double* q, const double* t, const double* cam, const double* Xw, int mdim, double* Xc, double* proj)
{
	world_to_camera(q, t, Xw, Xc);
	camera_to_image(Xc, cam, mdim, proj);
}
void orc_rotate(const double* q, const double* v, double* out) { rotate(q, v, out); }
void orc_quat_to_rot(const double* q, double* R) { quat_to_rot(q, R); }
void orc_rot_to_quat(const double* R, double* q) { rot_to_quat(R, q); }
void orc_jacobians(const double* Xc, const double* q, const double* cam, int mdim, double* JP, double* JL) { jacobians(Xc, q, cam, mdim, JP, JL); }
double orc_robustify(int type, double delta, double e) { return robustify(type, delta, e); }
double orc_robust_weight(int type, double delta, double e) { return robust_weight(type, delta, e); }
void orc_sym3x3_inverse(const double* A, double* B) { sym3x3_inverse(A, B); }
void orc_se3_exp(const double* upd, double* q, double* t) { se3_exp(upd, q, t); }
void orc_pose_update(const double* upd, double* q, double* t) { pose_update(upd, q, t); }

// Solve a symmetric positive definite block system given as upper-triangular BSR (6x6 blocks).
int orc_block_cholesky_solve(int nb, const int* rowPtr, const int* colInd, const double* values, const double* b, double* x)
{
	BlockCholesky c;
	c.analyze(nb, rowPtr, colInd);
	if (!c.factorize(rowPtr, colInd, values)) return 1;
	c.solve(b, x);
	return 0;
}

void* orc_create(int Pt, int Pf, int Lt, int Lf, const double* q, const double* t, const double* cam, const double* Xw,
	int E, const int* eP, const int* eL, const uint8_t* eDim, const double* meas3, const double* omega)
{
	Problem* p = new Problem;
	p->Pt = Pt; p->Pf = Pf; p->Lt = Lt; p->Lf = Lf; p->E = E;
	p->q.assign(q, q + 4 * (size_t)Pt); p->t.assign(t, t + 3 * (size_t)Pt); p->cam.assign(cam, cam + 5 * (size_t)Pt);
	p->Xw.assign(Xw, Xw + 3 * (size_t)Lt);
	p->eP.assign(eP, eP + E); p->eL.assign(eL, eL + E); p->eDim.assign(eDim, eDim + E);
	p->meas.assign(meas3, meas3 + 3 * (size_t)E); p->omega.assign(omega, omega + E);
	return p;
}
void orc_destroy(void* h) { delete (Problem*)h; }
void orc_set_robust_kernel(void* h, int edgeType, int kind, double delta)
{
	Problem* p = (Problem*)h;
	p->rkType[edgeType] = kind; p->rkDelta[edgeType] = delta;
}
void orc_build_structure(void* h) { ((Problem*)h)->buildStructure(); }
double orc_compute_errors(void* h) { return ((Problem*)h)->computeErrors(); }
void orc_build_system(void* h) { ((Problem*)h)->buildSystem(); }
double orc_max_diagonal(void* h) { return ((Problem*)h)->maxDiagonal(); }
void orc_set_lambda(void* h, double lam) { ((Problem*)h)->setLambda(lam); }
void orc_restore_diagonal(void* h) { ((Problem*)h)->restoreDiagonal(); }
int orc_solve(void* h) { return ((Problem*)h)->solve() ? 1 : 0; }
void orc_schur(void* h) { ((Problem*)h)->schur(); }
// split solve (for the landmark-partitioned multi-process driver test): reduced system only / back-substitution only
int orc_solve_reduced(void* h)
{
	Problem* p = (Problem*)h;
	if (p->Pf == 0) return 1;
	if (!p->chol.factorize(p->hscRowPtr.data(), p->hscColInd.data(), p->hscVal.data())) return 0;
	p->chol.solve(p->bsc.data(), p->xp.data());
	return 1;
}
void orc_back_substitute(void* h) { ((Problem*)h)->backSubstitute(); }
void orc_update(void* h) { ((Problem*)h)->update(); }
double orc_compute_scale(void* h, double lam) { return ((Problem*)h)->computeScale(lam); }
void orc_push(void* h) { ((Problem*)h)->push(); }
void orc_pop(void* h) { ((Problem*)h)->pop(); }
int orc_optimize(void* h, int niter, double* chi2, double* lambdas, int* trials) { return ((Problem*)h)->optimize(niter, chi2, lambdas, trials); }
void orc_chi_squares(void* h, double* out) { ((Problem*)h)->chiSquares(out); }

void orc_get_state(void* h, double* q, double* t, double* Xw)
{
	Problem* p = (Problem*)h;
	if (q) std::memcpy(q, p->q.data(), p->q.size() * 8);
	if (t) std::memcpy(t, p->t.data(), p->t.size() * 8);
	if (Xw) std::memcpy(Xw, p->Xw.data(), p->Xw.size() * 8);
}
void orc_set_state(void* h, const double* q, const double* t, const double* Xw)
{
	Problem* p = (Problem*)h;
	if (q) std::memcpy(p->q.data(), q, p->q.size() * 8);
	if (t) std::memcpy(p->t.data(), t, p->t.size() * 8);
	if (Xw) std::memcpy(p->Xw.data(), Xw, p->Xw.size() * 8);
}
int orc_hsc_nblocks(void* h) { return (int)((Problem*)h)->hscColInd.size(); }
void orc_get_hsc(void* h, int* rowPtr, int* colInd, double* values)
{
	Problem* p = (Problem*)h;
	std::memcpy(rowPtr, p->hscRowPtr.data(), p->hscRowPtr.size() * 4);
	std::memcpy(colInd, p->hscColInd.data(), p->hscColInd.size() * 4);
	if (values) std::memcpy(values, p->hscVal.data(), p->hscVal.size() * 8);
}
// which: 0 Hpp(36Pf) 1 bp(6Pf) 2 Hll(9Lf) 3 bl(3Lf) 4 Hpl(18E) 5 bsc(6Pf) 6 xp(6Pf) 7 xl(3Lf) 8 invHll(9Lf) 9 err(3E) 10 Xc(3E)
long orc_get_array(void* h, int which, double* out)
{
	Problem* p = (Problem*)h;
	const std::vector<double>* v = nullptr;
	switch (which)
	{
	case 0: v = &p->Hpp; break; case 1: v = &p->bp; break; case 2: v = &p->Hll; break; case 3: v = &p->bl; break;
	case 4: v = &p->Hpl; break; case 5: v = &p->bsc; break; case 6: v = &p->xp; break; case 7: v = &p->xl; break;
	case 8: v = &p->invHll; break; case 9: v = &p->err; break; case 10: v = &p->Xc; break;
	default: return -1;
	}
	if (out) std::memcpy(out, v->data(), v->size() * 8);
	return (long)v->size();
}

// overwrite an internal array (same ids as orc_get_array, plus 11 = Hsc values) -- used to inject all-reduced sums
long orc_set_array(void* h, int which, const double* in)
{
	Problem* p = (Problem*)h;
	std::vector<double>* v = nullptr;
	switch (which)
	{
	case 0: v = &p->Hpp; break; case 1: v = &p->bp; break; case 2: v = &p->Hll; break; case 3: v = &p->bl; break;
	case 5: v = &p->bsc; break; case 6: v = &p->xp; break; case 7: v = &p->xl; break; case 11: v = &p->hscVal; break;
	default: return -1;
	}
	std::memcpy(v->data(), in, v->size() * 8);
	return (long)v->size();
}

}  // extern "C"
