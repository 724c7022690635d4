// ba_math.hpp -- per-edge / per-vertex device math of the bundle-adjustment hot path (gfx950).
//
// Behavioural reference (what is computed, not how): /root/reference/src/cuda_block_solver.cu:238-727.
// Everything here is register-resident straight-line code: rotation matrix once per lane, mono and
// stereo edges share one code path (a monocular edge is a stereo edge whose third row is zero), so a
// 64-lane wavefront never diverges on the edge type.
#pragma once

#include <hip/hip_runtime.h>

namespace cubahip
{

// fp64 by default; -DCUBA_HIP_FLOAT32 builds the single-precision variant (the reference's USE_FLOAT32 option,
// /root/reference/src/scalar.h:25-29).  The C ABI stays double at the boundary in both builds.
#ifdef CUBA_HIP_FLOAT32
using Scalar = float;
#else
using Scalar = double;
#endif

constexpr int PDIM = 6;   // se(3) increment [rotation; translation]
constexpr int LDIM = 3;

enum RobustKind : int { ROBUST_NONE = 0, ROBUST_HUBER = 1, ROBUST_TUKEY = 2 };

struct RobustKernel { int kind; Scalar delta; };

// rho(e): robustified squared error.  Ref: cuda_block_solver.cu:676-727.
__device__ __forceinline__ Scalar robust_rho(int kind, Scalar delta, Scalar e)
{
	const Scalar d2 = delta * delta;
	if (kind == ROBUST_HUBER) return e <= d2 ? e : (2 * sqrt(e) * delta - d2);
	if (kind == ROBUST_TUKEY)
	{
		const Scalar u = 1 - e / d2;
		const Scalar top = (Scalar(1) / 3) * d2;
		return e <= d2 ? top * (1 - u * u * u) : top;
	}
	return e;
}

// rho'(e): IRLS weight factor.
__device__ __forceinline__ Scalar robust_weight(int kind, Scalar delta, Scalar e)
{
	const Scalar d2 = delta * delta;
	if (kind == ROBUST_HUBER) return e <= d2 ? Scalar(1) : (delta / sqrt(e));
	if (kind == ROBUST_TUKEY)
	{
		const Scalar u = 1 - e / d2;
		return e <= d2 ? u * u : Scalar(0);
	}
	return 1;
}

// Unit quaternion (x,y,z,w) -> rotation matrix, rows r0,r1,r2.  Ref: cuda_block_solver.cu:292-321.
// (templated on the arithmetic type: the mixed-precision Schur passes of the fp64 library linearise in float)
template <typename T> struct Rot3T { T m[3][3]; };
using Rot3 = Rot3T<Scalar>;

template <typename T>
__device__ __forceinline__ Rot3T<T> quat_to_rot(T x, T y, T z, T w)
{
	const T tx = 2 * x, ty = 2 * y, tz = 2 * z;
	const T twx = tx * w, twy = ty * w, twz = tz * w;
	const T txx = tx * x, txy = ty * x, txz = tz * x;
	const T tyy = ty * y, tyz = tz * y, tzz = tz * z;
	Rot3T<T> R;
	R.m[0][0] = 1 - (tyy + tzz); R.m[0][1] = txy - twz;       R.m[0][2] = txz + twy;
	R.m[1][0] = txy + twz;       R.m[1][1] = 1 - (txx + tzz); R.m[1][2] = tyz - twx;
	R.m[2][0] = txz - twy;       R.m[2][1] = tyz + twx;       R.m[2][2] = 1 - (txx + tyy);
	return R;
}

// q (x) v (x) q^-1 with two cross products.  Ref: cuda_block_solver.cu:238-260.
__device__ __forceinline__ void quat_rotate(const Scalar q[4], const Scalar v[3], Scalar out[3])
{
	Scalar a0 = q[1] * v[2] - q[2] * v[1];
	Scalar a1 = q[2] * v[0] - q[0] * v[2];
	Scalar a2 = q[0] * v[1] - q[1] * v[0];
	a0 += a0; a1 += a1; a2 += a2;
	out[0] = v[0] + q[3] * a0 + (q[1] * a2 - q[2] * a1);
	out[1] = v[1] + q[3] * a1 + (q[2] * a0 - q[0] * a2);
	out[2] = v[2] + q[3] * a2 + (q[0] * a1 - q[1] * a0);
}

// One observation: everything an edge contributes to the normal equations.
template <typename T>
struct EdgeLinT
{
	T r[3];        // residual proj - meas (r[2] = 0 for monocular)
	T JP[3][6];    // d err / d [omega; upsilon]  (row 2 zero for monocular)
	T JL[3][3];    // d err / d Xw
};
using EdgeLin = EdgeLinT<Scalar>;

// Residual only.  Ref: computeActiveErrorsKernel, cuda_block_solver.cu:733-786 (projectW2C/projectC2I :262-290).
// Returns squared residual norm (unweighted).
__device__ __forceinline__ Scalar edge_residual(const Scalar q[4], const Scalar t[3], const Scalar cam[5],
	const Scalar Xw[3], const Scalar meas[3], bool stereo, Scalar r[3], Scalar Xc[3])
{
	quat_rotate(q, Xw, Xc);
	Xc[0] += t[0]; Xc[1] += t[1]; Xc[2] += t[2];
	const Scalar invZ = 1 / Xc[2];
	const Scalar u = cam[0] * invZ * Xc[0] + cam[2];
	const Scalar v = cam[1] * invZ * Xc[1] + cam[3];
	r[0] = u - meas[0];
	r[1] = v - meas[1];
	r[2] = stereo ? (u - cam[4] * invZ) - meas[2] : Scalar(0);
	return r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
}

// Analytic Jacobians at camera-frame point Xc.  Ref: computeJacobians<2>/<3>, cuda_block_solver.cu:329-415.
template <typename T>
__device__ __forceinline__ void edge_jacobians(const T Xc[3], const Rot3T<T>& R, const T cam[5], bool stereo, EdgeLinT<T>& L)
{
	const T X = Xc[0], Y = Xc[1];
	const T invZ = 1 / Xc[2], invZZ = invZ * invZ;
	const T fu = cam[0], fv = cam[1], bf = stereo ? cam[4] : T(0);
	const T s = stereo ? T(1) : T(0);
#pragma unroll
	for (int j = 0; j < 3; j++)
	{
		L.JL[0][j] = -fu * R.m[0][j] * invZ + fu * X * R.m[2][j] * invZZ;
		L.JL[1][j] = -fv * R.m[1][j] * invZ + fv * Y * R.m[2][j] * invZZ;
		L.JL[2][j] = s * (L.JL[0][j] - bf * R.m[2][j] * invZZ);
	}
	L.JP[0][0] = X * Y * invZZ * fu;
	L.JP[0][1] = -(1 + (X * X * invZZ)) * fu;
	L.JP[0][2] = Y * invZ * fu;
	L.JP[0][3] = -invZ * fu;
	L.JP[0][4] = 0;
	L.JP[0][5] = X * invZZ * fu;
	L.JP[1][0] = (1 + Y * Y * invZZ) * fv;
	L.JP[1][1] = -X * Y * invZZ * fv;
	L.JP[1][2] = -X * invZ * fv;
	L.JP[1][3] = 0;
	L.JP[1][4] = -invZ * fv;
	L.JP[1][5] = Y * invZZ * fv;
	L.JP[2][0] = s * (L.JP[0][0] - bf * Y * invZZ);
	L.JP[2][1] = s * (L.JP[0][1] + bf * X * invZZ);
	L.JP[2][2] = s * L.JP[0][2];
	L.JP[2][3] = s * L.JP[0][3];
	L.JP[2][4] = 0;
	L.JP[2][5] = s * (L.JP[0][5] - bf * invZZ);
}

// Inverse of a symmetric 3x3 given by its 6 unique entries (00,01,02,11,12,22); adjugate / determinant,
// no pivoting.  Ref: Sym3x3Inv, cuda_block_solver.cu:417-452.
__device__ __forceinline__ void sym3_inverse(const Scalar A[6], Scalar B[6])
{
	const Scalar A00 = A[0], A01 = A[1], A02 = A[2], A11 = A[3], A12 = A[4], A22 = A[5];
	const Scalar det = A00 * A11 * A22 + A01 * A12 * A02 + A02 * A01 * A12
		- A00 * A12 * A12 - A02 * A11 * A02 - A01 * A01 * A22;
	const Scalar id = 1 / det;
	B[0] = id * (A11 * A22 - A12 * A12);
	B[1] = id * (A02 * A12 - A01 * A22);
	B[2] = id * (A01 * A12 - A02 * A11);
	B[3] = id * (A00 * A22 - A02 * A02);
	B[4] = id * (A02 * A01 - A00 * A12);
	B[5] = id * (A00 * A11 - A01 * A01);
}

// index of (i,j) in the 6-entry symmetric 3x3 packing
__device__ __forceinline__ constexpr int sym3_idx(int i, int j)
{
	return i <= j ? (i == 0 ? j : (i == 1 ? 2 + j : 5)) : (j == 0 ? i : (j == 1 ? 2 + i : 5));
}

// rotation matrix -> quaternion, the branch for a non-positive trace with i the largest diagonal entry, j = i + 1, k = j + 1 (mod 3)
template <int I, int J, int K>
__device__ __forceinline__ void shepperd_branch(const Scalar (&R)[3][3], Scalar (&qe)[4])
{
	Scalar tr = sqrt(R[I][I] - R[J][J] - R[K][K] + 1);
	qe[I] = Scalar(0.5) * tr;
	tr = Scalar(0.5) / tr;
	qe[3] = (R[K][J] - R[J][K]) * tr;
	qe[J] = (R[J][I] + R[I][J]) * tr;
	qe[K] = (R[K][I] + R[I][K]) * tr;
}

// SE3 exponential + left-multiplicative pose update T <- exp([omega; upsilon]) * T.
// Ref: updateExp / updatePose and helpers, cuda_block_solver.cu:454-592.
__device__ __forceinline__ void pose_exp_update(const Scalar upd[6], Scalar q[4], Scalar t[3])
{
	const Scalar wx = upd[0], wy = upd[1], wz = upd[2];
	const Scalar theta = sqrt(wx * wx + wy * wy + wz * wz);
	Scalar a1, a2, a3;
	if (theta < Scalar(0.00001)) { a1 = 1; a2 = Scalar(0.5); a3 = Scalar(1) / 6; }
	else
	{
		const Scalar sn = sin(theta), cs = cos(theta);
		a1 = sn / theta;
		a2 = (1 - cs) / (theta * theta);
		a3 = (theta - sn) / (theta * theta * theta);
	}
	// K = [w]x, K2 = K*K ; R = I + a1 K + a2 K2 ; V = I + a2 K + a3 K2
	const Scalar xx = wx * wx, yy = wy * wy, zz = wz * wz, xy = wx * wy, yz = wy * wz, zx = wz * wx;
	const Scalar K[3][3] = { { 0, -wz, wy }, { wz, 0, -wx }, { -wy, wx, 0 } };
	const Scalar K2[3][3] = { { -yy - zz, xy, zx }, { xy, -zz - xx, yz }, { zx, yz, -xx - yy } };
	Scalar R[3][3], te[3];
#pragma unroll
	for (int i = 0; i < 3; i++)
	{
		Scalar acc = 0;
#pragma unroll
		for (int j = 0; j < 3; j++)
		{
			const Scalar I = (i == j) ? Scalar(1) : Scalar(0);
			R[i][j] = I + a1 * K[i][j] + a2 * K2[i][j];
			acc += (I + a2 * K[i][j] + a3 * K2[i][j]) * upd[3 + j];
		}
		te[i] = acc;
	}
	// rotation matrix -> quaternion (Eigen / Shepperd branches, :492-521)
	Scalar qe[4];
	Scalar tr = R[0][0] + R[1][1] + R[2][2];
	if (tr > 0)
	{
		tr = sqrt(tr + 1);
		qe[3] = Scalar(0.5) * tr;
		tr = Scalar(0.5) / tr;
		qe[0] = (R[2][1] - R[1][2]) * tr;
		qe[1] = (R[0][2] - R[2][0]) * tr;
		qe[2] = (R[1][0] - R[0][1]) * tr;
	}
	else
	{
		// i = index of the largest diagonal entry (same comparisons as the reference); three explicit cases so that every
		// index is a compile-time constant (a runtime index would put R and qe into scratch memory)
		const bool c1 = R[1][1] > R[0][0];
		const bool c2 = R[2][2] > (c1 ? R[1][1] : R[0][0]);
		if (c2) shepperd_branch<2, 0, 1>(R, qe);
		else if (c1) shepperd_branch<1, 2, 0>(R, qe);
		else shepperd_branch<0, 1, 2>(R, qe);
	}
	// t <- t_exp + R(q_exp) t ; q <- normalise(q_exp * q), w >= 0   (:523-539, :581-592)
	Scalar u[3];
	quat_rotate(qe, t, u);
	t[0] = te[0] + u[0]; t[1] = te[1] + u[1]; t[2] = te[2] + u[2];
	Scalar c[4];
	c[3] = qe[3] * q[3] - qe[0] * q[0] - qe[1] * q[1] - qe[2] * q[2];
	c[0] = qe[3] * q[0] + qe[0] * q[3] + qe[1] * q[2] - qe[2] * q[1];
	c[1] = qe[3] * q[1] + qe[1] * q[3] + qe[2] * q[0] - qe[0] * q[2];
	c[2] = qe[3] * q[2] + qe[2] * q[3] + qe[0] * q[1] - qe[1] * q[0];
	Scalar invn = 1 / sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3]);
	if (c[3] < 0) invn = -invn;
	q[0] = invn * c[0]; q[1] = invn * c[1]; q[2] = invn * c[2]; q[3] = invn * c[3];
}

// In-place Cholesky-based inverse of a symmetric positive definite 6x6 (full storage, row r col c at A[c*6+r]).
// Returns false on a non-positive pivot.  Used for the block-Jacobi preconditioner.
__device__ __forceinline__ bool spd6_inverse(const Scalar A[36], Scalar Ainv[36])
{
	Scalar Lm[6][6];
	bool ok = true;
#pragma unroll
	for (int c = 0; c < 6; c++)
	{
		Scalar d = A[c * 6 + c];
#pragma unroll
		for (int p = 0; p < c; p++) d -= Lm[c][p] * Lm[c][p];
		if (!(d > 0)) { ok = false; d = 1; }
		const Scalar sd = sqrt(d);
		Lm[c][c] = sd;
		const Scalar isd = 1 / sd;
#pragma unroll
		for (int r = c + 1; r < 6; r++)
		{
			Scalar s = A[c * 6 + r];
#pragma unroll
			for (int p = 0; p < c; p++) s -= Lm[r][p] * Lm[c][p];
			Lm[r][c] = s * isd;
		}
	}
	// Linv = L^-1 (lower), then Ainv = Linv^T Linv
	Scalar Li[6][6];
#pragma unroll
	for (int c = 0; c < 6; c++)
	{
		Li[c][c] = 1 / Lm[c][c];
#pragma unroll
		for (int r = c + 1; r < 6; r++)
		{
			Scalar s = 0;
#pragma unroll
			for (int p = c; p < r; p++) s -= Lm[r][p] * Li[p][c];
			Li[r][c] = s / Lm[r][r];
		}
	}
#pragma unroll
	for (int i = 0; i < 6; i++)
#pragma unroll
		for (int j = 0; j <= i; j++)
		{
			Scalar s = 0;
#pragma unroll
			for (int p = i; p < 6; p++) s += Li[p][i] * Li[p][j];
			Ainv[j * 6 + i] = s;
			Ainv[i * 6 + j] = s;
		}
	return ok;
}

}  // namespace cubahip
