// ba_linearize.hip -- linearisation + Schur reduction of one trial, destination-major and atomic-free:
//   lm_pass_kernel   landmark pass: Jacobians, IRLS weights, Hll / bl, (Hll + lambda I)^-1, one 64-byte record per edge
//                    (constructQuadraticForm + addLambda + first half of computeBschure, cuda_block_solver.cu:788-839, 906-953)
//   pose_pass        wave = free pose: diagonal block of Hsc, bp, bsc (pose half of constructQuadraticForm, computeBschure :933-962)
//   block_pass       16 lanes = one off-diagonal block of Hsc over its product list (computeHschure :964-977)
//   schur_pass_kernel = the two in ONE launch; pose_maxdiag_kernel (maxDiagonal :877-904).

#include "ba_device.hpp"

namespace cubahip
{

// ===================================================================================================
// Destination-major Schur assembly (default).  No atomics, no per-landmark pair loops:
//   1. lm_pass_kernel     lane = edge, wave = landmarks: Hll/bl reduced in LDS, (Hll+lambda I)^-1, and a
//                         64-byte linearisation record per edge {Xc, w' (sign = stereo), r, landmark};
//   2. pose_pass_kernel   wave = free pose: every edge of the pose contributes Hpp_e - W_e Hpl_e^T, bp_e,
//                         bp_e - Hpl_e Hll^-1 bl to registers, one wave reduction, plain stores;
//   3. block_pass_kernel  16 lanes (a whole wave for blocks with more than BP_HEAVY products) = one off-diagonal block (a,b):
//                         the products of all landmarks seen by both poses, rebuilt from the two records in camera-frame form
//                         (no Hpl tile is ever stored), reduced over the lanes, one plain store.
// Every output has exactly one writer and a fixed summation order => results are reproducible bit for bit.
// ===================================================================================================
constexpr int REC = 8;   // numbers per edge record: [0..2] Xc, [3] w' (sign bit = stereo), [4..6] r, [7] landmark (integer bits)

// Record element type ET = the arithmetic type of the pose / block passes: Scalar, or float for the mixed-precision mode
// of the fp64 library (option "mixed_precision": records and per-edge Jacobian arithmetic in fp32, every accumulation that
// crosses edges and the whole reduced system in fp64 -- the reference's USE_FLOAT32 idea, src/scalar.h:25-29, applied only
// where it is safe).  The landmark travels as an integer bit pattern, exact for any landmark count.
__device__ __forceinline__ double tag_encode(int tag, double) { return __longlong_as_double((long long)tag); }
__device__ __forceinline__ float tag_encode(int tag, float) { return __int_as_float(tag); }
__device__ __forceinline__ int tag_decode(double v) { return (int)__double_as_longlong(v); }
__device__ __forceinline__ int tag_decode(float v) { return __float_as_int(v); }
__device__ __forceinline__ bool sign_flag(double v) { return __double_as_longlong(v) < 0; }
__device__ __forceinline__ bool sign_flag(float v) { return __float_as_int(v) < 0; }
__device__ __forceinline__ double abs_value(double v) { return __builtin_fabs(v); }
__device__ __forceinline__ float abs_value(float v) { return __builtin_fabsf(v); }

template <typename ET>
__device__ __forceinline__ void write_record(Scalar* base, size_t e, const Scalar Xc[3], Scalar wr, const Scalar r[3], int il, bool stereo)
{
	ET* rec = reinterpret_cast<ET*>(base) + REC * e;
	const ET w = (ET)wr;
	rec[0] = (ET)Xc[0]; rec[1] = (ET)Xc[1]; rec[2] = (ET)Xc[2]; rec[3] = stereo ? -w : w;       // (-0.0 keeps the flag of a zero weight)
	rec[4] = (ET)r[0]; rec[5] = (ET)r[1]; rec[6] = (ET)r[2];
	rec[7] = tag_encode(il, ET());
}

template <typename ET>
__device__ __forceinline__ void load_pose_as(const DeviceGraph& g, int ip, ET q[4], ET cam[5])
{
#pragma unroll
	for (int i = 0; i < 4; i++) q[i] = (ET)g.q[4 * (size_t)ip + i];
#pragma unroll
	for (int i = 0; i < 5; i++) cam[i] = (ET)g.cam[5 * (size_t)ip + i];
}

// Workgroups beyond nLmGroups (optimize() only) copy the state into its backup: the push() of the LM loop rides in this launch.
template <int MODE, typename ET>
__device__ __forceinline__ void lm_pass_body(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, Scalar lambda,
	unsigned nLmGroups, const Scalar* __restrict__ backupSrc, Scalar* __restrict__ backupDst, size_t backupCount, unsigned gridX)
{
	__shared__ Scalar lds_all[(LIN_BLOCK / WAVE) * WAVE * 9];
	if (blockIdx.x >= nLmGroups)
	{
		const size_t stride = (size_t)(gridX - nLmGroups) * LIN_BLOCK;
		for (size_t i = (size_t)(blockIdx.x - nLmGroups) * LIN_BLOCK + threadIdx.x; i < backupCount; i += stride) backupDst[i] = backupSrc[i];
		return;
	}
	lambda = launch_lambda(sys, lambda);
	const int lane = threadIdx.x & 63;
	const int wv = threadIdx.x >> 6;
	const int wave = blockIdx.x * (LIN_BLOCK / WAVE) + wv;
	if (wave >= st.nWaves) return;
	Scalar* lds = lds_all + wv * WAVE * 9;
	const int lm0 = st.wave_lm[2 * wave], lm1 = st.wave_lm[2 * wave + 1];
	const int e0 = g.lm_ptr[lm0], e1 = g.lm_ptr[lm1];
	const int e = e0 + lane;
	const bool valid = e < e1;
	int il = lm0, seg0 = 0, seg1 = 0;
	Scalar h[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
	if (valid)
	{
		const int pe = g.e_pose[e];
		const bool stereo = (pe & STEREO_BIT) != 0;
		const int ip = pe & ~STEREO_BIT;
		il = g.e_lm[e];
		Scalar q[4], t[3], cam[5], Xw[3], meas[3], Xc[3];
		EdgeLin L;
		load_pose(g, ip, q, t, cam);
#pragma unroll
		for (int i = 0; i < 3; i++) Xw[i] = g.Xw[3 * (size_t)il + i];
		meas[0] = g.e_mu[e]; meas[1] = g.e_mv[e]; meas[2] = g.e_mr[e];
		const Scalar w = g.e_w[e];
		const Scalar ss = edge_residual(q, t, cam, Xw, meas, stereo, L.r, Xc);
		const int kind = stereo ? g.rk[1].kind : g.rk[0].kind;
		const Scalar delta = stereo ? g.rk[1].delta : g.rk[0].delta;
		const Scalar wr = w * robust_weight(kind, delta, w * ss);
		write_record<ET>(st.e_rec, (size_t)e, Xc, wr, L.r, il, stereo);
		if (il < g.Lf)
		{
			const Rot3 R = quat_to_rot(q[0], q[1], q[2], q[3]);
			edge_jacobians(Xc, R, cam, stereo, L);
#pragma unroll
			for (int i = 0; i < 3; i++)
			{
#pragma unroll
				for (int j = i; j < 3; j++)
					h[sym3_idx(i, j)] = wr * (L.JL[0][i] * L.JL[0][j] + L.JL[1][i] * L.JL[1][j] + L.JL[2][i] * L.JL[2][j]);
				h[6 + i] = wr * (L.JL[0][i] * L.r[0] + L.JL[1][i] * L.r[1] + L.JL[2][i] * L.r[2]);
			}
			seg0 = g.lm_ptr[il] - e0;
			seg1 = g.lm_ptr[il + 1] - e0;
		}
	}
#pragma unroll
	for (int k = 0; k < 9; k++) lds[lane * 9 + k] = h[k];
	wave_lds_sync();
	const bool head = valid && il < g.Lf && lane == seg0;
	Scalar m = 0;
	if (head)
	{
		Scalar H[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
		for (int j = seg0; j < seg1; j++)
#pragma unroll
			for (int k = 0; k < 9; k++) H[k] += lds[j * 9 + k];
		Scalar* ls = sys.lm_sys + 9 * (size_t)il;
		if (MODE == 0)
		{
#pragma unroll
			for (int k = 0; k < 9; k++) ls[k] = H[k];
			m = fmax(H[0], fmax(H[3], H[5]));
		}
		else
		{
			Scalar inv[6];
			H[0] += lambda; H[3] += lambda; H[5] += lambda;
			sym3_inverse(H, inv);
#pragma unroll
			for (int k = 0; k < 6; k++) ls[k] = inv[k];
			if (st.inv_rows8)                               // the block pass reads this copy: 64-byte rows, one sector per gather
			{
				Scalar* li = sys.lm_inv + 8 * (size_t)il;
#pragma unroll
				for (int k = 0; k < 6; k++) li[k] = inv[k];
				li[6] = 0; li[7] = 0;                        // (whole sectors: no read-modify-write at the memory side)
			}
#pragma unroll
			for (int k = 0; k < 3; k++) ls[6 + k] = H[6 + k];
		}
	}
	if (MODE == 0)
	{
		m = wave_max(m);
		if (lane == 0) atomic_max_nonneg(sys.maxdiag + (wave & 63), m);
	}
}


template <int MODE, typename ET>
__global__ __launch_bounds__(LIN_BLOCK) void lm_pass_kernel(DeviceGraph g, DeviceStructure st, DeviceSystem sys, Scalar lambda,
	unsigned nLmGroups, const Scalar* __restrict__ backupSrc, Scalar* __restrict__ backupDst, size_t backupCount)
{
	lm_pass_body<MODE, ET>(g, st, sys, lambda, nLmGroups, backupSrc, backupDst, backupCount, gridDim.x);
}

// batched forms (cuba_hip_optimize_batch): blockIdx.y = graph, arguments from the device table, the damping from device memory
template <typename ET>
__global__ __launch_bounds__(LIN_BLOCK) void lm_pass_batch_kernel(const BatchEntry* __restrict__ tab)
{
	const BatchEntry& e = tab[blockIdx.y];
	if (blockIdx.x >= e.t.lmGrid) return;
	lm_pass_body<1, ET>(e.g, e.st, e.sys, Scalar(-1), e.t.lmGroups, e.t.backupSrc, e.t.backupDst, e.t.backupCount, e.t.lmGrid);
}

// landmarks with more than 64 observations: one workgroup each
template <int MODE, typename ET>
__global__ __launch_bounds__(256) void big_lm_pass_kernel(DeviceGraph g, DeviceStructure st, DeviceSystem sys, Scalar lambda)
{
	__shared__ Scalar red[4][9];
	lambda = launch_lambda(sys, lambda);
	const int il = st.big_lm[blockIdx.x];
	const int e0 = g.lm_ptr[il], e1 = g.lm_ptr[il + 1];
	Scalar acc[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
	for (int e = e0 + threadIdx.x; e < e1; e += 256)
	{
		LaneEdge le;
		linearize_edge(g, e, le);
		// linearize_edge does not return Xc: recompute it for the record
		Scalar q[4], t[3], cam[5], Xw[3], Xc[3];
		load_pose(g, le.ip, q, t, cam);
#pragma unroll
		for (int i = 0; i < 3; i++) Xw[i] = g.Xw[3 * (size_t)il + i];
		quat_rotate(q, Xw, Xc);
		Xc[0] += t[0]; Xc[1] += t[1]; Xc[2] += t[2];
		write_record<ET>(st.e_rec, (size_t)e, Xc, le.wr, le.lin.r, il, le.stereo);
		if (il < g.Lf)
		{
			const EdgeLin& L = le.lin;
#pragma unroll
			for (int i = 0; i < 3; i++)
			{
#pragma unroll
				for (int j = i; j < 3; j++)
					acc[sym3_idx(i, j)] += le.wr * (L.JL[0][i] * L.JL[0][j] + L.JL[1][i] * L.JL[1][j] + L.JL[2][i] * L.JL[2][j]);
				acc[6 + i] += le.wr * (L.JL[0][i] * L.r[0] + L.JL[1][i] * L.r[1] + L.JL[2][i] * L.r[2]);
			}
		}
	}
#pragma unroll
	for (int k = 0; k < 9; k++) acc[k] = wave_sum(acc[k]);
	if ((threadIdx.x & 63) == 0)
#pragma unroll
		for (int k = 0; k < 9; k++) red[threadIdx.x >> 6][k] = acc[k];
	__syncthreads();
	if (threadIdx.x == 0 && il < g.Lf)
	{
		Scalar H[9];
#pragma unroll
		for (int k = 0; k < 9; k++) H[k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
		Scalar* ls = sys.lm_sys + 9 * (size_t)il;
		if (MODE == 0)
		{
#pragma unroll
			for (int k = 0; k < 9; k++) ls[k] = H[k];
			atomic_max_nonneg(sys.maxdiag, fmax(H[0], fmax(H[3], H[5])));
		}
		else
		{
			Scalar inv[6];
			H[0] += lambda; H[3] += lambda; H[5] += lambda;
			sym3_inverse(H, inv);
#pragma unroll
			for (int k = 0; k < 6; k++) ls[k] = inv[k];
			if (st.inv_rows8)
			{
				Scalar* li = sys.lm_inv + 8 * (size_t)il;
#pragma unroll
				for (int k = 0; k < 6; k++) li[k] = inv[k];
				li[6] = 0; li[7] = 0;
			}
#pragma unroll
			for (int k = 0; k < 3; k++) ls[6 + k] = H[6 + k];
		}
	}
}

// Camera-frame form of an edge.  With D = d(projection)/d(Xc) (3x3, five non-zeros: rows (d00, 0, d02), (0, d11, d12) and, for a
// stereo edge, (d00, 0, d22)) the Jacobians of computeJacobians (cuda_block_solver.cu:329-415) are JL = D R and JP = D G with
// G = [-[Xc]x | I].  Everything the Schur passes need is then a 3x3 (or 3-vector) expression in the camera frame, sandwiched between
// G^T and G, i.e. between cross products with Xc:
//     K = w' D^T D (symmetric, K01 = 0)      M = K R      v = w' D^T r
//     Hpp_e = G^T K G      Hpl_e = G^T M      bp_e = G^T v      Hpl_a inv Hpl_b^T = G_a^T [M_a inv M_b^T] G_b
// About 150 multiply-adds per edge in the pose pass (330 with explicit Jacobians) and 200 per product in the block pass (330), and
// neither the 3x6 nor the 3x3 Jacobians are ever held in registers.
template <typename ET>
struct CameraFrameEdge { ET X[3]; ET k00, k02, k11, k12, k22; ET d00, d02, d11, d12, d22; ET w; bool stereo; };

template <typename ET>
__device__ __forceinline__ void camera_frame_edge(const ET* rec, const ET cam[5], CameraFrameEdge<ET>& c)
{
	const ET X = rec[0], Y = rec[1], Z = rec[2], ws = rec[3];
	c.stereo = sign_flag(ws);
	c.w = abs_value(ws);
	c.X[0] = X; c.X[1] = Y; c.X[2] = Z;
	const ET invZ = 1 / Z, invZZ = invZ * invZ;
	c.d00 = -cam[0] * invZ; c.d02 = cam[0] * X * invZZ; c.d11 = -cam[1] * invZ; c.d12 = cam[1] * Y * invZZ;
	c.d22 = c.stereo ? c.d02 - cam[4] * invZZ : ET(0);
	c.k00 = c.w * (c.stereo ? 2 * c.d00 * c.d00 : c.d00 * c.d00);
	c.k02 = c.w * (c.d00 * c.d02 + (c.stereo ? c.d00 * c.d22 : ET(0)));
	c.k11 = c.w * c.d11 * c.d11; c.k12 = c.w * c.d11 * c.d12;
	c.k22 = c.w * (c.d02 * c.d02 + c.d12 * c.d12 + c.d22 * c.d22);
}

template <typename ET>
__device__ __forceinline__ void camera_frame_m(const CameraFrameEdge<ET>& c, const Rot3T<ET>& R, ET (&M)[3][3])
{
#pragma unroll
	for (int j = 0; j < 3; j++)
	{
		M[0][j] = c.k00 * R.m[0][j] + c.k02 * R.m[2][j];
		M[1][j] = c.k11 * R.m[1][j] + c.k12 * R.m[2][j];
		M[2][j] = c.k02 * R.m[0][j] + c.k12 * R.m[1][j] + c.k22 * R.m[2][j];
	}
}

// wave = free pose: diagonal block (upper triangle), bp, bsc.  ET = record / per-edge arithmetic type; sums over edges are
// always accumulated in Scalar.
template <int MODE, typename ET>
__device__ __forceinline__ void pose_pass_body(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, int bid)
{
	const int lane = threadIdx.x & 63;
	const int ip = bid * 4 + (threadIdx.x >> 6);
	if (ip >= g.Pf) return;
	ET q[4], cam[5];
	load_pose_as<ET>(g, ip, q, cam);
	const Rot3T<ET> R = quat_to_rot(q[0], q[1], q[2], q[3]);
	Scalar acc[33];
#pragma unroll
	for (int k = 0; k < 33; k++) acc[k] = 0;
	const int p1 = st.pe_end[ip];
	for (int p = st.pe_beg[ip] + lane; p < p1; p += 64)
	{
		const ET* rec = reinterpret_cast<const ET*>(st.e_rec) + REC * (size_t)st.pe_edge[p];
		CameraFrameEdge<ET> c;
		camera_frame_edge<ET>(rec, cam, c);
		const ET r0 = rec[4], r1 = rec[5], r2 = c.stereo ? rec[6] : ET(0);
		const int il = tag_decode(rec[7]);
		// S = K - M inv M^T (mode 1, free landmark), v = w' D^T r, v' = v - M inv bl
		ET S[6] = { c.k00, 0, c.k02, c.k11, c.k12, c.k22 };
		ET v[3] = { c.w * c.d00 * (r0 + r2), c.w * c.d11 * r1, c.w * (c.d02 * r0 + c.d12 * r1 + c.d22 * r2) };
		ET vs[3] = { v[0], v[1], v[2] };
		if (MODE == 1 && il < g.Lf)
		{
			const Scalar* ls = sys.lm_sys + 9 * (size_t)il;
			ET M[3][3], P[3][3], inv[6], bl[3];
#pragma unroll
			for (int k = 0; k < 6; k++) inv[k] = (ET)ls[k];
#pragma unroll
			for (int k = 0; k < 3; k++) bl[k] = (ET)ls[6 + k];
			camera_frame_m<ET>(c, R, M);
#pragma unroll
			for (int i = 0; i < 3; i++)
#pragma unroll
				for (int k = 0; k < 3; k++)
					P[i][k] = M[i][0] * inv[sym3_idx(0, k)] + M[i][1] * inv[sym3_idx(1, k)] + M[i][2] * inv[sym3_idx(2, k)];
#pragma unroll
			for (int i = 0; i < 3; i++)
			{
#pragma unroll
				for (int j = i; j < 3; j++)
					S[sym3_idx(i, j)] -= P[i][0] * M[j][0] + P[i][1] * M[j][1] + P[i][2] * M[j][2];
				vs[i] -= P[i][0] * bl[0] + P[i][1] * bl[1] + P[i][2] * bl[2];
			}
		}
		// G^T S G = [[ U [X]x^T, U ], [ ., S ]] with U = [X]x S; upper triangle, acc[c (c + 1) / 2 + r] for r <= c
		const ET X = c.X[0], Y = c.X[1], Z = c.X[2];
		ET U[3][3];
#pragma unroll
		for (int j = 0; j < 3; j++)
		{
			const ET s0 = S[sym3_idx(0, j)], s1 = S[sym3_idx(1, j)], s2 = S[sym3_idx(2, j)];
			U[0][j] = Y * s2 - Z * s1;
			U[1][j] = Z * s0 - X * s2;
			U[2][j] = X * s1 - Y * s0;
		}
#pragma unroll
		for (int i = 0; i < 3; i++)
		{
			// row i of U [X]x^T = X x U_i
			const ET t[3] = { Y * U[i][2] - Z * U[i][1], Z * U[i][0] - X * U[i][2], X * U[i][1] - Y * U[i][0] };
#pragma unroll
			for (int j = i; j < 3; j++) acc[j * (j + 1) / 2 + i] += (Scalar)t[j];
#pragma unroll
			for (int j = 0; j < 3; j++) acc[(3 + j) * (4 + j) / 2 + i] += (Scalar)U[i][j];
#pragma unroll
			for (int j = i; j < 3; j++) acc[(3 + j) * (4 + j) / 2 + 3 + i] += (Scalar)S[sym3_idx(i, j)];
		}
		// G^T v = [X x v ; v]
		acc[21] += (Scalar)(Y * v[2] - Z * v[1]); acc[22] += (Scalar)(Z * v[0] - X * v[2]); acc[23] += (Scalar)(X * v[1] - Y * v[0]);
		acc[24] += (Scalar)v[0]; acc[25] += (Scalar)v[1]; acc[26] += (Scalar)v[2];
		if (MODE == 1)
		{
			acc[27] += (Scalar)(Y * vs[2] - Z * vs[1]); acc[28] += (Scalar)(Z * vs[0] - X * vs[2]); acc[29] += (Scalar)(X * vs[1] - Y * vs[0]);
			acc[30] += (Scalar)vs[0]; acc[31] += (Scalar)vs[1]; acc[32] += (Scalar)vs[2];
		}
	}
#pragma unroll
	for (int k = 0; k < 33; k++) acc[k] = wave_sum(acc[k]);
	if (lane == 0)
	{
		Scalar* blk = sys.hsc + 36 * (size_t)st.hsc_rowptr[ip];
#pragma unroll
		for (int c = 0; c < 6; c++)
		{
#pragma unroll
			for (int r = 0; r <= c; r++) blk[c * 6 + r] = acc[c * (c + 1) / 2 + r];
			sys.bp[6 * (size_t)ip + c] = acc[21 + c];
			if (MODE == 1) sys.bsc[6 * (size_t)ip + c] = acc[27 + c];
		}
	}
}

// GROUP lanes = one block (a,b) of Hsc with a != b (or a == b for the rare duplicate-observation products): 16, or the whole wave for
// the first st.nHeavy blocks of the list (more than BP_HEAVY products: KITTI-00's longest list, 424 products, is 7 trips instead of 27).
// ET = record / per-product arithmetic type: a lane's own partial sum is kept in ET, the sum across the lanes and the stored block
// are Scalar.
constexpr int BP_GROUP = 16;

template <int MODE, typename ET>
__global__ __launch_bounds__(256) void pose_pass_kernel(DeviceGraph g, DeviceStructure st, DeviceSystem sys)
{
	pose_pass_body<MODE, ET>(g, st, sys, blockIdx.x);
}

// a product of the block pass in camera-frame form: T_ab = G_a^T [ M_a inv M_b^T ] G_b
template <typename ET>
struct ProductOperand { ET X[3]; ET M[3][3]; };

template <typename ET>
__device__ __forceinline__ void product_operand(const ET* rec, const Rot3T<ET>& R, const ET cam[5], ProductOperand<ET>& o)
{
	CameraFrameEdge<ET> c;
	camera_frame_edge<ET>(rec, cam, c);
	o.X[0] = c.X[0]; o.X[1] = c.X[1]; o.X[2] = c.X[2];
	camera_frame_m<ET>(c, R, o.M);
}

template <typename ET>
__device__ __forceinline__ void product_accumulate(const ProductOperand<ET>& A, const ProductOperand<ET>& B, const ET inv[6], ET (&T)[6][6])
{
	// N = M_a inv M_b^T
	ET P[3][3], W[3][6];
#pragma unroll
	for (int i = 0; i < 3; i++)
#pragma unroll
		for (int k = 0; k < 3; k++)
			P[i][k] = A.M[i][0] * inv[sym3_idx(0, k)] + A.M[i][1] * inv[sym3_idx(1, k)] + A.M[i][2] * inv[sym3_idx(2, k)];
#pragma unroll
	for (int i = 0; i < 3; i++)
	{
#pragma unroll
		for (int j = 0; j < 3; j++) W[i][3 + j] = P[i][0] * B.M[j][0] + P[i][1] * B.M[j][1] + P[i][2] * B.M[j][2];
		// N_i (-[Xb]x) = Xb x N_i
		W[i][0] = B.X[1] * W[i][5] - B.X[2] * W[i][4];
		W[i][1] = B.X[2] * W[i][3] - B.X[0] * W[i][5];
		W[i][2] = B.X[0] * W[i][4] - B.X[1] * W[i][3];
	}
	// T += [ [Xa]x ; I ] W
#pragma unroll
	for (int c = 0; c < 6; c++)
	{
		T[0][c] += A.X[1] * W[2][c] - A.X[2] * W[1][c];
		T[1][c] += A.X[2] * W[0][c] - A.X[0] * W[2][c];
		T[2][c] += A.X[0] * W[1][c] - A.X[1] * W[0][c];
		T[3][c] += W[0][c]; T[4][c] += W[1][c]; T[5][c] += W[2][c];
	}
}

// grp = position in st.od_blocks (or -1: idle lanes), gl = lane within the group
template <typename ET, int GROUP>
__device__ __forceinline__ void block_pass_group(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, int grp, int gl)
{
	const int blk0 = grp >= 0 ? st.od_blocks[grp] : -1;      // (-1 inside the list: unused slot of an XCD-aware order)
	const bool on = blk0 >= 0;
	const int blk = on ? blk0 : 0;
	const int a = on ? st.hsc_blkrow[blk] : 0, b = on ? st.hsc_colind[blk] : 0;
	ET qa[4], cama[5], qb[4], camb[5];
	load_pose_as<ET>(g, a, qa, cama);
	load_pose_as<ET>(g, b, qb, camb);
	const Rot3T<ET> Ra = quat_to_rot(qa[0], qa[1], qa[2], qa[3]);
	const Rot3T<ET> Rb = quat_to_rot(qb[0], qb[1], qb[2], qb[3]);
	ET T[6][6];
#pragma unroll
	for (int r = 0; r < 6; r++)
#pragma unroll
		for (int c = 0; c < 6; c++) T[r][c] = 0;
	const ET* recs = reinterpret_cast<const ET*>(st.e_rec);
	const int p1 = on ? st.prod_end[blk] : 0;
	for (int p = (on ? st.prod_beg[blk] : 0) + gl; p < p1; p += GROUP)
	{
		// three gathers of one 64-byte sector each, issued together (the landmark comes from the product list, not from a record)
		const ET* ra = recs + REC * (size_t)st.prod_ea[p];
		const ET* rb = recs + REC * (size_t)st.prod_eb[p];
		const Scalar* li = st.inv_rows8 ? sys.lm_inv + 8 * (size_t)st.prod_lm[p] : sys.lm_sys + 9 * (size_t)st.prod_lm[p];
		ET inv[6];
#pragma unroll
		for (int k = 0; k < 6; k++) inv[k] = (ET)li[k];
		ProductOperand<ET> A, B;
		product_operand<ET>(ra, Ra, cama, A);
		product_operand<ET>(rb, Rb, camb, B);
		product_accumulate<ET>(A, B, inv, T);
	}
	// reduce over the lanes of the group (in Scalar)
	Scalar Ts[6][6];
#pragma unroll
	for (int r = 0; r < 6; r++)
#pragma unroll
		for (int c = 0; c < 6; c++)
		{
			Scalar v = (Scalar)T[r][c];
			if (GROUP == 64) v = wave_sum(v);
			else { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); }
			Ts[r][c] = v;
		}
	if (!on) return;
	Scalar* dst = sys.hsc + 36 * (size_t)blk;
	if (a != b)
	{
#pragma unroll
		for (int c = 0; c < 6; c++)
#pragma unroll
			for (int r = 0; r < 6; r++)
				if (GROUP == 64 ? (c * 6 + r) == gl : (c * 6 + r) % GROUP == gl) dst[c * 6 + r] = -Ts[r][c];
	}
	else if (gl == 0)
	{
		// duplicate observations of one pose by one landmark: symmetric update of the diagonal block
#pragma unroll
		for (int c = 0; c < 6; c++)
#pragma unroll
			for (int r = 0; r <= c; r++) dst[c * 6 + r] -= Ts[r][c] + Ts[c][r];
	}
}

// workgroup bid of the block pass: the heavy blocks first (one per wave), then 16 light blocks per workgroup
__host__ __device__ __forceinline__ int block_pass_heavy_groups(int nHeavy) { return (nHeavy + 3) / 4; }
__host__ __device__ __forceinline__ int block_pass_groups(int nOd, int nHeavy) { return block_pass_heavy_groups(nHeavy) + ((nOd - nHeavy) * BP_GROUP + 255) / 256; }

template <typename ET>
__device__ __forceinline__ void block_pass_body(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, int bid)
{
	const int nh = block_pass_heavy_groups(st.nHeavy);
	if (bid < nh)
	{
		const int grp = bid * 4 + (threadIdx.x >> 6);
		block_pass_group<ET, 64>(g, st, sys, grp < st.nHeavy ? grp : -1, threadIdx.x & 63);
	}
	else
	{
		const int grp = st.nHeavy + ((bid - nh) * 256 + threadIdx.x) / BP_GROUP;
		block_pass_group<ET, BP_GROUP>(g, st, sys, grp < st.nOd ? grp : -1, threadIdx.x & (BP_GROUP - 1));
	}
}

template <typename ET>
__global__ __launch_bounds__(256) void block_pass_kernel(DeviceGraph g, DeviceStructure st, DeviceSystem sys)
{
	block_pass_body<ET>(g, st, sys, blockIdx.x);
}

// Pose pass and block pass in one launch: they write disjoint parts of the reduced system (diagonal blocks / bp / bsc vs the
// off-diagonal blocks) from the same records.  The pose workgroups come first (one wave per pose: 7 dependent trips at KITTI-00)
// and run under the block workgroups.
template <typename ET>
__global__ __launch_bounds__(256) void schur_pass_kernel(DeviceGraph g, DeviceStructure st, DeviceSystem sys, int nPoseGroups)
{
	if ((int)blockIdx.x < nPoseGroups) pose_pass_body<1, ET>(g, st, sys, blockIdx.x);
	else block_pass_body<ET>(g, st, sys, blockIdx.x - nPoseGroups);
}

template <typename ET>
__global__ __launch_bounds__(256) void schur_pass_batch_kernel(const BatchEntry* __restrict__ tab)
{
	const BatchEntry& e = tab[blockIdx.y];
	if (blockIdx.x >= e.t.schurGrid) return;
	if ((int)blockIdx.x < e.t.poseGroups) pose_pass_body<1, ET>(e.g, e.st, e.sys, blockIdx.x);
	else block_pass_body<ET>(e.g, e.st, e.sys, blockIdx.x - e.t.poseGroups);
}

static DeviceStructure block_pass_view(const DeviceStructure& st, const BlockPassRange& r)
{
	DeviceStructure v = st;
	v.od_blocks = st.od_blocks + r.begin; v.nOd = r.end - r.begin; v.nHeavy = r.heavy;
	return v;
}

template <typename ET>
static void launch_linearize_dm_t(const DeviceGraph& g, const DeviceStructure& stAll, const DeviceSystem& sys, int mode, Scalar lambda, hipStream_t s,
	const Scalar* backupSrc, Scalar* backupDst, size_t backupCount, const BlockPassRange* range)
{
	// (the landmark and pose passes read nothing of the block list: one view serves the whole launch sequence)
	const DeviceStructure st = range ? block_pass_view(stAll, *range) : stAll;
	if (st.nWaves > 0)
	{
		const unsigned grid = (st.nWaves + (LIN_BLOCK / WAVE) - 1) / (LIN_BLOCK / WAVE);
		const unsigned nCopy = backupSrc ? (unsigned)std::min<size_t>(512, (backupCount + LIN_BLOCK - 1) / LIN_BLOCK) : 0;
		if (mode == 0) hipLaunchKernelGGL((lm_pass_kernel<0, ET>), dim3(grid + nCopy), dim3(LIN_BLOCK), 0, s, g, st, sys, lambda, grid, backupSrc, backupDst, backupCount);
		else hipLaunchKernelGGL((lm_pass_kernel<1, ET>), dim3(grid + nCopy), dim3(LIN_BLOCK), 0, s, g, st, sys, lambda, grid, backupSrc, backupDst, backupCount);
	}
	else if (backupSrc && backupCount)
		(void)hipMemcpyAsync(backupDst, backupSrc, backupCount * sizeof(Scalar), hipMemcpyDeviceToDevice, s);
	if (st.nBig > 0)
	{
		if (mode == 0) hipLaunchKernelGGL((big_lm_pass_kernel<0, ET>), dim3(st.nBig), dim3(256), 0, s, g, st, sys, lambda);
		else hipLaunchKernelGGL((big_lm_pass_kernel<1, ET>), dim3(st.nBig), dim3(256), 0, s, g, st, sys, lambda);
	}
	const int nbp = block_pass_groups(st.nOd, st.nHeavy);
	if (mode == 1 && g.Pf > 0 && st.nOd > 0 && st.nDiagProd == 0)     // (duplicate observations: the block pass updates diagonal blocks after the pose pass)
	{
		const int np = (g.Pf + 3) / 4;
		hipLaunchKernelGGL((schur_pass_kernel<ET>), dim3(np + nbp), dim3(256), 0, s, g, st, sys, np);
		return;
	}
	if (g.Pf > 0)
	{
		if (mode == 0) hipLaunchKernelGGL((pose_pass_kernel<0, ET>), dim3((g.Pf + 3) / 4), dim3(256), 0, s, g, st, sys);
		else hipLaunchKernelGGL((pose_pass_kernel<1, ET>), dim3((g.Pf + 3) / 4), dim3(256), 0, s, g, st, sys);
	}
	if (mode == 1 && st.nOd > 0)
		hipLaunchKernelGGL((block_pass_kernel<ET>), dim3(nbp), dim3(256), 0, s, g, st, sys);
}

void launch_linearize_dm(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, int mode, Scalar lambda, hipStream_t s,
	const Scalar* backupSrc, Scalar* backupDst, size_t backupCount, const BlockPassRange* range)
{
	if (st.mixed && sizeof(Scalar) == 8) launch_linearize_dm_t<float>(g, st, sys, mode, lambda, s, backupSrc, backupDst, backupCount, range);
	else launch_linearize_dm_t<Scalar>(g, st, sys, mode, lambda, s, backupSrc, backupDst, backupCount, range);
}

// what launch_linearize_dm_t (mode 1, whole graph, no landmark with more than 64 observations, no duplicate observation) launches for one
// graph, as grid sizes of the batched launches
void batch_fill_linearize(const DeviceGraph& g, const DeviceStructure& st, const Scalar* backupSrc, Scalar* backupDst, size_t backupCount, BatchTrial& t)
{
	t.lmGroups = (st.nWaves + (LIN_BLOCK / WAVE) - 1) / (LIN_BLOCK / WAVE);
	const unsigned nCopy = backupSrc ? (unsigned)std::min<size_t>(512, (backupCount + LIN_BLOCK - 1) / LIN_BLOCK) : 0;
	t.lmGrid = t.lmGroups + nCopy;
	t.backupSrc = backupSrc; t.backupDst = backupDst; t.backupCount = backupCount;
	t.poseGroups = (g.Pf + 3) / 4;
	t.schurGrid = (unsigned)(t.poseGroups + block_pass_groups(st.nOd, st.nHeavy));
}

void launch_batch_linearize(const BatchEntry* tab, int n, unsigned lmGridMax, unsigned schurGridMax, bool mixed, hipStream_t s)
{
	if (mixed && sizeof(Scalar) == 8)
	{
		hipLaunchKernelGGL((lm_pass_batch_kernel<float>), dim3(lmGridMax, n), dim3(LIN_BLOCK), 0, s, tab);
		hipLaunchKernelGGL((schur_pass_batch_kernel<float>), dim3(schurGridMax, n), dim3(256), 0, s, tab);
	}
	else
	{
		hipLaunchKernelGGL((lm_pass_batch_kernel<Scalar>), dim3(lmGridMax, n), dim3(LIN_BLOCK), 0, s, tab);
		hipLaunchKernelGGL((schur_pass_batch_kernel<Scalar>), dim3(schurGridMax, n), dim3(256), 0, s, tab);
	}
}

void launch_block_pass(const DeviceGraph& g, const DeviceStructure& stAll, const DeviceSystem& sys, const BlockPassRange& range, hipStream_t s)
{
	if (range.end <= range.begin) return;
	const DeviceStructure st = block_pass_view(stAll, range);
	const int nbp = block_pass_groups(st.nOd, st.nHeavy);
	if (st.mixed && sizeof(Scalar) == 8) hipLaunchKernelGGL((block_pass_kernel<float>), dim3(nbp), dim3(256), 0, s, g, st, sys);
	else hipLaunchKernelGGL((block_pass_kernel<Scalar>), dim3(nbp), dim3(256), 0, s, g, st, sys);
}

// ---------------------------------------------------------------------------------------------------
// max diagonal of Hpp (diagonal blocks of hsc after an assemble pass).  Ref: maxDiagonalKernel :877-904.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pose_maxdiag_kernel(DeviceGraph g, DeviceStructure st, DeviceSystem sys)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	Scalar m = 0;
	if (i < g.Pf * 6)
	{
		const int p = i / 6, k = i % 6;
		m = sys.hsc[36 * (size_t)st.hsc_rowptr[p] + k * 7];
	}
	m = wave_max(m);
	if ((threadIdx.x & 63) == 0) atomic_max_nonneg(sys.maxdiag, m);
}

void launch_pose_maxdiag(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, hipStream_t s)
{
	if (g.Pf <= 0) return;
	hipLaunchKernelGGL(pose_maxdiag_kernel, dim3((g.Pf * 6 + 255) / 256), dim3(256), 0, s, g, st, sys);
}

}  // namespace cubahip
