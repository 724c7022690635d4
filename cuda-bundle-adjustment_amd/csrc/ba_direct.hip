// ba_direct.hip -- the EXACT reduced solve: Hsc dxp = bsc by a dense blocked Cholesky factorisation on the matrix cores
// (v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32) plus the two triangular solves.  It stands where the reference calls cuSOLVER's
// sparse Cholesky (/root/reference/src/cuda_linear_solver.cpp:386-415: solve() is exact for any positive-definite Hsc and reports
// failure only on a non-positive pivot, :406-410, which CudaBundleAdjustmentImpl::optimize turns into rho = -1,
// /root/reference/src/cuda_bundle_adjustment.cpp:824-830) for the systems the block PCG of ba_pcg.hip is the wrong tool for: a nearly
// singular reduced matrix (most observations of many poses at zero robust weight) needs thousands of CG iterations or never gets there.
// ba_lm.hip hands a solve over to this file when the PCG has used up its iteration budget, broke down, or an earlier solve of the same
// Levenberg-Marquardt run already had to come here.
//
// Layout: A is [(N + 160) x N] column-major with N = 6 Pf rounded up to 128; only the tiles on and below the diagonal are read or written.
// Row N carries the right-hand side: the factorisation of the bordered matrix [A b; b^T .] leaves L^-1 b in that row, so the forward
// substitution costs nothing extra (the row is simply one more tile row of every kernel below).  Rows N + 1 .. N + 127 are zero padding
// of the right-hand side's macro row, the last 32 rows keep the column stride off the powers of two.  Unknowns n .. N - 1 are identity
// padding.
//
// Right-looking, two-level blocking (panel = 128 columns = 4 tile columns of 32):
//   for every tile column k of the panel:   chol_panel_kernel  X_k = chol(A_kk)^-1 (every workgroup, in LDS: one sweep of 32 dependent steps
//                                                              that eliminates A_kk and the identity beside it), A_ik <- A_ik X_k^T (matrix cores)
//                                           chol_upd32_kernel  A_ij -= A_ik A_jk^T for the remaining tile columns j of the panel
//   once per panel:                         chol_trail_kernel  A_IJ -= W_I W_J^T over 128 x 128 macro tiles, k = 128 (where the flops are:
//                                                              n^3 / 3 in total, C read and written once per panel instead of once per tile column)
//   at the end, per tile column k = T-1..0: chol_back_kernel   x_k = X_k^T y_k, then y_j -= L_kj^T x_k for every j < k (one workgroup per j)
// Every sum has a fixed order: the solve is reproducible bit for bit like the rest of the path.

#include "ba_mfma.hpp"

namespace cubahip
{

constexpr int CH_T = 32;       // tile edge
constexpr int CH_P = 128;      // panel width = macro tile edge
constexpr int CH_PAD = 160;    // rows below the matrix (right-hand-side macro row + 32)

size_t dense_cholesky_elems(int n, int* N, int* ld)
{
	const int NN = (n + CH_P - 1) / CH_P * CH_P;
	if (N) *N = NN;
	if (ld) *ld = NN + CH_PAD;
	return (size_t)(NN + CH_PAD) * NN;
}

// upper-triangular BSR (damped, diagonal blocks full) -> lower triangle of the dense matrix
__global__ __launch_bounds__(256) void dense_fill_kernel(DeviceStructure st, DeviceSystem sys, DenseCholesky d)
{
	const int b = blockIdx.x * 4 + (threadIdx.x >> 6), e = threadIdx.x & 63;
	if (b >= st.nblk || e >= 36) return;
	const int bi = st.hsc_blkrow[b], bj = st.hsc_colind[b];        // bi <= bj; element (r, c) of the 6 x 6 block, column-major
	const int r = e % 6, c = e / 6;
	const int row = 6 * bj + c, col = 6 * bi + r;                  // its mirror image below the diagonal
	if (row >= col) d.A[(size_t)col * d.ld + row] = sys.hsc[36 * (size_t)b + e];
}

__global__ __launch_bounds__(256) void dense_fill_rhs_kernel(const Scalar* __restrict__ b, DenseCholesky d)
{
	const int j = blockIdx.x * 256 + threadIdx.x;
	if (j >= d.N) return;
	d.A[(size_t)j * d.ld + d.N] = j < d.n ? b[j] : Scalar(0);
	if (j >= d.n) d.A[(size_t)j * d.ld + j] = Scalar(1);
}

void launch_dense_fill(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, const DenseCholesky& d, hipStream_t s)
{
	(void)g;
	(void)hipMemsetAsync(d.A, 0, sizeof(Scalar) * (size_t)d.ld * d.N, s);
	if (st.nblk) hipLaunchKernelGGL(dense_fill_kernel, dim3((st.nblk + 3) / 4), dim3(256), 0, s, st, sys, d);
	hipLaunchKernelGGL(dense_fill_rhs_kernel, dim3((d.N + 255) / 256), dim3(256), 0, s, sys.bsc, d);
	(void)hipMemsetAsync(d.fail, 0, sizeof(int), s);
}

// Tile column k: every workgroup factorises the diagonal tile for itself (the chain is latency, not work: doing it once and handing it
// over would cost a launch boundary per tile column) and applies L_kk^-T to ITS tile (i, k), i = k + 1 + blockIdx.x (the last one is the
// right-hand side's tile).  The diagonal tile itself stays as it was in the matrix -- nobody reads it again; workgroup 0 stores
// X_k = L_kk^-1 (row-major) for the backward substitution and raises the failure flag on a non-positive pivot (the factorisation then
// carries on with pivot 1 so that nothing downstream sees a NaN; the solve is reported as failed).
__global__ __launch_bounds__(256) void chol_panel_kernel(DenseCholesky d, int k)
{
	__shared__ Scalar Tm[CH_T][CH_T + 1];
	__shared__ Scalar Xs[CH_T][CH_T + 1];
	__shared__ Scalar F[CH_T][CH_T + 1];
	__shared__ Scalar rdiag[CH_T];
	const int tid = threadIdx.x, r = tid & 31, cb = tid >> 5;
	const size_t ld = d.ld;
	const int i = k + 1 + blockIdx.x;
	const Scalar* Akk = d.A + (size_t)(k * CH_T) * ld + (size_t)k * CH_T;
	Scalar* Aik = d.A + (size_t)(k * CH_T) * ld + (size_t)i * CH_T;
	Scalar own[4], fv[4];
#pragma unroll
	for (int u = 0; u < 4; u++)
	{
		own[u] = Akk[(size_t)(cb + 8 * u) * ld + r];
		fv[u] = Aik[(size_t)(cb + 8 * u) * ld + r];
	}
#pragma unroll
	for (int u = 0; u < 4; u++) { Tm[r][cb + 8 * u] = own[u]; F[r][cb + 8 * u] = fv[u]; }
	__syncthreads();
	// Unscaled right-looking elimination, 32 dependent steps with one barrier each: after step j the columns <= j + 1 of Tm are final,
	// T[r][c] = L[r][c] L[c][c].  The same row operations -- row r -= (T[r][j] / T[j][j]) row j -- applied to the identity beside it give
	// M = (unit-lower factor)^-1, so X = L^-1 = diag(T)^-1/2 M comes out of the SAME sweep.  (Measured against a second sweep of 32 steps
	// for the triangular inversion: the same 19 us per tile column -- a step is bound by its LDS operations, ten reads and five writes here,
	// not by its barrier --, so the one-sweep form is kept for being shorter, not faster: profiles/r05n_exact_solve_times.txt.)
	// Thread (r, cb) keeps its four elements (r, cb + 8u) of T and of M in registers and publishes column j + 1 of T and row j + 1 of M
	// for the next step.
	Scalar mm[4];
#pragma unroll
	for (int u = 0; u < 4; u++) { mm[u] = r == cb + 8 * u ? Scalar(1) : Scalar(0); Xs[r][cb + 8 * u] = mm[u]; }
	__syncthreads();
	bool bad = false;
#pragma unroll
	for (int j = 0; j < CH_T; j++)
	{
		Scalar dj = Tm[j][j];
		if (!(dj > Scalar(0))) { bad = true; dj = Scalar(1); }
		const Scalar lr = r > j ? Tm[r][j] * fast_rcp(dj) : Scalar(0);
#pragma unroll
		for (int u = 0; u < 4; u++)
		{
			const int c = cb + 8 * u;
			if (c > j && r >= c) own[u] -= lr * Tm[c][j];
			if (c <= j) mm[u] -= lr * Xs[j][c];                 // (row j of M is final: rows below it change)
			if (c == j + 1) Tm[r][c] = own[u];
		}
		if (r == j + 1)
		{
#pragma unroll
			for (int u = 0; u < 4; u++) Xs[r][cb + 8 * u] = mm[u];
		}
		__syncthreads();
	}
	if (tid < CH_T)
	{
		const Scalar dd = Tm[tid][tid];
		rdiag[tid] = dd > Scalar(0) ? Scalar(1) / sqrt(dd) : Scalar(1);
	}
	__syncthreads();
	{
		const Scalar rd = rdiag[r];
#pragma unroll
		for (int u = 0; u < 4; u++) Xs[r][cb + 8 * u] = cb + 8 * u <= r ? mm[u] * rd : Scalar(0);       // X = L^-1, zero above the diagonal
	}
	__syncthreads();
	// own tile: Out[r][c] = sum_m F[r][m] X[c][m] on the matrix cores (wave w owns the 16 x 16 output tile (w >> 1, w & 1))
	const int wv = tid >> 6, lane = tid & 63, wi = wv >> 1, wj = wv & 1;
	MfmaAcc o = mfma_zero();
#pragma unroll
	for (int s4 = 0; s4 < CH_T; s4 += 4)
		o = mfma_16x16x4(F[16 * wi + (lane & 15)][s4 + (lane >> 4)], Xs[16 * wj + (lane & 15)][s4 + (lane >> 4)], o);
	__syncthreads();
#pragma unroll
	for (int q = 0; q < 4; q++) F[16 * wi + mfma_row(lane, q)][16 * wj + (lane & 15)] = mfma_get(o, q);
	__syncthreads();
#pragma unroll
	for (int u = 0; u < 4; u++) Aik[(size_t)(cb + 8 * u) * ld + r] = F[r][cb + 8 * u];
	if (blockIdx.x == 0)
	{
		Scalar* X = d.invL + (size_t)k * CH_T * CH_T;
#pragma unroll
		for (int u = 0; u < 4; u++) X[(cb + 8 * u) * CH_T + r] = Xs[cb + 8 * u][r];
		if (bad && tid == 0) *d.fail = 1;
	}
}

// A_ij -= A_ik A_jk^T for the tile columns j = j0t + blockIdx.y of the panel that are still to be factorised, i = j + blockIdx.x
__global__ __launch_bounds__(256) void chol_upd32_kernel(DenseCholesky d, int k, int j0t, int Tr)
{
	__shared__ Scalar F[CH_T][CH_T + 1];
	__shared__ Scalar G[CH_T][CH_T + 1];
	const int j = j0t + blockIdx.y, i = j + blockIdx.x;
	if (i > Tr) return;
	const int tid = threadIdx.x, r = tid & 31, cb = tid >> 5;
	const size_t ld = d.ld;
	const Scalar* Aik = d.A + (size_t)(k * CH_T) * ld + (size_t)i * CH_T;
	const Scalar* Ajk = d.A + (size_t)(k * CH_T) * ld + (size_t)j * CH_T;
	Scalar* Aij = d.A + (size_t)(j * CH_T) * ld + (size_t)i * CH_T;
	Scalar fv[4], gv[4], sv[4];
#pragma unroll
	for (int u = 0; u < 4; u++)
	{
		const size_t off = (size_t)(cb + 8 * u) * ld + r;
		fv[u] = Aik[off]; gv[u] = Ajk[off]; sv[u] = Aij[off];
	}
#pragma unroll
	for (int u = 0; u < 4; u++) { F[r][cb + 8 * u] = fv[u]; G[r][cb + 8 * u] = gv[u]; }
	__syncthreads();
	const int wv = tid >> 6, lane = tid & 63, wi = wv >> 1, wj = wv & 1;
	MfmaAcc o = mfma_zero();
#pragma unroll
	for (int s4 = 0; s4 < CH_T; s4 += 4)
		o = mfma_16x16x4(F[16 * wi + (lane & 15)][s4 + (lane >> 4)], G[16 * wj + (lane & 15)][s4 + (lane >> 4)], o);
	__syncthreads();
#pragma unroll
	for (int q = 0; q < 4; q++) F[16 * wi + mfma_row(lane, q)][16 * wj + (lane & 15)] = mfma_get(o, q);
	__syncthreads();
#pragma unroll
	for (int u = 0; u < 4; u++) Aij[(size_t)(cb + 8 * u) * ld + r] = sv[u] - F[r][cb + 8 * u];
}

// Trailing update of one panel: C_IJ -= W_I W_J^T, W = the panel's 128 columns (in place), one 128 x 128 macro tile per workgroup,
// I >= J (2-D grid, the workgroups above the diagonal return at once; I = N / 128 is the right-hand side's macro row).  Four waves in
// a 2 x 2 arrangement, 64 x 64 outputs = 4 x 4 matrix-core tiles each; the panel operands go through LDS in chunks of 16 columns,
// double-buffered (the next chunk's global loads are in flight while the matrix cores work on this one).  The products are formed
// transposed -- the instruction's "A" operand is the COLUMN-side panel -- so that a lane ends up with 16 consecutive rows of a
// column across its half-row of lanes: the read-modify-write of C is 128-byte segments, no LDS transposition.
constexpr int TR_KC = 16;
constexpr int TR_LD = CH_P + 16;
typedef Scalar Scalar2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256, 2) void chol_trail_kernel(DenseCholesky d, int c0, int Jfirst)
{
	const int J = Jfirst + blockIdx.x, I = Jfirst + blockIdx.y;
	if (I < J) return;
	__shared__ __attribute__((aligned(16))) Scalar sA[2][TR_KC][TR_LD];      // row-side operand (macro row I)
	__shared__ __attribute__((aligned(16))) Scalar sB[2][TR_KC][TR_LD];      // column-side operand (macro row J)
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wi = wv >> 1, wj = wv & 1;
	const size_t ld = d.ld;
	const Scalar* pA = d.A + (size_t)c0 * ld + (size_t)I * CH_P + 2 * lane;
	const Scalar* pB = d.A + (size_t)c0 * ld + (size_t)J * CH_P + 2 * lane;
	Scalar2 ra[4], rb[4];
#pragma unroll
	for (int p = 0; p < 4; p++)
	{
		const size_t off = (size_t)(4 * p + wv) * ld;
		ra[p] = *(const Scalar2*)(pA + off);
		rb[p] = *(const Scalar2*)(pB + off);
	}
#pragma unroll
	for (int p = 0; p < 4; p++)
	{
		*(Scalar2*)&sA[0][4 * p + wv][2 * lane] = ra[p];
		*(Scalar2*)&sB[0][4 * p + wv][2 * lane] = rb[p];
	}
	__syncthreads();
	MfmaAcc acc[4][4];
#pragma unroll
	for (int mi = 0; mi < 4; mi++)
#pragma unroll
		for (int nj = 0; nj < 4; nj++) acc[mi][nj] = mfma_zero();
	constexpr int NCH = CH_P / TR_KC;
#pragma unroll 1
	for (int ch = 0; ch < NCH; ch++)
	{
		const int buf = ch & 1;
		if (ch + 1 < NCH)
		{
#pragma unroll
			for (int p = 0; p < 4; p++)
			{
				const size_t off = (size_t)((ch + 1) * TR_KC + 4 * p + wv) * ld;
				ra[p] = *(const Scalar2*)(pA + off);
				rb[p] = *(const Scalar2*)(pB + off);
			}
		}
#pragma unroll
		for (int ks = 0; ks < TR_KC / 4; ks++)
		{
			const int kk = 4 * ks + (lane >> 4);
			Scalar a[4], b[4];
#pragma unroll
			for (int t = 0; t < 4; t++)
			{
				a[t] = sB[buf][kk][64 * wj + 16 * t + (lane & 15)];
				b[t] = sA[buf][kk][64 * wi + 16 * t + (lane & 15)];
			}
#pragma unroll
			for (int mi = 0; mi < 4; mi++)
#pragma unroll
				for (int nj = 0; nj < 4; nj++) acc[mi][nj] = mfma_16x16x4(a[nj], b[mi], acc[mi][nj]);
		}
		if (ch + 1 < NCH)
		{
#pragma unroll
			for (int p = 0; p < 4; p++)
			{
				*(Scalar2*)&sA[buf ^ 1][4 * p + wv][2 * lane] = ra[p];
				*(Scalar2*)&sB[buf ^ 1][4 * p + wv][2 * lane] = rb[p];
			}
		}
		__syncthreads();
	}
	// lane l holds C[row = 16 mi + (l & 15)][column = 16 nj + mfma_row(l, q)] of its wave's 64 x 64 piece
	Scalar* C = d.A + (size_t)(J * CH_P + 64 * wj) * ld + (size_t)I * CH_P + 64 * wi + (lane & 15);
#pragma unroll
	for (int nj = 0; nj < 4; nj++)
#pragma unroll
		for (int q = 0; q < 4; q++)
		{
			Scalar* col = C + (size_t)(16 * nj + mfma_row(lane, q)) * ld;
			Scalar v[4];
#pragma unroll
			for (int mi = 0; mi < 4; mi++) v[mi] = col[16 * mi];
#pragma unroll
			for (int mi = 0; mi < 4; mi++) col[16 * mi] = v[mi] - mfma_get(acc[mi][nj], q);
		}
}

__global__ __launch_bounds__(256) void chol_extract_y_kernel(DenseCholesky d)
{
	const int j = blockIdx.x * 256 + threadIdx.x;
	if (j < d.N) d.y[j] = d.A[(size_t)j * d.ld + d.N];
}

// Backward substitution, tile column k (k = T-1 .. 0, one launch each): every workgroup forms x_k = X_k^T y_k for itself (y_k is final:
// all launches k' > k have subtracted their share), workgroup j < k then subtracts L_kj^T x_k from y_j; workgroup 0 stores x_k.
__global__ __launch_bounds__(256) void chol_back_kernel(DenseCholesky d, int k, Scalar* __restrict__ x)
{
	__shared__ Scalar red[8][CH_T + 1];
	__shared__ Scalar xk[CH_T];
	__shared__ Scalar P[CH_T][CH_T + 1];
	const int tid = threadIdx.x, r = tid & 31, cb = tid >> 5;
	const size_t ld = d.ld;
	const int j = blockIdx.x;
	const bool upd = j < k;
	const Scalar* Lkj = d.A + (size_t)(j * CH_T) * ld + (size_t)k * CH_T;
	Scalar lv[4];
#pragma unroll
	for (int u = 0; u < 4; u++) lv[u] = upd ? Lkj[(size_t)(cb + 8 * u) * ld + r] : Scalar(0);
	const Scalar* X = d.invL + (size_t)k * CH_T * CH_T;          // X[row][column], row-major
	Scalar part = 0;
#pragma unroll
	for (int u = 0; u < 4; u++) part += X[(4 * cb + u) * CH_T + r] * d.y[k * CH_T + 4 * cb + u];
	red[cb][r] = part;
	__syncthreads();
	if (tid < CH_T)
	{
		const Scalar s = ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])) + ((red[4][tid] + red[5][tid]) + (red[6][tid] + red[7][tid]));
		xk[tid] = s;
		if (blockIdx.x == 0 && k * CH_T + tid < d.n) x[k * CH_T + tid] = s;
	}
	__syncthreads();
	if (!upd) return;
#pragma unroll
	for (int u = 0; u < 4; u++) P[cb + 8 * u][r] = lv[u] * xk[r];
	__syncthreads();
	if (tid < CH_T)
	{
		Scalar s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
		for (int m = 0; m < CH_T; m += 4) { s0 += P[tid][m]; s1 += P[tid][m + 1]; s2 += P[tid][m + 2]; s3 += P[tid][m + 3]; }
		d.y[j * CH_T + tid] -= (s0 + s1) + (s2 + s3);
	}
}

void launch_dense_cholesky_solve(const DenseCholesky& d, Scalar* x, hipStream_t s)
{
	const int T = d.N / CH_T, NM = d.N / CH_P, Tr = T;       // Tr: the right-hand side's tile row
	for (int p = 0; p < NM; p++)
	{
		for (int t = 0; t < CH_P / CH_T; t++)
		{
			const int k = 4 * p + t;
			hipLaunchKernelGGL(chol_panel_kernel, dim3(Tr - k), dim3(256), 0, s, d, k);
			if (t < 3) hipLaunchKernelGGL(chol_upd32_kernel, dim3(Tr - k, 3 - t), dim3(256), 0, s, d, k, k + 1, Tr);
		}
		const int m = NM - 1 - p;
		if (m > 0) hipLaunchKernelGGL(chol_trail_kernel, dim3(m, m + 1), dim3(256), 0, s, d, CH_P * p, p + 1);
	}
	hipLaunchKernelGGL(chol_extract_y_kernel, dim3((d.N + 255) / 256), dim3(256), 0, s, d);
	for (int k = T - 1; k >= 0; k--) hipLaunchKernelGGL(chol_back_kernel, dim3(k > 0 ? k : 1), dim3(256), 0, s, d, k, x);
}

}  // namespace cubahip
