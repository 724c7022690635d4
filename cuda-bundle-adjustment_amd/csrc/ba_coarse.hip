// ba_coarse.hip -- coarse level of the two-level preconditioner: assembly of P^T A P from the reduced matrix and its explicit
// inverse by a blocked Gauss-Jordan sweep (tile products on the matrix cores: v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32),
// plus the fp32 copy the iteration kernels read.  Stands where the reference calls cuSOLVER's csrcholFactor
// (/root/reference/src/cuda_linear_solver.cpp:147-232); runs on a second stream under the PCG of an earlier trial.

#include "ba_mfma.hpp"

namespace cubahip
{

// ---------------------------------------------------------------------------------------------------
// Two-level preconditioner  M^-1 = blockdiag(A)^-1 + P (P^T A P)^-1 P^T.
// P is piecewise constant over aggregates of `agg` consecutive free poses (6 coarse dof per aggregate):
// keyframe chains are stiff along the trajectory, and these are exactly the slowly converging drift
// modes of block-Jacobi CG (1887 -> 226 iterations on the KITTI-00-shaped system at lambda_9).
// The coarse matrix is dense and small (6*nc <= ~1500), so its explicit inverse is formed on the device by
// a blocked Gauss-Jordan sweep (SPD => no pivoting) and applied as a dense mat-vec inside the PCG.
// ---------------------------------------------------------------------------------------------------
// one 64-lane wave per non-empty pair of aggregates (I,J): lanes 0..35 own one element of the 6x6 blocks and add the fine
// blocks of the list in a fixed order (no atomics => the coarse matrix, its inverse and hence the whole CG are
// reproducible).  With the linear coarse functions (cl = 2) a fine block (i,j) goes into four coarse blocks with the
// weights 1, w_j, w_i, w_i w_j.
__global__ __launch_bounds__(256) void coarse_assemble_kernel(DeviceStructure st, DeviceSystem sys, Scalar* Ac, int Pf)
{
	// one workgroup per pair of aggregates: its four waves take every fourth entry of the list, the partial sums are
	// added in wave order
	__shared__ Scalar sh[4][4][36];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int cb = blockIdx.x;
	const int r = lane % 6, c = (lane / 6) % 6;
	const int CD = 6 * sys.cl, Nc = CD * sys.nc;
	Scalar acc[2][2] = { { 0, 0 }, { 0, 0 } };
	const int p1 = st.cb_ptr[cb + 1];
	for (int p = st.cb_ptr[cb] + wv; p < p1; p += 16)
	{
		int b[4]; Scalar v[4], wi[4], wj[4];
#pragma unroll
		for (int m = 0; m < 4; m++)
		{
			const int q = min(p + 4 * m, p1 - 1);
			b[m] = st.cb_blk[q];
			wi[m] = sys.cl == 2 ? st.cb_wi[q] : Scalar(0); wj[m] = sys.cl == 2 ? st.cb_wj[q] : Scalar(0);
		}
#pragma unroll
		for (int m = 0; m < 4; m++) v[m] = p + 4 * m < p1 ? sys.hsc[36 * (size_t)(b[m] & 0x7fffffff) + (b[m] < 0 ? r * 6 + c : c * 6 + r)] : Scalar(0);
#pragma unroll
		for (int m = 0; m < 4; m++) { acc[0][0] += v[m]; acc[0][1] += v[m] * wj[m]; acc[1][0] += wi[m] * v[m]; acc[1][1] += wi[m] * v[m] * wj[m]; }
	}
	if (lane < 36)
	{
		sh[wv][0][lane] = acc[0][0]; sh[wv][1][lane] = acc[0][1]; sh[wv][2][lane] = acc[1][0]; sh[wv][3][lane] = acc[1][1];
	}
	__syncthreads();
	if (threadIdx.x >= 36 * 4) return;
	const int ab = threadIdx.x / 36, el = threadIdx.x - 36 * ab, a = ab >> 1, bb = ab & 1;
	if (a >= sys.cl || bb >= sys.cl) return;
	Scalar v = ((sh[0][ab][el] + sh[1][ab][el]) + sh[2][ab][el]) + sh[3][ab][el];
	const int rr = el % 6, cc = el / 6;
	if (ab == 3 && st.cb_I[cb] == st.cb_J[cb] && st.cb_I[cb] == sys.nc - 1 && Pf % sys.agg == 1) v = rr == cc ? Scalar(1) : Scalar(0);
	Ac[(size_t)(st.cb_J[cb] * CD + 6 * bb + cc) * Nc + st.cb_I[cb] * CD + 6 * a + rr] = v;
}

constexpr int GJ_B = 32;      // pivot block width of the Gauss-Jordan sweep = output tile edge

// Inverse of the 32 x 32 block a 256-thread workgroup holds as Dcur[r][c] in LDS (Dnext: identity outside [0, bk)^2): 2x2 block
// pivots -- 16 dependent steps instead of 32, one reciprocal (v_rcp_f64 + two Newton steps: the pivots of an SPD matrix are
// positive and well scaled) per step -- ping-ponging between the two LDS copies so that one barrier per step is enough.
// Returns the array holding the result (callers synchronise before reading it: the last step ends with a barrier).
// The chain is instruction issue + latency of one wave per SIMD (every element changes in every step), so the thread -> element
// map is chosen for the fewest instructions: thread (c, rb) = (tid & 31, tid >> 5) owns rows rb + 8u of column c, hence one
// (W D[P][c]) pair per thread instead of one per element, and whether a row is a pivot row is uniform over a wave (rows rb + 8u,
// rb in {2w, 2w + 1}: the pivot rows p, p + 1 are the element u = p / 8 of wave w = (p mod 8) / 2) -- a scalar branch, no selects.
__device__ __forceinline__ Scalar (*gj_pivot_inverse(Scalar (*Dcur)[GJ_B + 1], Scalar (*Dnext)[GJ_B + 1], int tid, int bk))[GJ_B + 1]
{
	const int c = tid & 31, rb = tid >> 5;
	const int wv2 = 2 * __builtin_amdgcn_readfirstlane(tid >> 6);
	(void)bk;          // a short block arrives padded with the identity, on which every step below is an exact no-op: always 16 steps,
	                   // fully unrolled -- every LDS address is then an immediate offset and every pivot-position test a constant
	Scalar d[4];
#pragma unroll
	for (int u = 0; u < 4; u++) d[u] = Dcur[rb + 8 * u][c];
#pragma unroll
	for (int p = 0; p < GJ_B; p += 2)
	{
		const Scalar a00 = Dcur[p][p], a01 = Dcur[p][p + 1], a10 = Dcur[p + 1][p], a11 = Dcur[p + 1][p + 1];
		const Scalar m0 = Dcur[p][c], m1 = Dcur[p + 1][c];
		Scalar mi0[4], mi1[4];
#pragma unroll
		for (int u = 0; u < 4; u++) { mi0[u] = Dcur[rb + 8 * u][p]; mi1[u] = Dcur[rb + 8 * u][p + 1]; }
		const Scalar rdet = fast_rcp(a00 * a11 - a01 * a10);
		const Scalar w00 = a11 * rdet, w01 = -a01 * rdet, w10 = -a10 * rdet, w11 = a00 * rdet;     // W = inverse of the 2x2 pivot block
		const Scalar t0 = w00 * m0 + w01 * m1, t1 = w10 * m0 + w11 * m1;                           // (W D[P][c])
		const bool jp = c == p || c == p + 1;
		const Scalar wA = c == p ? w00 : w01, wB = c == p ? w10 : w11;                              // column c - p of W
		const bool pivotWave = (p & 7) == wv2;                                                        // (scalar)
#pragma unroll
		for (int u = 0; u < 4; u++)
		{
			const int r = rb + 8 * u;
			Scalar v;
			if (pivotWave && u == (p >> 3))          // rows p (rb even) and p + 1 (rb odd)
			{
				const Scalar vRow = r == p ? t0 : t1;
				const Scalar vBoth = r == p ? wA : wB;
				v = jp ? vBoth : vRow;
			}
			else
			{
				const Scalar vGen = d[u] - (mi0[u] * t0 + mi1[u] * t1);
				const Scalar vCol = -(mi0[u] * wA + mi1[u] * wB);
				v = jp ? vCol : vGen;
			}
			d[u] = v; Dnext[r][c] = v;
		}
		__syncthreads();
		Scalar (*tmp)[GJ_B + 1] = Dcur; Dcur = Dnext; Dnext = tmp;
	}
	return Dcur;
}

// Inverse of the first pivot block (rows / columns [0, bk)) of the sweep -> pivOut[c * 32 + r]; every later pivot block is
// inverted by the step before it (below).
__device__ __forceinline__ void dense_gj_first_pivot_body(const Scalar* __restrict__ src, int n, int bk, Scalar* __restrict__ pivOut)
{
	__shared__ Scalar D[GJ_B][GJ_B + 1];
	__shared__ Scalar D2[GJ_B][GJ_B + 1];
	const int tid = threadIdx.x, r = tid & 31, cb = tid >> 5;
	Scalar dv[4];
#pragma unroll
	for (int u = 0; u < 4; u++) dv[u] = src[(size_t)min(cb + 8 * u, n - 1) * n + min(r, n - 1)];
#pragma unroll
	for (int u = 0; u < 4; u++)
	{
		const int c = cb + 8 * u;
		dv[u] = (r < bk && c < bk) ? dv[u] : (r == c ? Scalar(1) : Scalar(0));     // identity padding of a short block
		D[r][c] = dv[u];
		D2[r][c] = r == c ? Scalar(1) : Scalar(0);
	}
	__syncthreads();
	Scalar (*res)[GJ_B + 1] = gj_pivot_inverse(D, D2, tid, bk);
#pragma unroll
	for (int u = 0; u < 4; u++) pivOut[(cb + 8 * u) * GJ_B + r] = res[r][cb + 8 * u];
}

__global__ __launch_bounds__(256) void dense_gj_first_pivot_kernel(const Scalar* __restrict__ src, int n, int bk, Scalar* __restrict__ pivOut)
{
	dense_gj_first_pivot_body(src, n, bk, pivOut);
}

// batched sweep (cuba_hip_optimize_batch): the coarse matrices of several graphs, one launch per step for all of them
__global__ __launch_bounds__(256) void dense_gj_first_pivot_batch_kernel(const GjJob* __restrict__ jobs)
{
	const GjJob& j = jobs[blockIdx.x];
	dense_gj_first_pivot_body(j.buf[0], j.n, min(GJ_B, j.n), j.piv[0]);
}

// One blocked step of the symmetric sweep with pivot rows/cols [p0, p0+bk), p0 a multiple of GJ_B:  with D = A_pp^-1,
//     A_ij <- A_ij - A_ip D A_pj,   A_ip <- A_ip D,   A_pj <- D A_pj,   A_pp <- -D          (i, j != p)
// applied to EVERY block row / column, swept before or not.  A symmetric matrix stays symmetric under it, so only the tiles
// on and above the diagonal are computed and stored (tile (i, j), i <= j; a tile below the diagonal is read as the transpose
// of its mirror image): half the tiles, half the bytes of the plain Gauss-Jordan sweep this replaced.  After the last step
// the upper tiles hold -A^-1.  One 256-thread workgroup per 32x32 output tile; thread (r, cb) owns the elements
// (r, cb + 8u), u = 0..3, of every 32x32 array.  For coarse matrices of ~1000 unknowns the step time is latency, not flops
// (n/32 dependent launches), so:
//   * every global load of the kernel is issued before the first use (clamped addresses, selected afterwards);
//   * the inverse of the pivot block comes in ready-made (pivIn): the NEXT pivot block -- tile (p0/32 + 1, p0/32 + 1) is
//     final for this purpose once step p0 has updated it -- is inverted inside this launch (look-ahead) by ONE extra
//     workgroup (blockIdx.x = 0, dispatched first) that recomputes just that tile and goes straight into the 16-step
//     chain: 15.9 us per step when every workgroup inverted the pivot block itself, ~11 now;
//   * four LDS arrays (the chain's second array reuses an operand array): four workgroups per CU;
//   * the two 32x32x32 products run on the matrix cores.
__device__ __forceinline__ void dense_gj_step_body(const Scalar* __restrict__ src, Scalar* __restrict__ dst, int n, int p0, int bk,
	const Scalar* __restrict__ pivIn, Scalar* __restrict__ pivOut)
{
	__shared__ Scalar D[GJ_B][GJ_B + 1];
	__shared__ Scalar Apj[GJ_B][GJ_B + 1];
	__shared__ Scalar R[GJ_B][GJ_B + 1];
	__shared__ Scalar F[GJ_B][GJ_B + 1];
	const int tid = threadIdx.x;
	TRACE_DECL
	TRACE_MARK();
	const int pNext = p0 + GJ_B;                            // look-ahead: the tile (pNext, pNext) is the next pivot block
	const bool ahead = blockIdx.x == 0;                     // the look-ahead workgroup; workgroups 1.. are the upper tiles, column by column
	if (ahead && pNext >= n) return;
	int ti, tj;
	if (ahead) ti = tj = pNext / GJ_B;
	else
	{
		const int t = blockIdx.x - 1;                       // t = tj (tj + 1) / 2 + ti, ti <= tj
		tj = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
		while ((tj + 1) * (tj + 2) / 2 <= t) tj++;
		while (tj * (tj + 1) / 2 > t) tj--;
		ti = t - tj * (tj + 1) / 2;
	}
	const int i0 = ti * GJ_B, j0 = tj * GJ_B;
	const int r = tid & 31, cb = tid >> 5;
	const bool rowTile = i0 == p0, colTile = j0 == p0;      // this tile lies in the pivot rows / pivot columns
	const bool fT = i0 > p0, gT = j0 < p0;                  // A[i, p] / A[p, j] lies below the diagonal: read the mirror tile, transposed
	Scalar dv[4], av[4], fv[4], sv[4];
#pragma unroll
	for (int u = 0; u < 4; u++)
	{
		const int c = cb + 8 * u;
		// element (r, c) of a tile stored as is: A[row0 + r, col0 + c] = src[(col0 + c) n + row0 + r]; of the mirror tile the same
		// expression with the two offsets swapped, landing in LDS as element (c, r)
		const size_t gRow = (size_t)min((gT ? j0 : p0) + r, n - 1), gCol = (size_t)min((gT ? p0 : j0) + c, n - 1);
		const size_t fRow = (size_t)min((fT ? p0 : i0) + r, n - 1), fCol = (size_t)min((fT ? i0 : p0) + c, n - 1);
		const size_t gi = (size_t)min(i0 + r, n - 1), gj = (size_t)min(j0 + c, n - 1);
		dv[u] = pivIn[c * GJ_B + r];   // D[r][c] = inverse of the pivot block A[p0.., p0..]
		av[u] = src[gCol * n + gRow];  // Apj = A[p0.., j0..]
		fv[u] = src[fCol * n + fRow];  // F   = A[i0.., p0..]
		sv[u] = src[gj * n + gi];      // own tile
	}
#pragma unroll
	for (int u = 0; u < 4; u++)
	{
		const int c = cb + 8 * u;
		D[r][c] = dv[u];
		// Apj[k][m] is live for k < bk, j0 + m < n;  F[m][k] for k < bk, i0 + m < n
		if (gT) Apj[c][r] = (c < bk && j0 + r < n) ? av[u] : Scalar(0);
		else Apj[r][c] = (r < bk && j0 + c < n) ? av[u] : Scalar(0);
		if (fT) F[c][r] = (r < bk && i0 + c < n) ? fv[u] : Scalar(0);
		else F[r][c] = (c < bk && i0 + r < n) ? fv[u] : Scalar(0);
	}
	__syncthreads();
	TRACE_MARK();
	// The two 32 x 32 x 32 tile products on the matrix cores: wave w owns the 16 x 16 output tile (w >> 1, w & 1), eight
	// v_mfma_f64_16x16x4_f64 k-steps each (operands straight from LDS, one number per lane: A[i = lane & 15][k = lane >> 4],
	// B[k = lane >> 4][j = lane & 15]).  This is the one GEMM-shaped piece of the whole path.
	const int wv = tid >> 6, lane = tid & 63;
	const int wi = wv >> 1, wj = wv & 1;
	// R = Dinv * Apj (not needed by the tiles of the pivot columns)
	if (!colTile)
	{
		MfmaAcc acc = mfma_zero();
#pragma unroll
		for (int s4 = 0; s4 < GJ_B; s4 += 4)
			acc = mfma_16x16x4(D[16 * wi + (lane & 15)][s4 + (lane >> 4)], Apj[s4 + (lane >> 4)][16 * wj + (lane & 15)], acc);
#pragma unroll
		for (int q = 0; q < 4; q++)
		{
			const int rr = 16 * wi + mfma_row(lane, q);
			R[rr][16 * wj + (lane & 15)] = rr < bk ? mfma_get(acc, q) : Scalar(0);
		}
	}
	__syncthreads();
	Scalar out[4];
	if (rowTile && colTile)
	{
#pragma unroll
		for (int u = 0; u < 4; u++) out[u] = -dv[u];
	}
	else if (rowTile)
	{
#pragma unroll
		for (int u = 0; u < 4; u++) out[u] = R[r][cb + 8 * u];
	}
	else
	{
		Scalar (*B)[GJ_B + 1] = colTile ? D : R;               // pivot columns: F Dinv; elsewhere: S - F R
		MfmaAcc acc = mfma_zero();
#pragma unroll
		for (int s4 = 0; s4 < GJ_B; s4 += 4)
			acc = mfma_16x16x4(F[16 * wi + (lane & 15)][s4 + (lane >> 4)], B[s4 + (lane >> 4)][16 * wj + (lane & 15)], acc);
		// back to the thread -> element map of the loads / stores through LDS (Apj is free by now)
		__syncthreads();
#pragma unroll
		for (int q = 0; q < 4; q++) Apj[16 * wi + mfma_row(lane, q)][16 * wj + (lane & 15)] = mfma_get(acc, q);
		__syncthreads();
#pragma unroll
		for (int u = 0; u < 4; u++) out[u] = colTile ? Apj[r][cb + 8 * u] : sv[u] - Apj[r][cb + 8 * u];
	}
	TRACE_MARK();
	if (!ahead)
	{
#pragma unroll
		for (int u = 0; u < 4; u++)
		{
			const int gi = i0 + r, gj = j0 + cb + 8 * u;
			if (gi < n && gj < n) dst[(size_t)gj * n + gi] = out[u];
		}
		TRACE_MARK();
		TRACE_FLUSH(2, (blockIdx.x * 4 + (threadIdx.x >> 6)) % 8000);
		return;
	}
	// look-ahead: invert the next pivot block for the next launch (the chain's second array: Apj -- the operands are through)
	const int bkN = min(GJ_B, n - pNext);
	__syncthreads();
#pragma unroll
	for (int u = 0; u < 4; u++)
	{
		const int c = cb + 8 * u;
		out[u] = (r < bkN && c < bkN) ? out[u] : (r == c ? Scalar(1) : Scalar(0));     // identity padding of a short last block
		D[r][c] = out[u];
		Apj[r][c] = r == c ? Scalar(1) : Scalar(0);
	}
	__syncthreads();
	Scalar (*res)[GJ_B + 1] = gj_pivot_inverse(D, Apj, tid, bkN);
#pragma unroll
	for (int u = 0; u < 4; u++) pivOut[(cb + 8 * u) * GJ_B + r] = res[r][cb + 8 * u];
	TRACE_MARK();
	TRACE_FLUSH(2, 8000 + (threadIdx.x >> 6));              // (kept apart: the last launch of a sweep has no look-ahead workgroup)
}

__global__ __launch_bounds__(256) void dense_gj_step_kernel(const Scalar* __restrict__ src, Scalar* __restrict__ dst, int n, int p0, int bk,
	const Scalar* __restrict__ pivIn, Scalar* __restrict__ pivOut)
{
	dense_gj_step_body(src, dst, n, p0, bk, pivIn, pivOut);
}

__global__ __launch_bounds__(256) void dense_gj_step_batch_kernel(const GjJob* __restrict__ jobs, int step)
{
	const GjJob& j = jobs[blockIdx.y];
	const int p0 = step * GJ_B;
	if (p0 >= j.n || (int)blockIdx.x > j.tiles * (j.tiles + 1) / 2) return;
	dense_gj_step_body(j.buf[step & 1], j.buf[(step & 1) ^ 1], j.n, p0, min(GJ_B, j.n - p0), j.piv[step & 1], j.piv[(step & 1) ^ 1]);
}

// jobs[m] on the device; every job's matrix starts in its buf[0] and ends in buf[steps & 1] (as launch_dense_inverse leaves it)
void launch_dense_inverse_batch(const GjJob* jobs, int m, int nMax, hipStream_t s)
{
	if (m <= 0 || nMax <= 0) return;
	const int tiles = (nMax + GJ_B - 1) / GJ_B;
	hipLaunchKernelGGL(dense_gj_first_pivot_batch_kernel, dim3(m), dim3(256), 0, s, jobs);
	for (int step = 0; step < tiles; step++)
		hipLaunchKernelGGL(dense_gj_step_batch_kernel, dim3(1 + tiles * (tiles + 1) / 2, m), dim3(256), 0, s, jobs, step);
}

// the first two launches of launch_coarse_setup alone (zero + assemble; `assembled` recorded behind them)
void launch_coarse_assemble(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, Scalar* work0, hipStream_t s)
{
	const int Nc = 6 * sys.cl * sys.nc;
	(void)hipMemsetAsync(work0, 0, sizeof(Scalar) * (size_t)Nc * Nc, s);
	if (st.nCb) hipLaunchKernelGGL(coarse_assemble_kernel, dim3(st.nCb), dim3(256), 0, s, st, sys, work0, g.Pf);
}

// symmetric sweep: work0 holds the matrix on entry (its upper triangle is what is read); returns the buffer (work0 or work1)
// whose tiles on and above the diagonal hold -A^-1 -- launch_coarse_finish / launch_coarse_to_fp32 turn that into what the
// iteration kernels read.  pivots: 2 x 32 x 32 numbers of scratch (the inverse of the current / of the next pivot block)
Scalar* launch_dense_inverse(Scalar* work0, Scalar* work1, int n, Scalar* pivots, hipStream_t s)
{
	Scalar* src = work0; Scalar* dst = work1;
	const int tiles = (n + GJ_B - 1) / GJ_B;
	Scalar* pivIn = pivots; Scalar* pivOut = pivots + GJ_B * GJ_B;
	hipLaunchKernelGGL(dense_gj_first_pivot_kernel, dim3(1), dim3(256), 0, s, src, n, min(GJ_B, n), pivIn);
	for (int p0 = 0; p0 < n; p0 += GJ_B)
	{
		hipLaunchKernelGGL(dense_gj_step_kernel, dim3(1 + tiles * (tiles + 1) / 2), dim3(256), 0, s, src, dst, n, p0, min(GJ_B, n - p0), pivIn, pivOut);
		Scalar* tmp = src; src = dst; dst = tmp;
		tmp = pivIn; pivIn = pivOut; pivOut = tmp;
	}
	return src;
}

// swept buffer (upper triangle = -A^-1) -> full symmetric +A^-1 in dst (dst = swept: in place)
// (no __restrict__: the overlapped schedule runs it in place, dst == swept -- every load of a tile reaches LDS before the barrier and
// workgroups own disjoint tile pairs, so the in-place use is well defined only WITHOUT a no-alias promise)
__global__ __launch_bounds__(256) void coarse_finish_kernel(const Scalar* swept, Scalar* dst, int n)
{
	__shared__ Scalar tile[32][33];
	// one workgroup per 32 x 32 tile on or above the diagonal (2D grid, the tiles below return at once): coalesced reads of
	// the tile, coalesced writes of the tile and of its mirror image
	if (blockIdx.x > blockIdx.y) return;
	const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32, r = threadIdx.x & 31, cb = threadIdx.x >> 5;
#pragma unroll
	for (int u = 0; u < 4; u++)
	{
		const int c = cb + 8 * u, gi = i0 + r, gj = j0 + c;
		Scalar v = (gi < n && gj < n) ? -swept[(size_t)gj * n + gi] : Scalar(0);
		if (blockIdx.x == blockIdx.y && r > c) v = (gi < n && gj < n) ? -swept[(size_t)gi * n + gj] : Scalar(0);   // diagonal tile: its upper half on both sides
		tile[r][c] = v;
	}
	__syncthreads();
#pragma unroll
	for (int u = 0; u < 4; u++)
	{
		const int c = cb + 8 * u;
		if (i0 + r < n && j0 + c < n) dst[(size_t)(j0 + c) * n + i0 + r] = tile[r][c];
		if (blockIdx.x != blockIdx.y && j0 + r < n && i0 + c < n) dst[(size_t)(i0 + c) * n + j0 + r] = tile[c][r];
	}
}

void launch_coarse_finish(const Scalar* swept, Scalar* dst, int n, hipStream_t s)
{
	const int tiles = (n + 31) / 32;
	if (n > 0) hipLaunchKernelGGL(coarse_finish_kernel, dim3(tiles, tiles), dim3(256), 0, s, swept, dst, n);
}

// Assemble P^T A P from the (already damped) reduced matrix and sweep it; returns the buffer (work0 or work1) whose upper
// triangle holds -(P^T A P)^-1 (launch_dense_inverse).
Scalar* launch_coarse_setup(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, Scalar* work0, Scalar* work1, hipStream_t s, hipEvent_t assembled)
{
	const int Nc = 6 * sys.cl * sys.nc;
	(void)hipMemsetAsync(work0, 0, sizeof(Scalar) * (size_t)Nc * Nc, s);
	if (st.nCb) hipLaunchKernelGGL(coarse_assemble_kernel, dim3(st.nCb), dim3(256), 0, s, st, sys, work0, g.Pf);
	if (assembled) (void)hipEventRecord(assembled, s);      // from here on the sweep no longer reads the reduced matrix
	return launch_dense_inverse(work0, work1, Nc, sys.gj_pivots, s);
}

// swept buffer (upper triangle = -A^-1, column-major) -> +A^-1 in fp32, exactly symmetric, rows padded with zeros to ld
__global__ __launch_bounds__(256) void coarse_to_fp32_kernel(const Scalar* __restrict__ src, float* __restrict__ dst, int n, int ld)
{
	const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (t >= (size_t)n * ld) return;
	const int row = (int)(t / ld), j = (int)(t - (size_t)row * ld);
	dst[t] = j < n ? -(float)src[(size_t)max(row, j) * n + min(row, j)] : 0.0f;
}

void launch_coarse_to_fp32(const Scalar* src, float* dst, int n, hipStream_t s)
{
	const int ld = (n + 3) & ~3;
	const size_t total = (size_t)n * ld;
	if (total) hipLaunchKernelGGL(coarse_to_fp32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, dst, n, ld);
}

}  // namespace cubahip

#ifdef CUBA_HIP_TRACE
extern "C" int cuba_hip_debug_read_trace_coarse(unsigned long long* out)   // this translation unit's 3 x 8192 x 8 timestamps (100 MHz)
{
	return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cubahip::cuba_trace_buf), sizeof(unsigned long long) * 3 * 8192 * 8);
}
#endif
