// ba_direct.hip -- the EXACT reduced solve: Hsc dxp = bsc by a SPARSE tile Cholesky factorisation (tile products on the matrix cores,
// v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32) plus the two triangular solves.  It stands where the reference calls cuSOLVER's
// sparse Cholesky (/root/reference/src/cuda_linear_solver.cpp:386-415: solve() is exact for any positive-definite Hsc and reports
// failure only on a non-positive pivot, :406-410, which CudaBundleAdjustmentImpl::optimize turns into rho = -1,
// /root/reference/src/cuda_bundle_adjustment.cpp:824-830), with the three phases of that solver:
//   ordering + symbolic analysis, once per structure  (ref: cuda_linear_solver.cpp:278-348 METIS / csrcholAnalysis)  -> sparse_chol_plan (host)
//   numeric factorisation, per solve                  (ref: :147-190 csrcholFactor)                                   -> schol_factor_level_kernel
//   triangular solves                                 (ref: :192-232 csrcholSolve)                                    -> folded into the factorisation (forward) /
//                                                                                                                        schol_back_level_kernel (backward)
//
// Granularity: SC_TP = 5 consecutive free poses of the internal (trajectory) order form a SEGMENT = one 32 x 32 tile row / column of the
// matrix (30 unknowns + 2 identity-padded ones).  The symbolic phase eliminates the segment graph by minimum degree with multiple
// elimination -- every round removes an independent set of segments whose degree is within `slack` of the minimum --, which keeps the
// fill of a trajectory with loop closures at band level (KITTI-00 shape: 2.5 k tiles = 21 MB, 10 000-pose graph: 19 k tiles = 154 MB)
// and, unlike a band order, leaves a SHALLOW elimination tree (45 / 94 levels instead of 267 / 2000 tile columns).  Columns of one level
// do not depend on one another: the numeric phase is ONE launch per level.
//
// Numeric phase, left-looking: the workgroup of tile (i, j) gathers A_ij - sum_k L_ik L_jk^T over the finished columns k that reach both
// rows (a fixed list in ascending k: no atomics, one writer per tile, results reproducible bit for bit) and, redundantly per workgroup
// of the column, the same for the diagonal tile; one wave then runs the 32-step elimination of the diagonal tile in REGISTERS (lane =
// row, pivot columns broadcast with v_readlane: no LDS round trips, no barriers) and the triangular solve of its own tile against it.  The
// diagonal tile's workgroup also carries the right-hand side (its gather doubles as the forward substitution).  The backward
// substitution walks the levels in reverse, one workgroup per column.
// Every sum has a fixed order: the solve is reproducible bit for bit like the rest of the path.

#include <algorithm>
#include <climits>
#include <cstring>

#include "ba_mfma.hpp"

namespace cubahip
{

// =====================================================================================================================================
// symbolic phase (host)
// =====================================================================================================================================
namespace
{

struct Elimination
{
	std::vector<int> order;                  // segment eliminated k-th
	std::vector<std::vector<int>> nbrs;      // its neighbours (segments) at that moment = the rows of tile column k
};

// Minimum degree on an explicit bit-matrix of the elimination graph (T <= a few thousand segments: T^2 / 8 bytes), multiple elimination:
// per round every segment whose degree is <= minimum + slack and that is not adjacent to a segment eliminated earlier in the round.
void eliminate_min_degree(int T, const std::vector<uint64_t>& adj0, int W, int slack, Elimination& out)
{
	std::vector<uint64_t> adj(adj0), blocked((size_t)W);
	std::vector<int> deg((size_t)T);
	std::vector<uint8_t> alive((size_t)T, 1);
	for (int v = 0; v < T; v++)
	{
		int d = 0;
		for (int w = 0; w < W; w++) d += __builtin_popcountll(adj[(size_t)v * W + w]);
		deg[v] = d;
	}
	out.order.clear(); out.nbrs.clear();
	out.order.reserve(T); out.nbrs.reserve(T);
	std::vector<int> nb;
	int left = T;
	while (left > 0)
	{
		int m = INT_MAX;
		for (int v = 0; v < T; v++) if (alive[v]) m = std::min(m, deg[v]);
		std::fill(blocked.begin(), blocked.end(), 0);
		for (int v = 0; v < T; v++)
		{
			if (!alive[v] || deg[v] > m + slack || ((blocked[v >> 6] >> (v & 63)) & 1)) continue;
			const uint64_t* rv = &adj[(size_t)v * W];
			nb.clear();
			for (int w = 0; w < W; w++)
				for (uint64_t x = rv[w]; x; x &= x - 1) nb.push_back(64 * w + __builtin_ctzll(x));
			out.order.push_back(v); out.nbrs.push_back(nb);
			alive[v] = 0; left--;
			for (int w = 0; w < W; w++) blocked[w] |= rv[w];
			for (int u : nb)
			{
				uint64_t* ru = &adj[(size_t)u * W];
				for (int w = 0; w < W; w++) ru[w] |= rv[w];
				ru[u >> 6] &= ~(1ULL << (u & 63));
				ru[v >> 6] &= ~(1ULL << (v & 63));
				int d = 0;
				for (int w = 0; w < W; w++) d += __builtin_popcountll(ru[w]);
				deg[u] = d;
			}
		}
	}
}

// tile (row position r, column position c) in column c's sorted row list; -1 if it is not part of the factor's pattern
inline int find_tile(const SparseCholPlan& p, int r, int c)
{
	const int* b = p.rowIdx.data() + p.colPtr[c] + 1;
	const int* e = p.rowIdx.data() + p.colPtr[c + 1];
	const int* it = std::lower_bound(b, e, r);
	return it != e && *it == r ? (int)(it - p.rowIdx.data()) : -1;
}

bool build_plan(int Pf, const int* rowptr, const int* colind, const Elimination& el, int slack, size_t maxTiles, SparseCholPlan& p)
{
	const int T = (int)el.order.size();
	p = SparseCholPlan();
	p.Pf = Pf; p.T = T; p.slack = slack;
	p.segOfPos = el.order;
	p.posOfSeg.assign(T, 0);
	for (int k = 0; k < T; k++) p.posOfSeg[el.order[k]] = k;
	p.colPtr.assign((size_t)T + 1, 0);
	size_t nTiles = 0;
	for (int k = 0; k < T; k++) { p.colPtr[k] = (int)nTiles; nTiles += 1 + el.nbrs[k].size(); if (nTiles > maxTiles) return false; }
	p.colPtr[T] = (int)nTiles; p.nTiles = (int)nTiles;
	p.rowIdx.resize(nTiles); p.colOfTile.resize(nTiles);
	for (int k = 0; k < T; k++)
	{
		int* r = p.rowIdx.data() + p.colPtr[k];
		r[0] = k;
		for (size_t a = 0; a < el.nbrs[k].size(); a++) r[1 + a] = p.posOfSeg[el.nbrs[k][a]];
		std::sort(r + 1, r + 1 + el.nbrs[k].size());
		for (int t = p.colPtr[k]; t < p.colPtr[k + 1]; t++) p.colOfTile[t] = k;
	}
	// row lists: the finished columns k < j that reach row j, ascending, with the tile (j, k); levels of the elimination tree
	std::vector<int> rowCnt((size_t)T + 1, 0);
	for (int k = 0; k < T; k++)
		for (int t = p.colPtr[k] + 1; t < p.colPtr[k + 1]; t++) rowCnt[p.rowIdx[t] + 1]++;
	for (int j = 0; j < T; j++) rowCnt[j + 1] += rowCnt[j];
	std::vector<int> rowTile((size_t)rowCnt[T]), fillPos(rowCnt.begin(), rowCnt.end() - 1);
	for (int k = 0; k < T; k++)
		for (int t = p.colPtr[k] + 1; t < p.colPtr[k + 1]; t++) rowTile[fillPos[p.rowIdx[t]]++] = t;
	std::vector<int> level((size_t)T, 0);
	int nLevels = T > 0 ? 1 : 0;
	for (int j = 0; j < T; j++)
	{
		int l = 0;
		for (int e = rowCnt[j]; e < rowCnt[j + 1]; e++) l = std::max(l, level[p.colOfTile[rowTile[e]]] + 1);
		level[j] = l; nLevels = std::max(nLevels, l + 1);
	}
	p.nLevels = nLevels;
	// gather lists
	const int zeroTile = p.nTiles;
	p.gPtr.assign(nTiles + 1, 0);
	size_t total = 0;
	for (int j = 0; j < T; j++)
	{
		const size_t len = ((size_t)(rowCnt[j + 1] - rowCnt[j]) + 3) & ~(size_t)3;       // (a multiple of 4: the kernel takes four entries per trip)
		for (int t = p.colPtr[j]; t < p.colPtr[j + 1]; t++) { p.gPtr[t] = (int)total; total += len; }
		if (total > ((size_t)1 << 29)) return false;
	}
	p.gPtr[nTiles] = (int)total;
	p.gather.assign(4 * total, 0);
	p.entries = 0;
	for (int j = 0; j < T; j++)
		for (int t = p.colPtr[j]; t < p.colPtr[j + 1]; t++)
		{
			int* g = p.gather.data() + 4 * (size_t)p.gPtr[t];
			const int i = p.rowIdx[t];
			int n = 0;
			for (int e = rowCnt[j]; e < rowCnt[j + 1]; e++, n++)
			{
				const int tb = rowTile[e], k = p.colOfTile[tb];
				const int ta = i == j ? tb : find_tile(p, i, k);
				g[4 * n] = ta >= 0 ? ta : zeroTile; g[4 * n + 1] = tb; g[4 * n + 2] = k; g[4 * n + 3] = 0;
				p.entries += ta >= 0 ? 2 : 1;
			}
			for (; n < p.gPtr[t + 1] - p.gPtr[t]; n++) { g[4 * n] = zeroTile; g[4 * n + 1] = zeroTile; g[4 * n + 2] = 0; g[4 * n + 3] = 0; }
		}
	// work lists by level (stable: tiles / columns in ascending order inside a level)
	p.lvlPtr.assign((size_t)nLevels + 1, 0); p.lvlColPtr.assign((size_t)nLevels + 1, 0);
	for (int j = 0; j < T; j++) { p.lvlPtr[level[j] + 1] += p.colPtr[j + 1] - p.colPtr[j]; p.lvlColPtr[level[j] + 1]++; }
	for (int l = 0; l < nLevels; l++) { p.lvlPtr[l + 1] += p.lvlPtr[l]; p.lvlColPtr[l + 1] += p.lvlColPtr[l]; }
	p.lvlTiles.resize(nTiles); p.lvlCols.resize(T);
	{
		std::vector<int> a(p.lvlPtr.begin(), p.lvlPtr.end() - 1), b(p.lvlColPtr.begin(), p.lvlColPtr.end() - 1);
		for (int j = 0; j < T; j++)
		{
			p.lvlCols[b[level[j]]++] = j;
			for (int t = p.colPtr[j]; t < p.colPtr[j + 1]; t++) p.lvlTiles[a[level[j]]++] = t;
		}
	}
	// one record per workgroup of the factorisation, in work-list order: everything its first instructions need in ONE scalar load
	p.wgRec.assign(8 * nTiles, 0);
	for (size_t w = 0; w < nTiles; w++)
	{
		const int t = p.lvlTiles[w], j = p.colOfTile[t];
		int* r = p.wgRec.data() + 8 * w;
		r[0] = t; r[1] = p.colPtr[j]; r[2] = j; r[3] = p.gPtr[t]; r[4] = p.gPtr[t + 1] - p.gPtr[t];
	}
	// destination of every block of the upper-triangular BSR storage
	p.blkTile.resize(rowptr[Pf]);
	for (int bi = 0; bi < Pf; bi++)
		for (int b = rowptr[bi]; b < rowptr[bi + 1]; b++)
		{
			const int bj = colind[b];
			const int pi = p.posOfSeg[bi / SC_TP], pj = p.posOfSeg[bj / SC_TP];
			int tile, tr = 0;
			if (pi == pj) tile = p.colPtr[pi];
			else if (pj > pi) tile = find_tile(p, pj, pi);
			else { tile = find_tile(p, pi, pj); tr = 1; }
			if (tile < 0) return false;          // (cannot happen: the factor's pattern contains the matrix's)
			p.blkTile[b] = tile | (tr << 30);
		}
	return true;
}

}  // namespace

double SparseCholPlan::secondsEstimate() const
{
	// ~11 us per level of the factorisation + ~6 us per level of the backward substitution (dependent launches), the gathered tile
	// products at the rate the memory-side cache feeds them, the fill
	return 17e-6 * nLevels + (double)entries * 1.5 * sizeof(Scalar) * SC_TT / 2.5e12 + (double)nTiles * sizeof(Scalar) * SC_TT / 3e12 + 20e-6;
}

bool sparse_chol_plan(int Pf, const int* rowptr, const int* colind, int slack, size_t maxTiles, SparseCholPlan& out)
{
	const int T = (Pf + SC_TP - 1) / SC_TP;
	if (T <= 0) { out = SparseCholPlan(); return true; }
	const int W = (T + 63) / 64;
	std::vector<uint64_t> adj((size_t)T * W, 0);
	for (int bi = 0; bi < Pf; bi++)
		for (int b = rowptr[bi]; b < rowptr[bi + 1]; b++)
		{
			const int a = bi / SC_TP, c = colind[b] / SC_TP;
			if (a == c) continue;
			adj[(size_t)a * W + (c >> 6)] |= 1ULL << (c & 63);
			adj[(size_t)c * W + (a >> 6)] |= 1ULL << (a & 63);
		}
	Elimination el;
	if (slack >= 0)
	{
		eliminate_min_degree(T, adj, W, slack, el);
		return build_plan(Pf, rowptr, colind, el, slack, maxTiles, out);
	}
	// automatic: a larger slack trades fill for a shallower tree; the cost model picks (deterministic: a function of the pattern only)
	bool have = false;
	static const int candidates[] = { 0, 2, 4, 8 };
	for (int s : candidates)
	{
		SparseCholPlan p;
		eliminate_min_degree(T, adj, W, s, el);
		if (!build_plan(Pf, rowptr, colind, el, s, maxTiles, p)) continue;
		if (!have || p.secondsEstimate() < out.secondsEstimate()) { out = std::move(p); have = true; }
	}
	return have;
}

// =====================================================================================================================================
// numeric phase (device)
// =====================================================================================================================================

// value of lane `lane` (a compile-time constant after unrolling) in every lane, through scalar registers
__device__ __forceinline__ Scalar lane_bcast(Scalar v, int lane)
{
	union { Scalar s; int w[sizeof(Scalar) / 4]; } u;
	u.s = v;
#pragma unroll
	for (int i = 0; i < (int)(sizeof(Scalar) / 4); i++) u.w[i] = __builtin_amdgcn_readlane(u.w[i], lane);
	return u.s;
}

// a wave-uniform pivot is usable: positive, normal, finite -- decided on the scalar unit from the bits of its high word
__device__ __forceinline__ bool pivot_ok(Scalar dj)
{
	union { Scalar s; int w[sizeof(Scalar) / 4]; } u;
	u.s = dj;
	const int hi = __builtin_amdgcn_readfirstlane(u.w[sizeof(Scalar) / 4 - 1]);
	return sizeof(Scalar) == 8 ? (unsigned)(hi - 0x00100000) < 0x7fe00000u : (unsigned)(hi - 0x00800000) < 0x7f000000u;
}

// upper-triangular BSR (damped, diagonal blocks full) -> the tiles of the permuted matrix's lower triangle
__global__ __launch_bounds__(256) void schol_fill_blocks_kernel(DeviceStructure st, DeviceSystem sys, SparseChol d)
{
	const int b = blockIdx.x * 4 + (threadIdx.x >> 6), e = threadIdx.x & 63;
	if (b >= st.nblk || e >= 36) return;
	const int bi = st.hsc_blkrow[b], bj = st.hsc_colind[b];        // bi <= bj; element (r, c) of the 6 x 6 block, column-major
	const int r = e % 6, c = e / 6;
	if (bi == bj && c < r) return;                                 // a diagonal block: one triangle
	const int tt = d.blkTile[b], tile = tt & 0x3fffffff;
	const int li = 6 * (bi % SC_TP) + r, lj = 6 * (bj % SC_TP) + c;
	const int row = (tt >> 30) ? li : lj, col = (tt >> 30) ? lj : li;
	d.tiles[(size_t)SC_TT * tile + col * SC_T + row] = sys.hsc[36 * (size_t)b + e];
}

// right-hand side into elimination order; identity on the padded unknowns
__global__ __launch_bounds__(256) void schol_fill_rhs_kernel(const Scalar* __restrict__ b, SparseChol d)
{
	const int idx = blockIdx.x * 256 + threadIdx.x;
	const int seg = idx >> 5, i = idx & 31;
	if (seg >= d.T) return;
	const int k = d.posOfSeg[seg];
	const int p = SC_TP * seg + i / 6;
	const bool valid = i < 6 * SC_TP && p < d.Pf;
	d.y[SC_T * (size_t)k + i] = valid ? b[6 * (size_t)p + i % 6] : Scalar(0);
	if (!valid) d.tiles[(size_t)SC_TT * d.colPtr[k] + i * SC_T + i] = Scalar(1);
}

void launch_sparse_chol_fill(const DeviceStructure& st, const DeviceSystem& sys, const SparseChol& d, hipStream_t s)
{
	(void)hipMemsetAsync(d.tiles, 0, sizeof(Scalar) * (size_t)SC_TT * ((size_t)d.nTiles + 1), s);
	if (st.nblk) hipLaunchKernelGGL(schol_fill_blocks_kernel, dim3((st.nblk + 3) / 4), dim3(256), 0, s, st, sys, d);
	hipLaunchKernelGGL(schol_fill_rhs_kernel, dim3((SC_T * d.T + 255) / 256), dim3(256), 0, s, sys.bsc, d);
	(void)hipMemsetAsync(d.fail, 0, sizeof(int), s);
}

// eight k-steps of a 16-row slab of a column-major tile, in the matrix-core operand layout: element (row0 + (lane & 15), 4 s + (lane >> 4))
__device__ __forceinline__ void load_slab(const Scalar* __restrict__ tile, int rowOff, Scalar out[8])
{
#pragma unroll
	for (int s = 0; s < 8; s++) out[s] = tile[128 * s + rowOff];
}

// One level of the elimination tree: workgroup = one tile (i, j) of a column j of the level.
//   phase 1 (four waves, one 16 x 16 quadrant each, transposed products so that a lane ends up with consecutive ROWS of a column):
//            D = A_jj - sum_k L_jk L_jk^T,  F = A_ij - sum_k L_ik L_jk^T;  the diagonal tile's workgroup instead: v = sum_k L_jk y_k
//   phase 2  D, F (or w = b_j - v) -> LDS
//   phase 3 (wave 0, lane = row): unscaled elimination of D in registers -- T_rc = L_rc L_cc --, then
//            off-diagonal tile:  L_ij = F L_jj^-T (row-wise triangular solve against the registers), stored plain and transposed
//            diagonal tile:      y_j = L_jj^-1 w, L_jj (mirrored into both triangles, in the diagonal tile's slot of tilesT) and 1 / L_cc stored; failure flag on a non-positive
//                                pivot (the factorisation carries on with pivot 1 so that nothing downstream sees a NaN)
__global__ __launch_bounds__(256) void schol_factor_level_kernel(SparseChol d, int first)
{
	__shared__ Scalar Ds[SC_T][SC_T + 1];
	__shared__ Scalar Fs[SC_T][SC_T + 1];
	__shared__ Scalar ws[SC_T];
	__shared__ Scalar pivS[2 * SC_T];
	const int4* __restrict__ R = reinterpret_cast<const int4*>(d.wgRec) + 2 * (size_t)(first + blockIdx.x);
	const int4 rec0 = R[0];
	const int t = rec0.x, t0 = rec0.y, j = rec0.z, g0 = rec0.w, len = R[1].x;
	const bool diag = t == t0;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wi = wv >> 1, wj = wv & 1;
	const int off = (lane >> 4) * SC_T + (lane & 15);
	const int offI = 16 * wi + off, offJ = 16 * wj + off;
	MfmaAcc aD = mfma_zero(), aF = mfma_zero();
	const int4* __restrict__ G = reinterpret_cast<const int4*>(d.gather) + g0;
	// Four entries per trip: all their operand loads are in flight before the first product (a trip costs one memory round trip -- the
	// tiles were written by other compute units, they come from the memory-side cache --, not one per entry), and the next trip's
	// entries are fetched under this trip's loads.
	if (len > 0)
	{
		int4 cur[4] = { G[0], G[1], G[2], G[3] };
		const bool yl = wj == 0 && (lane & 15) == 0;
		for (int e = 0; e < len; e += 4)
		{
			const int en = e + 4 < len ? e + 4 : e;
			const int4 nxt[4] = { G[en], G[en + 1], G[en + 2], G[en + 3] };
			Scalar rr[4][8], cc[4][8], ff[4][8];
			if (diag)
			{
#pragma unroll
				for (int u = 0; u < 4; u++)
				{
					const Scalar* L = d.tiles + (size_t)SC_TT * cur[u].y;
					load_slab(L, offI, rr[u]); load_slab(L, offJ, cc[u]);
#pragma unroll
					for (int s8 = 0; s8 < 8; s8++) ff[u][s8] = yl ? d.y[SC_T * (size_t)cur[u].z + 4 * s8 + (lane >> 4)] : Scalar(0);
				}
#pragma unroll
				for (int u = 0; u < 4; u++)
#pragma unroll
					for (int s8 = 0; s8 < 8; s8++) { aD = mfma_16x16x4(cc[u][s8], rr[u][s8], aD); aF = mfma_16x16x4(ff[u][s8], rr[u][s8], aF); }
			}
			else
			{
#pragma unroll
				for (int u = 0; u < 4; u++)
				{
					const Scalar* L = d.tiles + (size_t)SC_TT * cur[u].y;
					load_slab(L, offI, rr[u]); load_slab(L, offJ, cc[u]);
					load_slab(d.tiles + (size_t)SC_TT * cur[u].x, offI, ff[u]);
				}
#pragma unroll
				for (int u = 0; u < 4; u++)
#pragma unroll
					for (int s8 = 0; s8 < 8; s8++) { aD = mfma_16x16x4(cc[u][s8], rr[u][s8], aD); aF = mfma_16x16x4(cc[u][s8], ff[u][s8], aF); }
			}
#pragma unroll
			for (int u = 0; u < 4; u++) cur[u] = nxt[u];
		}
	}
	// lane l holds element (row 16 wi + (l & 15), column 16 wj + mfma_row(l, q)) of its wave's quadrant
	{
		const Scalar* A0 = d.tiles + (size_t)SC_TT * t0;
		const Scalar* At = d.tiles + (size_t)SC_TT * t;
		const int row = 16 * wi + (lane & 15);
#pragma unroll
		for (int q = 0; q < 4; q++)
		{
			const int col = 16 * wj + mfma_row(lane, q);
			Ds[row][col] = A0[col * SC_T + row] - mfma_get(aD, q);
			if (!diag) Fs[row][col] = At[col * SC_T + row] - mfma_get(aF, q);
		}
		if (diag && wj == 0 && lane < 16) ws[row] = d.y[SC_T * (size_t)j + row] - mfma_get(aF, 0);
	}
	__syncthreads();
	// (the serial part takes one wave: a different one from workgroup to workgroup, so that the workgroups sharing a compute unit do not
	// queue their chains on one SIMD)
	if (wv != (int)(blockIdx.x & 3)) return;
	const int r = lane & 31;
	Scalar T[SC_T];
#pragma unroll
	for (int c = 0; c < SC_T; c++) T[c] = Ds[r][c];
	// (pivot k and its reciprocal are wave-uniform: they reach the lane that owns row k through LDS -- every lane stores the same number
	// -- instead of through 32 compare masks; positivity is tested on the scalar unit)
	int bad = 0;
#pragma unroll
	for (int jj = 0; jj < SC_T; jj++)
	{
		Scalar dj = lane_bcast(T[jj], jj);
		if (!pivot_ok(dj)) { bad = 1; dj = Scalar(1); }
		const Scalar rj = fast_rcp(dj);
		pivS[jj] = dj; pivS[SC_T + jj] = rj;
		Scalar lr = T[jj] * rj;
		// (all broadcasts of the step, then all multiply-adds: interleaved one by one, every multiply-add waits for its scalar pair;
		// the empty asm statements pin that order)
		Scalar sc[SC_T];
#pragma unroll
		for (int c = jj + 1; c < SC_T; c++) sc[c] = lane_bcast(T[jj], c);
#pragma unroll
		for (int c = jj + 1; c < SC_T; c++) asm volatile("" : "+s"(sc[c]));
		asm volatile("" : "+v"(lr));
#pragma unroll
		for (int c = jj + 1; c < SC_T; c++) T[c] -= lr * sc[c];
	}
	const Scalar sv = Scalar(1) / sqrt(pivS[r]);       // 1 / L_rr
	const Scalar rjv = pivS[SC_T + r];                 // 1 / T_rr
	// (the solves below broadcast the same register / lane pairs as the elimination did: without this fence the compiler keeps all 496
	// broadcast values alive in scalar registers across both loops and spills a thousand of them)
#pragma unroll
	for (int c = 0; c < SC_T; c++) asm volatile("" : "+v"(T[c]));
	if (!diag)
	{
		Scalar a[SC_T];
#pragma unroll
		for (int c = 0; c < SC_T; c++) a[c] = Fs[r][c];
#pragma unroll
		for (int m = 0; m < SC_T; m++)
		{
			asm volatile("" : "+v"(T[m]), "+v"(a[m]));        // (keeps step m's broadcasts behind step m - 1: hoisted, they spill)
			Scalar z = a[m] * lane_bcast(rjv, m);
			Scalar sc[SC_T];
#pragma unroll
			for (int c = m + 1; c < SC_T; c++) sc[c] = lane_bcast(T[m], c);
#pragma unroll
			for (int c = m + 1; c < SC_T; c++) asm volatile("" : "+s"(sc[c]));
			asm volatile("" : "+v"(z));
#pragma unroll
			for (int c = m + 1; c < SC_T; c++) a[c] -= z * sc[c];
		}
#pragma unroll
		for (int m = 0; m < SC_T; m++) a[m] *= lane_bcast(sv, m);
		if (lane < SC_T)
		{
			Scalar* Lt = d.tiles + (size_t)SC_TT * t;
			Scalar* Ltt = d.tilesT + (size_t)SC_TT * t;
#pragma unroll
			for (int c = 0; c < SC_T; c++) { Lt[c * SC_T + r] = a[c]; Ltt[r * SC_T + c] = a[c]; }
		}
		return;
	}
	// the diagonal tile's workgroup: forward substitution of the right-hand side, then L_jj itself
	Scalar w = ws[r], yfin = Scalar(0);
#pragma unroll
	for (int m = 0; m < SC_T; m++)
	{
		const Scalar wm = lane_bcast(w, m);
		const Scalar zm = wm * lane_bcast(rjv, m);
		if (r == m) yfin = wm * sv;
		if (r > m) w -= T[m] * zm;
	}
	if (lane < SC_T)
	{
		d.y[SC_T * (size_t)j + r] = yfin;
		d.rinv[SC_T * (size_t)j + r] = sv;
	}
	// (into the transposed array's slot of the diagonal tile: the other workgroups of this column may still be reading A_jj)
	Scalar* L0 = d.tilesT + (size_t)SC_TT * t0;
#pragma unroll
	for (int c = 0; c < SC_T; c++)
	{
		const Scalar v = T[c] * lane_bcast(sv, c);
		if (lane < SC_T && c <= r) { L0[c * SC_T + r] = v; L0[r * SC_T + c] = v; }
	}
	if (bad != 0 && lane == 0) *d.fail = 1;
}

// One level of the backward substitution (levels in reverse): workgroup = one column j.
//   v = sum over the column's off-diagonal tiles of L_ij^T x_i  (wave w takes tiles w, w + 4, ..; lane = (column c, row half): the
//   transposed copy makes the loads of a row contiguous across the lanes), summed over halves and waves in a fixed order;
//   x_j = L_jj^-T (y_j - v): 32 dependent steps in one wave, lane m holding column m of L_jj (= row m of the mirrored tile)
__global__ __launch_bounds__(256) void schol_back_level_kernel(SparseChol d, int first)
{
	__shared__ Scalar part[4][2][SC_T];
	const int j = d.lvlCols[first + blockIdx.x];
	const int t0 = d.colPtr[j], s = d.colPtr[j + 1] - t0 - 1;
	const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, h = lane >> 5;
	const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
	Scalar acc = 0;
	for (int a = wv; a < s; a += 4)
	{
		const int t = t0 + 1 + a;
		const int i = d.rowIdx[t];
		const Scalar* __restrict__ Lt = d.tilesT + (size_t)SC_TT * t + (16 * h) * SC_T + c;
		const Scalar* __restrict__ xi = d.y + SC_T * (size_t)i + 16 * h;
#pragma unroll
		for (int rr = 0; rr < 16; rr++) acc += Lt[rr * SC_T] * xi[rr];
	}
	part[wv][h][c] = acc;
	__syncthreads();
	if (wv != 0) return;
	const Scalar v = ((part[0][0][c] + part[0][1][c]) + (part[1][0][c] + part[1][1][c])) + ((part[2][0][c] + part[2][1][c]) + (part[3][0][c] + part[3][1][c]));
	Scalar w = d.y[SC_T * (size_t)j + c] - v;
	const Scalar sOwn = d.rinv[SC_T * (size_t)j + c];
	const Scalar* __restrict__ L0 = d.tilesT + (size_t)SC_TT * t0 + c;
	Scalar Lc[SC_T];
#pragma unroll
	for (int cc = 0; cc < SC_T; cc++) Lc[cc] = L0[cc * SC_T];          // element (row c, column cc) of the mirrored tile = L[cc][c] for cc >= c
	Scalar xfin = 0;
#pragma unroll
	for (int cc = SC_T - 1; cc >= 0; cc--)
	{
		const Scalar xc = lane_bcast(w * sOwn, cc);
		if (c == cc) xfin = xc;
		if (c < cc) w -= Lc[cc] * xc;
	}
	if (lane < SC_T) d.y[SC_T * (size_t)j + c] = xfin;
}

__global__ __launch_bounds__(256) void schol_extract_kernel(SparseChol d, Scalar* __restrict__ x)
{
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= 6 * d.Pf) return;
	const int p = idx / 6, c = idx % 6;
	x[idx] = d.y[SC_T * (size_t)d.posOfSeg[p / SC_TP] + 6 * (p % SC_TP) + c];
}

void launch_sparse_chol_solve(const SparseChol& d, const SparseCholPlan& plan, Scalar* x, hipStream_t s)
{
	for (int l = 0; l < plan.nLevels; l++)
	{
		const int n = plan.lvlPtr[l + 1] - plan.lvlPtr[l];
		if (n > 0) hipLaunchKernelGGL(schol_factor_level_kernel, dim3(n), dim3(256), 0, s, d, plan.lvlPtr[l]);
	}
	for (int l = plan.nLevels - 1; l >= 0; l--)
	{
		const int n = plan.lvlColPtr[l + 1] - plan.lvlColPtr[l];
		if (n > 0) hipLaunchKernelGGL(schol_back_level_kernel, dim3(n), dim3(256), 0, s, d, plan.lvlColPtr[l]);
	}
	if (d.Pf > 0) hipLaunchKernelGGL(schol_extract_kernel, dim3((6 * d.Pf + 255) / 256), dim3(256), 0, s, d, x);
}

}  // namespace cubahip
