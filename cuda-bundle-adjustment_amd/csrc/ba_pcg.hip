// ba_pcg.hip -- the reduced solve: block preconditioned conjugate gradients on Hsc dxp = bsc, in the seat of cuSOLVER's sparse
// Cholesky (/root/reference/src/cuda_linear_solver.cpp:147-232, 301-335).  Two launches per iteration -- SpMV on a row-ordered
// copy of the upper-triangular BSR matrix (pcg_spmv_kernel / pcg_spmv_row_kernel), then update + restriction + two-level
// preconditioner + r.z in one kernel (pcg2_fused_kernel; pcg_update_kernel in block-Jacobi-only mode) --, all CG scalars on
// the device as per-workgroup partial sums in fixed order, iterations replayed as hipGraphs (graph_add_pcg_chunk).

#include "ba_device.hpp"

namespace cubahip
{

// ---------------------------------------------------------------------------------------------------
// Block-Jacobi preconditioned conjugate gradients on Hsc dxp = bsc  (replaces cuSOLVER csrchol,
// /root/reference/src/cuda_linear_solver.cpp:147-232,301-335).  Scalars never leave the device:
//   rz[k] = r_k . z_k, pq[k] = p_k . A p_k live in NSLOT partial-sum slots per iteration;
//   every workgroup re-derives alpha / beta / the stop test from them, so a finished solve turns the
//   remaining queued launches into no-ops without a host round trip.
// ---------------------------------------------------------------------------------------------------
// ROWCOPY: also store the damped diagonal block into the row-ordered copy of the matrix (the fused launch: its expand workgroups
// leave the diagonal entries alone, because this body rewrites the diagonal blocks they would read)
template <bool ROWCOPY>
__device__ __forceinline__ void pcg_setup_body(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, Scalar lambda, int bid, int nb)
{
	// one pose per thread, 64 poses per workgroup (only its first wave works): every load / store of a thread is 288 bytes from its
	// neighbour's, so the set-up is the address path of the CUs it runs on -- spread over four times as many of them
	const int i = threadIdx.x < PCG_SETUP_POSES ? bid * PCG_SETUP_POSES + (int)threadIdx.x : g.Pf;
	lambda = launch_lambda(sys, lambda);
	Scalar rz = 0;
	if (i < g.Pf)
	{
		Scalar* blk = sys.hsc + 36 * (size_t)st.hsc_rowptr[i];
		Scalar A[36], Ai[36];
#pragma unroll
		for (int c = 0; c < 6; c++)
#pragma unroll
			for (int r = 0; r <= c; r++)
			{
				Scalar v = blk[c * 6 + r];
				if (r == c) v += lambda;
				A[c * 6 + r] = v;
				A[r * 6 + c] = v;
			}
#pragma unroll
		for (int k = 0; k < 36; k++) blk[k] = A[k];   // full symmetric diagonal block, damping included
		if (ROWCOPY && !sys.upper)
		{
			// adjacency of a row = its lower neighbours, then its own blocks, the diagonal one first
			const int pos = (st.adj_ptr[i + 1] - st.adj_ptr[i]) - (st.hsc_rowptr[i + 1] - st.hsc_rowptr[i]);
			if (pos < 20 * st.ell_m)
			{
				Scalar* dst = sys.hrow + 36 * ((size_t)i * st.ell_m * 20 + pos);
#pragma unroll
				for (int k = 0; k < 36; k++) dst[k] = A[k];      // (symmetric: row-major = column-major)
			}
		}
		if (!spd6_inverse(A, Ai)) *sys.fail = 1;
#pragma unroll
		for (int k = 0; k < 36; k++) sys.minv[36 * (size_t)i + k] = Ai[k];
		Scalar rr[6];
#pragma unroll
		for (int k = 0; k < 6; k++) rr[k] = sys.bsc[6 * (size_t)i + k];
#pragma unroll
		for (int r = 0; r < 6; r++)
		{
			Scalar z = 0;
#pragma unroll
			for (int c = 0; c < 6; c++) z += Ai[c * 6 + r] * rr[c];
			sys.r[6 * (size_t)i + r] = rr[r];
			sys.z[6 * (size_t)i + r] = z;   // overwritten by pcg2_fused_kernel when the coarse level is on
			sys.xp[6 * (size_t)i + r] = 0;
			sys.p0[6 * (size_t)i + r] = 0;
			sys.p1[6 * (size_t)i + r] = 0;
			rz += rr[r] * z;
		}
	}
	rz = wave_sum(rz);
	if (sys.agg == 0)   // block-Jacobi only: this kernel produces r0.z0 -> slot 0 and, as "r_k.z_k for k = 0", ring slot 1
	{
		if (threadIdx.x == 0)
		{
			const Scalar s2 = rz;
			sys.rz[bid] = s2;
			sys.rz[sys.rzStride + bid] = s2;
		}
		for (int t = nb + bid * 256 + threadIdx.x; t < sys.nrz; t += nb * 256) sys.rz[sys.rzStride + t] = 0;
	}
	if (bid == 0 && threadIdx.x == 0) { *sys.iters = 0; *sys.done = 0; *sys.kbase = 0; }
}

__global__ __launch_bounds__(256) void pcg_setup_kernel(DeviceGraph g, DeviceStructure st, DeviceSystem sys, Scalar lambda)
{
	pcg_setup_body<false>(g, st, sys, lambda, blockIdx.x, gridDim.x);
}

void launch_pcg_setup(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, Scalar lambda, hipStream_t s)
{
	if (g.Pf > 0) hipLaunchKernelGGL(pcg_setup_kernel, dim3((g.Pf + PCG_SETUP_POSES - 1) / PCG_SETUP_POSES), dim3(256), 0, s, g, st, sys, lambda);
}

__device__ __forceinline__ bool pcg_active(const DeviceSystem& sys, int k, int maxIter, Scalar tol2, int lane, Scalar& rzk)
{
	if (*sys.done) return false;
	rzk = wave_sum(load_parts(rz_slot(sys, k), rz_count(sys, k), lane));
	const Scalar rz0 = wave_sum(load_parts(sys.rz, sys.nrz0, lane));
	const bool on = k < maxIter && *sys.fail == 0 && rzk > tol2 * rz0 && rzk == rzk;
	if (!on && blockIdx.x == 0 && threadIdx.x == 0) { *sys.done = 1; if (!(rzk == rzk)) *sys.fail = 3; }   // NaN: a failed solve, not a converged one
	return on;
}

// A(k): p_k = z_k + beta p_{k-1} (recomputed on the fly for the neighbour rows), q = A p_k, pq slot k = p.q partials.
// A p_k is formed as A z_k + beta A p_{k-1} from two accumulators, so beta enters only after the last FMA.  The kernel has
// two memory round trips: (1) the fixed-width index rows and the reduction partials (addresses known at launch: slots
// depend on the chunk-local k & 3), (2) all operands of the row in one batch.  k's parity equals the parity of the
// chunk-local argument (chunk lengths are multiples of 4), so the p ping-pong needs no load either.
// spmv_entry: rows wider than the fixed-width part read their remaining entries from the upper-triangular storage.
__device__ __forceinline__ void spmv_entry(const DeviceStructure& st, const DeviceSystem& sys, const Scalar* pold, int a, int rr,
	Scalar& accz, Scalar& accp)
{
	const int bi = st.adj_blk[a];
	const int j = st.adj_col[a];
	const Scalar* B = sys.hsc + 36 * (size_t)(bi & 0x7fffffff);
	const int sr = bi < 0 ? 6 : 1, sc = bi < 0 ? 1 : 6;   // transposed read of the stored upper block for the lower half
	Scalar av[6], zv[6], pv[6];
#pragma unroll
	for (int c = 0; c < 6; c++)
	{
		av[c] = B[rr * sr + c * sc];
		zv[c] = sys.z[6 * (size_t)j + c];
		pv[c] = pold[6 * (size_t)j + c];
	}
#pragma unroll
	for (int c = 0; c < 6; c++) { accz += av[c] * zv[c]; accp += av[c] * pv[c]; }
}

typedef Scalar Scalar2 __attribute__((ext_vector_type(2)));

// Row-ordered copy of the reduced matrix for the SpMV: entry (row, m, slot) of the fixed-width adjacency rows holds its
// 6x6 block as seen from that row (transposed for the lower half), row-major, so that lane (slot, rr) reads the six
// numbers it needs as three aligned 16-byte loads (half as many load instructions as element-wise strided reads of
// the upper storage; the per-CU load path, not bandwidth, limits this kernel at KITTI-00 size).
// FUSED: runs beside pcg_setup_body in one launch, which damps and symmetrises the diagonal blocks meanwhile -- they are left to it.
template <bool FUSED>
__device__ __forceinline__ void hsc_expand_body(const DeviceStructure& st, const DeviceSystem& sys, size_t total, size_t bid)
{
	const size_t t = bid * 256 + threadIdx.x;
	if (t >= total) return;
	const size_t slot = t / 36;
	const int e = (int)(t - 36 * slot);
	const int2 en = st.ell[slot];
	if (en.y < 0) return;
	const int rr = e / 6, c = e - 6 * rr;
	const Scalar* B = sys.hsc + 36 * (size_t)(en.x & 0x7fffffff);
	if (FUSED && en.y == (int)(slot / ((size_t)st.ell_m * 20))) return;       // the row's own diagonal block: written by the set-up body
	sys.hrow[t] = B[en.x < 0 ? rr * 6 + c : c * 6 + rr];
}

__global__ __launch_bounds__(256) void hsc_expand_kernel(DeviceStructure st, DeviceSystem sys, size_t total)
{
	hsc_expand_body<false>(st, sys, total, blockIdx.x);
}

// pcg_setup (a handful of workgroups, one 6x6 inverse per thread: 11 us of latency) and the row-ordered copy (9 us of streaming)
// in one launch
// (+ optionally, in the last nCopy workgroups, the copy of a freshly inverted coarse matrix into the buffer the iteration graphs read)
__global__ __launch_bounds__(256) void pcg_setup_expand_kernel(DeviceGraph g, DeviceStructure st, DeviceSystem sys, Scalar lambda, int nSetup, size_t total,
	unsigned nExpand, const Scalar2* __restrict__ copySrc, Scalar2* __restrict__ copyDst, size_t copyPairs)
{
	if ((int)blockIdx.x < nSetup) pcg_setup_body<true>(g, st, sys, lambda, blockIdx.x, nSetup);
	else if (blockIdx.x < nSetup + nExpand) hsc_expand_body<true>(st, sys, total, blockIdx.x - nSetup);
	else
	{
		const size_t stride = (size_t)(gridDim.x - nSetup - nExpand) * 256;
		for (size_t i = (size_t)(blockIdx.x - nSetup - nExpand) * 256 + threadIdx.x; i < copyPairs; i += stride) copyDst[i] = copySrc[i];
	}
}

__global__ __launch_bounds__(256) void pcg_setup_expand_batch_kernel(const BatchEntry* __restrict__ tab)
{
	const BatchEntry& e = tab[blockIdx.y];
	const BatchTrial& t = e.t;
	if (blockIdx.x >= t.setupGrid) return;
	if ((int)blockIdx.x < t.nSetup) pcg_setup_body<true>(e.g, e.st, e.sys, Scalar(-1), blockIdx.x, t.nSetup);
	else if (blockIdx.x < t.nSetup + t.nExpand) hsc_expand_body<true>(e.st, e.sys, t.expandTotal, blockIdx.x - t.nSetup);
	else
	{
		const Scalar2* __restrict__ src = reinterpret_cast<const Scalar2*>(t.copySrc);
		Scalar2* __restrict__ dst = reinterpret_cast<Scalar2*>(t.copyDst);
		const size_t stride = (size_t)(t.setupGrid - t.nSetup - t.nExpand) * 256;
		for (size_t i = (size_t)(blockIdx.x - t.nSetup - t.nExpand) * 256 + threadIdx.x; i < t.copyPairs; i += stride) dst[i] = src[i];
	}
}

void launch_hsc_expand(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, hipStream_t s)
{
	const size_t total = (size_t)g.Pf * st.ell_m * 20 * 36;
	if (total) hipLaunchKernelGGL(hsc_expand_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, st, sys, total);
}

void launch_pcg_setup_expand(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, Scalar lambda, hipStream_t s,
	const Scalar* copySrc, Scalar* copyDst, size_t copyCount)
{
	if (g.Pf <= 0) return;
	const size_t total = sys.upper ? 0 : (size_t)g.Pf * st.ell_m * 20 * 36;      // (the upper-triangle iteration reads the BSR storage itself)
	const int nSetup = (g.Pf + PCG_SETUP_POSES - 1) / PCG_SETUP_POSES;
	const unsigned nExpand = (unsigned)((total + 255) / 256);
	const size_t pairs = copySrc ? copyCount / 2 : 0;            // (coarse dimensions are even)
	const unsigned nCopy = pairs ? (unsigned)std::min<size_t>(1024, (pairs + 255) / 256) : 0;
	hipLaunchKernelGGL(pcg_setup_expand_kernel, dim3(nSetup + nExpand + nCopy), dim3(256), 0, s, g, st, sys, lambda, nSetup, total, nExpand,
		reinterpret_cast<const Scalar2*>(copySrc), reinterpret_cast<Scalar2*>(copyDst), pairs);
}

// launch_pcg_setup_expand's grid composition for one graph of a batch
void batch_fill_setup(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, const Scalar* copySrc, Scalar* copyDst, size_t copyCount, BatchTrial& t)
{
	t.expandTotal = sys.upper ? 0 : (size_t)g.Pf * st.ell_m * 20 * 36;
	t.nSetup = (g.Pf + PCG_SETUP_POSES - 1) / PCG_SETUP_POSES;
	t.nExpand = (unsigned)((t.expandTotal + 255) / 256);
	t.copyPairs = copySrc ? copyCount / 2 : 0;
	const unsigned nCopy = t.copyPairs ? (unsigned)std::min<size_t>(1024, (t.copyPairs + 255) / 256) : 0;
	t.copySrc = copySrc; t.copyDst = copyDst;
	t.setupGrid = t.nSetup + t.nExpand + nCopy;
}

void launch_batch_setup(const BatchEntry* tab, int n, unsigned gridMax, hipStream_t s)
{
	if (gridMax) hipLaunchKernelGGL(pcg_setup_expand_batch_kernel, dim3(gridMax, n), dim3(256), 0, s, tab);
}

// N entries of one lane at once: all 9 N (16-byte) loads are issued before the first use. Padding entries (column -1)
// read z / p of column 0 against a zero matrix entry.
template <int N>
__device__ __forceinline__ void spmv_batch(const DeviceSystem& sys, const Scalar* pold, const int2 (&e)[3], const Scalar* Arow, int rr, Scalar& accz, Scalar& accp)
{
	Scalar2 av[N][3], zv[N][3], pv[N][3];
#pragma unroll
	for (int n = 0; n < N; n++)
	{
		const size_t j = e[n].y >= 0 ? e[n].y : 0;
		const Scalar2* A2 = reinterpret_cast<const Scalar2*>(Arow + (size_t)n * (20 * 36) + 6 * rr);
		const Scalar2* z2 = reinterpret_cast<const Scalar2*>(sys.z + 6 * j);
		const Scalar2* p2 = reinterpret_cast<const Scalar2*>(pold + 6 * j);
		// the matrix entry of a padding slot is never fetched (the lanes are masked off for these loads: ~25 % of the
		// fixed-width slots are padding, and at S2M / G4M size their bytes show); z / p of column 0 are cache hits
		const bool on = e[n].y >= 0;
#pragma unroll
		for (int c = 0; c < 3; c++) { av[n][c] = on ? A2[c] : Scalar2{ 0, 0 }; zv[n][c] = z2[c]; pv[n][c] = p2[c]; }
	}
#pragma unroll
	for (int n = 0; n < N; n++)
	{
#pragma unroll
		for (int c = 0; c < 3; c++)
		{
			const Scalar a0 = av[n][c].x, a1 = av[n][c].y;
			accz += a0 * zv[n][c].x; accp += a0 * pv[n][c].x;
			accz += a1 * zv[n][c].y; accp += a1 * pv[n][c].y;
		}
	}
}

// MIN_WAVES = 4 caps the kernel at 128 VGPRs (a few spills): worth it only when the rows need more than one round of
// waves at occupancy 3 -- at S2M / G4M size the kernel is bound by waves in flight x latency -- not at KITTI-00 size.
// ROWS = block rows per workgroup (2 or 4): more rows per workgroup mean fewer row-sum partials for the two-level kernel
// to add up (large graphs), fewer rows mean more workgroups to spread over the CUs (small graphs).
template <int ROWS>
__device__ __forceinline__ void pcg_spmv_body(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, int k, int maxIter, Scalar tol2)
{
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int half = wv & 1, lr = wv >> 1;          // the two waves of a row take 10 of its 20 entry slots each
	const Scalar* pold = (k & 1) ? sys.p1 : sys.p0;
	Scalar* pnew = (k & 1) ? sys.p0 : sys.p1;
	const int row = blockIdx.x * ROWS + lr;
	TRACE_DECL
	TRACE_MARK();
	// scalar loads, consumed at the very end (k is chunk-local here; the absolute number only enters the tests)
	const int kb_v = vector_load_flag(sys.kbase);
	const int failed_v = vector_load_flag(sys.fail) | vector_load_flag(sys.done);
	const Scalar s_k = load_parts(rz_slot(sys, k), sys.nrz, lane);
	const Scalar s_0 = load_parts(sys.rz, sys.nrz0, lane);
	const Scalar s_m = load_parts(rz_slot(sys, k - 1), sys.nrz, lane);

	// every index comes from an address known at launch (fixed-width rows): one memory round trip for the indices and
	// the reduction scalars, one for all matrix / vector operands of the row
	const bool rowOn = row < g.Pf;
	const int slot = half * 10 + lane / 6, rr = lane % 6;
	int2 e[3];
	int a0 = 0, a1 = 0;
#pragma unroll
	for (int m = 0; m < 3; m++)
		e[m] = (rowOn && lane < 60 && m < st.ell_m) ? st.ell[((size_t)row * st.ell_m + m) * 20 + slot] : int2{ 0, -1 };
	if (rowOn && st.ell_over && lane < 60) { a0 = st.adj_ptr[row] + 20 * st.ell_m + slot; a1 = st.adj_ptr[row + 1]; }
	Scalar accz = 0, accp = 0, zi = 0, pi_old = 0;
	if (rowOn && half == 0 && lane < 6)
	{
		zi = sys.z[6 * (size_t)row + lane];
		pi_old = pold[6 * (size_t)row + lane];
	}
	TRACE_MARK();
	k += __builtin_amdgcn_readfirstlane(kb_v);
	const int failed = __builtin_amdgcn_readfirstlane(failed_v);
	const Scalar rzk = to_uniform(wave_sum(s_k)), rz0 = to_uniform(wave_sum(s_0)), rzm = to_uniform(wave_sum(s_m));
	if (!(k < maxIter && failed == 0 && rzk > tol2 * rz0 && rzk == rzk))   // uniform over the grid
	{
		if (blockIdx.x == 0 && threadIdx.x == 0) { *sys.done = 1; if (!(rzk == rzk)) *sys.fail = 3; }   // NaN: reported as a failed solve
		return;
	}
	const Scalar beta = k > 0 ? rzk / rzm : Scalar(0);
	if (rowOn)
	{
		int cnt = 0;                       // wave-uniform: entries of the fullest slot
#pragma unroll
		for (int m = 0; m < 3; m++) cnt += __any(e[m].y >= 0) ? 1 : 0;
		const Scalar* Arow = sys.hrow + 36 * ((size_t)row * st.ell_m * 20 + slot);     // entry (row, m, slot) at + m * 20 * 36
		if (cnt == 3) spmv_batch<3>(sys, pold, e, Arow, rr, accz, accp);
		else if (cnt == 2) spmv_batch<2>(sys, pold, e, Arow, rr, accz, accp);
		else if (cnt == 1) spmv_batch<1>(sys, pold, e, Arow, rr, accz, accp);
		for (int a = a0; a < a1; a += 20) spmv_entry(st, sys, pold, a, rr, accz, accp);   // rows wider than the fixed part
	}
	TRACE_MARK();
	// fold the 10 slots onto lanes 0..5
	accz += __shfl_down(accz, 30); accp += __shfl_down(accp, 30);
	Scalar tz = accz, tp = accp;
	tz += __shfl_down(accz, 6);  tp += __shfl_down(accp, 6);
	tz += __shfl_down(accz, 12); tp += __shfl_down(accp, 12);
	tz += __shfl_down(accz, 18); tp += __shfl_down(accp, 18);
	tz += __shfl_down(accz, 24); tp += __shfl_down(accp, 24);
	__shared__ Scalar other[ROWS][12];
	__shared__ Scalar qrow[ROWS][6];
	__shared__ Scalar part[ROWS];
	if (half == 1 && lane < 6) { other[lr][lane] = tz; other[lr][6 + lane] = tp; }
	__syncthreads();
	TRACE_MARK();
	Scalar dot = 0;
	if (half == 0)
	{
		Scalar q = 0;
		if (row < g.Pf && lane < 6)
		{
			tz += other[lr][lane]; tp += other[lr][6 + lane];
			const Scalar pi = zi + beta * pi_old;
			q = tz + beta * tp;
			pnew[6 * (size_t)row + lane] = pi;
			sys.ap[6 * (size_t)row + lane] = q;
			dot = pi * q;
		}
		if (lane < 6) qrow[lr][lane] = q;
		dot = wave_sum(dot);
		if (lane == 0) part[lr] = dot;
	}
	__syncthreads();
	if (threadIdx.x < 6 * sys.cl)   // (weighted) row sums of q over this workgroup's rows: the two-level kernel builds P^T q from these
	{
		const int a = threadIdx.x / 6, c = threadIdx.x - 6 * a;
		Scalar s2 = 0;
#pragma unroll
		for (int w = 0; w < ROWS; w++)
			s2 += (a == 0 ? Scalar(1) : agg_weight(blockIdx.x * ROWS + w, sys.agg, g.Pf)) * qrow[w][c];
		if (sys.qpart && sys.agg > 0)
		{
			// layout [m][coarse unknown], m = position of this workgroup inside its aggregate: the two-level kernel then reads
			// consecutive addresses across a wave for every m
			const int per = sys.agg / ROWS, J = blockIdx.x / per, m = blockIdx.x - J * per;
			sys.qpart[(size_t)m * (6 * sys.cl * sys.nc) + 6 * sys.cl * J + threadIdx.x] = s2;
		}
	}
	if (threadIdx.x == 64)
	{
		Scalar s2 = 0;
#pragma unroll
		for (int w = 0; w < ROWS; w++) s2 += part[w];
		pq_slot(sys, k)[blockIdx.x] = s2;
	}
	TRACE_MARK();
	TRACE_FLUSH(0, blockIdx.x * 2 * ROWS + wv);
}

template <int ROWS, int MIN_WAVES>
__global__ __launch_bounds__(128 * ROWS, MIN_WAVES) void pcg_spmv_kernel(DeviceGraph g, DeviceStructure st, DeviceSystem sys, int k, int maxIter, Scalar tol2)
{
	pcg_spmv_body<ROWS>(g, st, sys, k, maxIter, tol2);
}

// Batched form (cuba_hip_optimize_batch): blockIdx.y selects the graph, whose kernel arguments come from a device table instead of the
// kernel-argument segment; workgroups beyond the graph's own grid return at once.  Same body, same arithmetic, same bits.
template <int ROWS, int MIN_WAVES>
__global__ __launch_bounds__(128 * ROWS, MIN_WAVES) void pcg_spmv_batch_kernel(const BatchEntry* __restrict__ tab, int k, Scalar tol2)
{
	const BatchEntry& e = tab[blockIdx.y];
	if ((int)blockIdx.x >= e.gridSpmv) return;
	pcg_spmv_body<ROWS>(e.g, e.st, e.sys, k, e.maxIter, tol2);
}

// One wave per block row (large graphs).  With two waves per row S2M / G4M need 10 000 / 20 000 waves of ~5 KB each, i.e.
// 2.5 / 5 rounds of resident waves whose life is two dependent round trips + a barrier: the launch is bound by
// wave slots x latency, not by bytes.  Here lane (slot, r2) = (lane / 3, lane % 3) takes block rows 2 r2 and 2 r2 + 1 of one of
// the 20 entry slots: half the waves, twice the bytes per wave, no cross-wave exchange.  One level of the fixed-width row per
// step at FIVE waves per SIMD (82 VGPRs; five workgroups per CU = 1280 resident: S2M's 1250 workgroups are one round) beats
// two levels in flight at four (120 VGPRs): S2M 16.4 -> 14.7 us, G4M 25.1 -> 23.5 us (profiles/r04s_spmv_row_occupancy.txt).
__device__ __forceinline__ void spmv_level(const DeviceSystem& sys, const Scalar* pold, int2 e, const Scalar* Arow, int r2,
	Scalar& az0, Scalar& ap0, Scalar& az1, Scalar& ap1)
{
	Scalar2 a0v[3], a1v[3], zv[3], pv[3];
	const size_t j = e.y >= 0 ? e.y : 0;
	const Scalar2* A2 = reinterpret_cast<const Scalar2*>(Arow + 12 * r2);
	const Scalar2* z2 = reinterpret_cast<const Scalar2*>(sys.z + 6 * j);
	const Scalar2* p2 = reinterpret_cast<const Scalar2*>(pold + 6 * j);
	const bool on = e.y >= 0;          // padding slots: matrix entry not fetched
#pragma unroll
	for (int c = 0; c < 3; c++)
	{
		a0v[c] = on ? A2[c] : Scalar2{ 0, 0 }; a1v[c] = on ? A2[3 + c] : Scalar2{ 0, 0 };
		zv[c] = z2[c]; pv[c] = p2[c];
	}
#pragma unroll
	for (int c = 0; c < 3; c++)
	{
		az0 += a0v[c].x * zv[c].x; ap0 += a0v[c].x * pv[c].x;
		az0 += a0v[c].y * zv[c].y; ap0 += a0v[c].y * pv[c].y;
		az1 += a1v[c].x * zv[c].x; ap1 += a1v[c].x * pv[c].x;
		az1 += a1v[c].y * zv[c].y; ap1 += a1v[c].y * pv[c].y;
	}
}

template <int ROWS>
__global__ __launch_bounds__(64 * ROWS, 5) void pcg_spmv_row_kernel(DeviceGraph g, DeviceStructure st, DeviceSystem sys, int k, int maxIter, Scalar tol2)
{
	const int lane = threadIdx.x & 63, lr = threadIdx.x >> 6;
	const Scalar* pold = (k & 1) ? sys.p1 : sys.p0;
	Scalar* pnew = (k & 1) ? sys.p0 : sys.p1;
	const int row = blockIdx.x * ROWS + lr;
	const int kb_v = vector_load_flag(sys.kbase);
	const int failed_v = vector_load_flag(sys.fail) | vector_load_flag(sys.done);
	const Scalar s_k = load_parts(rz_slot(sys, k), sys.nrz, lane);
	const Scalar s_0 = load_parts(sys.rz, sys.nrz0, lane);
	const Scalar s_m = load_parts(rz_slot(sys, k - 1), sys.nrz, lane);
	const bool rowOn = row < g.Pf;
	const int slot = lane / 3, r2 = lane - 3 * slot;
	int2 e[3];
	int a0 = 0, a1 = 0;
#pragma unroll
	for (int m = 0; m < 3; m++)
		e[m] = (rowOn && lane < 60 && m < st.ell_m) ? st.ell[((size_t)row * st.ell_m + m) * 20 + slot] : int2{ 0, -1 };
	if (rowOn && st.ell_over && lane < 60) { a0 = st.adj_ptr[row] + 20 * st.ell_m + slot; a1 = st.adj_ptr[row + 1]; }
	Scalar zi = 0, pi_old = 0;
	if (rowOn && lane < 6)
	{
		zi = sys.z[6 * (size_t)row + lane];
		pi_old = pold[6 * (size_t)row + lane];
	}
	k += __builtin_amdgcn_readfirstlane(kb_v);
	const int failed = __builtin_amdgcn_readfirstlane(failed_v);
	const Scalar rzk = to_uniform(wave_sum(s_k)), rz0 = to_uniform(wave_sum(s_0)), rzm = to_uniform(wave_sum(s_m));
	if (!(k < maxIter && failed == 0 && rzk > tol2 * rz0 && rzk == rzk))   // uniform over the grid
	{
		if (blockIdx.x == 0 && threadIdx.x == 0) { *sys.done = 1; if (!(rzk == rzk)) *sys.fail = 3; }   // NaN: reported as a failed solve
		return;
	}
	const Scalar beta = k > 0 ? rzk / rzm : Scalar(0);
	Scalar az0 = 0, ap0 = 0, az1 = 0, ap1 = 0;
	if (rowOn)
	{
		int cnt = 0;                       // wave-uniform: entries of the fullest slot
#pragma unroll
		for (int m = 0; m < 3; m++) cnt += __any(e[m].y >= 0) ? 1 : 0;
		const Scalar* Arow = sys.hrow + 36 * ((size_t)row * st.ell_m * 20 + slot);     // entry (row, m, slot) at + m * 20 * 36
		if (cnt >= 1) spmv_level(sys, pold, e[0], Arow, r2, az0, ap0, az1, ap1);
		if (cnt >= 2) spmv_level(sys, pold, e[1], Arow + 20 * 36, r2, az0, ap0, az1, ap1);
		if (cnt == 3) spmv_level(sys, pold, e[2], Arow + 2 * (20 * 36), r2, az0, ap0, az1, ap1);
		for (int a = a0; a < a1; a += 20)          // rows wider than the fixed part: from the upper-triangular storage
		{
			const int bi = st.adj_blk[a];
			const size_t j = st.adj_col[a];
			const Scalar* B = sys.hsc + 36 * (size_t)(bi & 0x7fffffff);
			const int sr = bi < 0 ? 6 : 1, sc = bi < 0 ? 1 : 6;   // transposed read of the stored upper block for the lower half
#pragma unroll
			for (int c = 0; c < 6; c++)
			{
				const Scalar zc = sys.z[6 * j + c], pc = pold[6 * j + c];
				const Scalar b0 = B[(2 * r2) * sr + c * sc], b1 = B[(2 * r2 + 1) * sr + c * sc];
				az0 += b0 * zc; ap0 += b0 * pc; az1 += b1 * zc; ap1 += b1 * pc;
			}
		}
	}
	// fold the 20 slots (lanes 3 apart) onto lanes 0..2, then spread the six block rows over lanes 0..5
	az0 += __shfl_down(az0, 30); ap0 += __shfl_down(ap0, 30); az1 += __shfl_down(az1, 30); ap1 += __shfl_down(ap1, 30);
	az0 += __shfl_down(az0, 15); ap0 += __shfl_down(ap0, 15); az1 += __shfl_down(az1, 15); ap1 += __shfl_down(ap1, 15);
	Scalar t0 = az0, u0 = ap0, t1 = az1, u1 = ap1;
#pragma unroll
	for (int d = 3; d <= 12; d += 3) { t0 += __shfl_down(az0, d); u0 += __shfl_down(ap0, d); t1 += __shfl_down(az1, d); u1 += __shfl_down(ap1, d); }
	const Scalar tzA = __shfl(t0, lane >> 1), tzB = __shfl(t1, lane >> 1), tpA = __shfl(u0, lane >> 1), tpB = __shfl(u1, lane >> 1);
	const Scalar tz = (lane & 1) ? tzB : tzA, tp = (lane & 1) ? tpB : tpA;
	__shared__ Scalar qrow[ROWS][6];
	__shared__ Scalar part[ROWS];
	Scalar q = 0, dot = 0;
	if (rowOn && lane < 6)
	{
		const Scalar pi = zi + beta * pi_old;
		q = tz + beta * tp;
		pnew[6 * (size_t)row + lane] = pi;
		sys.ap[6 * (size_t)row + lane] = q;
		dot = pi * q;
	}
	if (lane < 6) qrow[lr][lane] = q;
	dot = wave_sum(dot);
	if (lane == 0) part[lr] = dot;
	__syncthreads();
	if (threadIdx.x < 6 * sys.cl)   // (weighted) row sums of q over this workgroup's rows: the two-level kernel builds P^T q from these
	{
		const int a = threadIdx.x / 6, c = threadIdx.x - 6 * a;
		Scalar s2 = 0;
#pragma unroll
		for (int w = 0; w < ROWS; w++)
			s2 += (a == 0 ? Scalar(1) : agg_weight(blockIdx.x * ROWS + w, sys.agg, g.Pf)) * qrow[w][c];
		if (sys.qpart && sys.agg > 0)
		{
			const int per = sys.agg / ROWS, J = blockIdx.x / per, m = blockIdx.x - J * per;
			sys.qpart[(size_t)m * (6 * sys.cl * sys.nc) + 6 * sys.cl * J + threadIdx.x] = s2;
		}
	}
	if (threadIdx.x == 64)
	{
		Scalar s2 = 0;
#pragma unroll
		for (int w = 0; w < ROWS; w++) s2 += part[w];
		pq_slot(sys, k)[blockIdx.x] = s2;
	}
}

// B(k): alpha = rz[k]/pq[k]; x += alpha p; r -= alpha q; z = Minv r; rz[k+1] += r.z
__global__ __launch_bounds__(256) void pcg_update_kernel(DeviceGraph g, DeviceStructure st, DeviceSystem sys, int k, int maxIter, Scalar tol2)
{
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	k += *sys.kbase;
	Scalar rzk;
	if (!pcg_active(sys, k, maxIter, tol2, lane, rzk)) return;
	const Scalar pqk = wave_sum(load_parts(pq_slot(sys, k), sys.npq, lane));
	if (!(pqk > 0))
	{
		if (blockIdx.x == 0 && threadIdx.x == 0) *sys.fail = 2;   // not positive definite along p
		return;
	}
	const Scalar alpha = rzk / pqk;
	const Scalar* p = (k & 1) ? sys.p0 : sys.p1;   // what A(k) wrote
	const int pose = (blockIdx.x * 4 + wv) * 10 + lane / 6;
	const int rr = lane % 6;
	Scalar rnew = 0;
	const bool on = lane < 60 && pose < g.Pf;
	if (on)
	{
		const size_t idx = 6 * (size_t)pose + rr;
		sys.xp[idx] += alpha * p[idx];
		rnew = sys.r[idx] - alpha * sys.ap[idx];
		sys.r[idx] = rnew;
	}
	Scalar z = 0;
	const int base = lane - rr;
#pragma unroll
	for (int c = 0; c < 6; c++)
	{
		const Scalar rc = __shfl(rnew, base + c);
		if (on) z += sys.minv[36 * (size_t)pose + c * 6 + rr] * rc;
	}
	Scalar dot = 0;
	if (on)
	{
		sys.z[6 * (size_t)pose + rr] = z;
		dot = rnew * z;
	}
	dot = wave_sum(dot);
	__shared__ Scalar part[4];
	if (lane == 0) part[wv] = dot;
	__syncthreads();
	if (threadIdx.x == 0) rz_slot_w(sys, k + 1)[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
	if (blockIdx.x == 0 && threadIdx.x == 0) *sys.iters = k + 1;
}

// Fused B(k) of the two-level PCG: [x += alpha p; r -= alpha q;]  rc = P^T r;  z = Minv r + P (Ac^-1 rc);
// rz[kOut] = r.z.  One 512-thread workgroup per aggregate; only the owner of an aggregate stores x, r, z.
// Every workgroup needs the WHOLE restricted residual rc = P^T r_k - alpha P^T q_k (6*nc values). Neither term is
// rebuilt from the full vectors: P^T r_k was stored by the owners one iteration earlier (sys.rc, ping-pong), and
// P^T q_k is summed from the per-workgroup row sums the SpMV kernel leaves in sys.qpart.  Together with the local-k
// slot addressing every global load of the kernel is issued in its first instructions (one memory round trip).
// doUpdate = 0 (once per solve: z_0 = M^-1 r_0) still restricts r directly.
constexpr int PCG2_T = 512;

// CL = coarse functions per aggregate and pose component: 1 = constant, 2 = constant + linear in the pose index.  Coarse
// unknown (aggregate J, function a, component c) has index (6 CL) J + 6 a + c.
// AC2: further column pairs per lane and row, fetched in a second batch once the restricted sums have freed their registers
// (coarse dimensions beyond 128 AC = 1536: large graphs with small aggregates; a column-by-column tail would pay one memory
// round trip per 64 columns)
// W: numbers per 16-byte load of the coarse inverse = 2 when it is stored in the library's Scalar, 4 when the fp64 library keeps it
// in fp32 (option "precond_fp32": the preconditioner only has to be a fixed SPD operator close to the inverse, so its storage
// precision changes the iteration count by nothing measurable and the solution not at all, while its bytes and its load
// instructions -- what bounds this kernel on the one CU a workgroup runs on -- halve).  With W = 4 the rows are padded to a
// multiple of 4 numbers (zeros), and so are the two coarse vectors in LDS.
template <typename T, int W> struct InvVec;
template <typename T> struct InvVec<T, 2> { typedef T type __attribute__((ext_vector_type(2))); };
template <typename T> struct InvVec<T, 4> { typedef T type __attribute__((ext_vector_type(4))); };

// PRE = true: the preconditioner alone -- third launch of the upper-triangle iteration (large graphs, below): the residual and P^T r
// of iteration k + 1 are already in place (pcg_rows_kernel), so this instantiation only applies M^-1 to them and forms r.z
// (doUpdate = 1 at run time; alpha = 0, no partial sums of P^T q, nothing of x / r / P^T r is stored).
template <int CL, int AC, int AC2, int W, bool PRE>
__device__ __forceinline__ void pcg2_fused_body(const DeviceGraph& g, const DeviceSystem& sys, int k, int kOut, int maxIter, Scalar tol2, int doUpdate)
{
	constexpr int CD = 6 * CL;         // coarse unknowns per aggregate
	constexpr int QV = 16;             // SpMV-workgroup partials prefetched per coarse unknown
	typedef typename std::conditional<W == 4, float, Scalar>::type PT;       // storage type of the coarse inverse
	typedef typename InvVec<PT, W>::type AV;
	extern __shared__ __align__(16) unsigned char pcg2_lds[];
	const int Nc = CD * sys.nc;
	const int NcP = (Nc + 3) & ~3;     // padded length of the coarse vectors in LDS (and of the rows of a W = 4 inverse)
	const int ld = W == 4 ? NcP : Nc;
	const PT* acinv = W == 4 ? reinterpret_cast<const PT*>(sys.acinv32) : reinterpret_cast<const PT*>(sys.acinv);
	Scalar* sR = reinterpret_cast<Scalar*>(pcg2_lds);
	Scalar* sQ = sR + NcP;
	Scalar* part = sQ + NcP;          // [8 waves][CD], reused for the 8 x CD partial sums of P^T r_{k+1}
	Scalar* yc = part + 8 * CD;
	Scalar* wsum = yc + CD;           // [4][8]: per-wave partials of r_k.z_k, r_0.z_0, p.Ap and of the new r.z
	Scalar* rown = wsum + 32;
	Scalar* qown = rown + 6 * sys.agg;
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const Scalar* p = (k & 1) ? sys.p0 : sys.p1;
	// the residual is double-buffered: other workgroups still read r_k of this aggregate while its owner stores r_{k+1}
	const Scalar* rin = ((k & 1) != 0) != PRE ? sys.r2 : sys.r;              // (PRE: what pcg_rows_kernel of this iteration stored)
	Scalar* rout = (k & 1) ? sys.r : sys.r2;
	const Scalar* rcin = sys.rc + (((k & 1) != 0) != PRE ? Nc : 0);
	Scalar* rcout = sys.rc + ((doUpdate != 0) == ((k & 1) != 0) ? 0 : Nc);    // doUpdate = 0 stores P^T r_0 where k = 0 reads it
	const int I = blockIdx.x;
	TRACE_DECL
	TRACE_MARK();
	const int kb_v = vector_load_flag(sys.kbase);          // k, kOut are chunk-local: slots depend on k & 3 only, the tests use k + kb
	const int failed_v = vector_load_flag(sys.fail) | (doUpdate ? vector_load_flag(sys.done) : 0);
	// ---- every global load of the common case is issued here, before the first use of any of them ----------------
	const int t = threadIdx.x;
	const int own0 = 6 * I * sys.agg;
	const int ownN = min(6 * g.Pf, own0 + 6 * sys.agg) - own0;
	const int per = sys.agg / sys.spmv_rows;             // SpMV workgroups per aggregate
	Scalar e_k = 0, e_0 = 0, e_q0 = 0, e_q1 = 0;        // reduction partials
	Scalar pre_r = 0, pre_q = 0, pre_p = 0, pre_x = 0, pre_m[6] = { 0, 0, 0, 0, 0, 0 };   // own rows
	Scalar2 sr = { 0, 0 }, qv[QV];                       // restricted sums: thread t takes the coarse unknowns 2 t, 2 t + 1 (Nc is even)
	// coarse inverse: rows CD I .. CD I + CD - 1 (= columns: symmetric, contiguous).  Wave w applies rows w, w + 8 (< CD) to
	// the whole coarse vector -- lane l takes the columns l, l + 64, ... -- so that a row costs ONE wave reduction in one
	// wave (a thread-per-column layout needs CD reductions in every wave plus a cross-wave stage).
	constexpr int AR = (CD + 7) / 8;   // rows per wave
	// AC = prefetched column PAIRS per lane and row (covers a coarse dimension of 128 AC; the rest is read later): 6 for
	// coarse dimensions up to 768 (KITTI-00: 672), 12 beyond -- every prefetch slot past the row's end is still a load
	// instruction on the workgroup's one CU, which is what bounds this kernel
	AV ainv[AR][AC];
#pragma unroll
	for (int m = 0; m < QV; m++) qv[m] = Scalar2{ 0, 0 };
	if (doUpdate)
	{
		if (t < sys.nrz) e_k = rz_slot(sys, k)[t];
		if (t < sys.nrz0) e_0 = sys.rz[t];
		if (!PRE && t < sys.npq) e_q0 = pq_slot(sys, k)[t];
		if (!PRE && t + PCG2_T < sys.npq) e_q1 = pq_slot(sys, k)[t + PCG2_T];
	}
	if (t < ownN)
	{
		const size_t gi = (size_t)own0 + t;
		pre_r = rin[gi];
		if (doUpdate && !PRE) { pre_q = sys.ap[gi]; pre_p = p[gi]; pre_x = sys.xp[gi]; }
		const size_t pose = gi / 6; const int comp = (int)(gi - 6 * pose);
#pragma unroll
		for (int c = 0; c < 6; c++) pre_m[c] = sys.minv[36 * pose + c * 6 + comp];
	}
	if (doUpdate && 2 * t < Nc)
	{
		sr = *reinterpret_cast<const Scalar2*>(rcin + 2 * t);
#pragma unroll
		for (int m = 0; m < QV; m++)      // (m < per is uniform over the grid; sets of workgroups that do not exist stay zero)
			if (!PRE && m < per) qv[m] = *reinterpret_cast<const Scalar2*>(sys.qpart + (size_t)m * Nc + 2 * t);
	}
#pragma unroll
	for (int a = 0; a < AR; a++)
	{
#pragma unroll
		for (int m = 0; m < AC; m++) ainv[a][m] = AV(0);
		if (wv + 8 * a < CD)                               // wave-uniform
		{
			const PT* Arow = acinv + (size_t)(CD * I + wv + 8 * a) * ld;
#pragma unroll
			for (int m = 0; m < AC; m++) ainv[a][m] = *reinterpret_cast<const AV*>(Arow + min(W * lane + 64 * W * m, ld - W));   // ld is a multiple of W
		}
	}
	TRACE_MARK();      // (trace build only: waits for every load issued above)
	// ---- rare remainders (more partials / coarse unknowns / own rows than threads) and the arithmetic ------------
	Scalar a_k = e_k, a_0 = e_0, a_q = e_q0 + e_q1;
	if (doUpdate)
	{
		for (int u = t + PCG2_T; u < sys.nrz; u += PCG2_T) a_k += rz_slot(sys, k)[u];
		for (int u = t + PCG2_T; u < sys.nrz0; u += PCG2_T) a_0 += sys.rz[u];
		if (!PRE) for (int u = t + 2 * PCG2_T; u < sys.npq; u += PCG2_T) a_q += pq_slot(sys, k)[u];
	}
	if (t < ownN) { rown[t] = pre_r; qown[t] = pre_q; }
	for (int w = t + PCG2_T; w < ownN; w += PCG2_T)
	{
		rown[w] = rin[own0 + w];
		qown[w] = doUpdate && !PRE ? sys.ap[own0 + w] : Scalar(0);
	}
	// restricted sums P^T r_k and P^T q_k (fixed summation order => reproducible)
	if (doUpdate)
	{
		for (int pj = t; 2 * pj < Nc; pj += PCG2_T)
		{
			Scalar2 s1, s2 = { 0, 0 };
			if (pj == t)
			{
				s1 = sr;
#pragma unroll
				for (int m = 0; m < QV; m++) s2 += qv[m];
			}
			else s1 = *reinterpret_cast<const Scalar2*>(rcin + 2 * pj);
			for (int m0 = pj == t ? QV : 0; !PRE && m0 < per; m0 += QV)      // further unknowns of this thread (large graphs): QV loads per trip
			{
				Scalar2 qx[QV];
				const Scalar* src = sys.qpart + (size_t)m0 * Nc + 2 * pj;
#pragma unroll
				for (int m = 0; m < QV; m++) qx[m] = m0 + m < per ? *reinterpret_cast<const Scalar2*>(src + (size_t)m * Nc) : Scalar2{ 0, 0 };      // uniform condition
#pragma unroll
				for (int m = 0; m < QV; m++) s2 += qx[m];
			}
			*reinterpret_cast<Scalar2*>(sR + 2 * pj) = s1;
			*reinterpret_cast<Scalar2*>(sQ + 2 * pj) = s2;
		}
		if (t < NcP - Nc) { sR[Nc + t] = 0; sQ[Nc + t] = 0; }      // padding of the coarse vectors (a 4-wide last load of a row)
	}
	else
	{
		for (int jc = t; jc < Nc; jc += PCG2_T)       // once per solve: P^T r_0 from the residual itself
		{
			const int Jj = jc / CD, rem = jc - CD * Jj;
			const int a = rem / 6, c = rem - 6 * a;
			const int i0 = Jj * sys.agg, i1 = min(g.Pf, i0 + sys.agg);
			Scalar s1 = 0;
			for (int i = i0; i < i1; i += 8)
			{
				Scalar rv[8];
#pragma unroll
				for (int m = 0; m < 8; m++) rv[m] = i + m < i1 ? rin[6 * (size_t)(i + m) + c] : Scalar(0);
#pragma unroll
				for (int m = 0; m < 8; m++) s1 += (a == 0 ? Scalar(1) : agg_weight_local(Jj, i + m - i0, sys, g.Pf)) * rv[m];
			}
			sR[jc] = s1; sQ[jc] = 0;
		}
		if (t < NcP - Nc) { sR[Nc + t] = 0; sQ[Nc + t] = 0; }
	}
	AV ainv2[AR][AC2 > 0 ? AC2 : 1];
	if (AC2 > 0 && Nc > 64 * W * AC)                    // (uniform over the grid)
	{
#pragma unroll
		for (int a = 0; a < AR; a++)
		{
			const PT* Arow = acinv + (size_t)(CD * I + min(wv + 8 * a, CD - 1)) * ld;
#pragma unroll
			for (int m = 0; m < AC2; m++) ainv2[a][m] = *reinterpret_cast<const AV*>(Arow + min(W * lane + 64 * W * (AC + m), ld - W));
		}
	}
	TRACE_MARK();
	a_k = wave_sum(a_k); a_0 = wave_sum(a_0); a_q = wave_sum(a_q);
	if (lane == 0) { wsum[wv] = a_k; wsum[8 + wv] = a_0; wsum[16 + wv] = a_q; }
	__syncthreads();
	Scalar alpha = 0;
	const int kabs = k + __builtin_amdgcn_readfirstlane(kb_v);
	const int failed = __builtin_amdgcn_readfirstlane(failed_v);
	if (doUpdate)
	{
		Scalar rzk = 0, rz0 = 0, pqk = 0;
#pragma unroll
		for (int w = 0; w < PCG2_T / 64; w++) { rzk += wsum[w]; rz0 += wsum[8 + w]; pqk += wsum[16 + w]; }
		if (!(kabs < maxIter && failed == 0 && rzk > tol2 * rz0 && rzk == rzk))
		{
			if (blockIdx.x == 0 && threadIdx.x == 0) { *sys.done = 1; if (!(rzk == rzk)) *sys.fail = 3; }
			return;
		}
		if (!PRE)
		{
			if (!(pqk > 0))
			{
				if (blockIdx.x == 0 && threadIdx.x == 0) *sys.fail = 2;
				return;
			}
			alpha = rzk / pqk;
		}
	}
	TRACE_MARK();
	// ---- own rows: r_{k+1}, x_{k+1} ---------------------------------------------------------------------------
	const int ow = t;
	for (int w = ow; w < ownN; w += PCG2_T)
	{
		const Scalar r = rown[w] - alpha * qown[w];      // rown[w] / qown[w] were written by this very thread
		rown[w] = r;
		if (doUpdate && !PRE)
		{
			rout[own0 + w] = r;
			sys.xp[own0 + w] = (w == ow ? pre_x : sys.xp[own0 + w]) + alpha * (w == ow ? pre_p : p[own0 + w]);
		}
	}
	// ---- yc = Ac^-1[CD I .. CD I + CD - 1, :] (P^T r - alpha P^T q) --------------------------------------------------
#pragma unroll
	for (int a = 0; a < AR; a++)
	{
		const int row = wv + 8 * a;
		Scalar acc = 0;
#pragma unroll
		for (int m = 0; m < AC; m++)
		{
			const int j = W * lane + 64 * W * m;
			if (j < Nc)
			{
#pragma unroll
				for (int i = 0; i < W; i++) acc += (Scalar)ainv[a][m][i] * (PRE ? sR[j + i] : sR[j + i] - alpha * sQ[j + i]);
			}
		}
		if (AC2 > 0 && Nc > 64 * W * AC)
		{
#pragma unroll
			for (int m = 0; m < AC2; m++)
			{
				const int j = W * lane + 64 * W * (AC + m);
				if (j < Nc)
				{
#pragma unroll
					for (int i = 0; i < W; i++) acc += (Scalar)ainv2[a][m][i] * (PRE ? sR[j + i] : sR[j + i] - alpha * sQ[j + i]);
				}
			}
		}
		if (row < CD)
			for (int j = lane + 64 * W * (AC + AC2); j < Nc; j += 64) acc += (Scalar)acinv[(size_t)(CD * I + row) * ld + j] * (PRE ? sR[j] : sR[j] - alpha * sQ[j]);
		acc = wave_sum(acc);
		if (lane == 0 && row < CD) yc[row] = acc;
	}
	__syncthreads();
	TRACE_MARK();
	// ---- z = Minv r + P yc for the poses of this aggregate; r.z ---------------------------------------------------
	Scalar dot = 0;
	for (int w = ow; w < ownN; w += PCG2_T)
	{
		const int il = w / 6, comp = w - 6 * il;
		Scalar z = yc[comp];
		if (CL == 2) z += agg_weight_local(I, il, sys, g.Pf) * yc[6 + comp];
#pragma unroll
		for (int c = 0; c < 6; c++)
			z += (w == ow ? pre_m[c] : sys.minv[36 * ((size_t)I * sys.agg + il) + c * 6 + comp]) * rown[6 * il + c];
		sys.z[(size_t)own0 + w] = z;
		dot += rown[w] * z;
	}
	dot = wave_sum(dot);
	if (lane == 0) wsum[24 + wv] = dot;
	// P^T r_{k+1} of the own aggregate for the next iteration, from the updated rows themselves: 8 interleaved partial
	// sums per coarse unknown here, folded after the barrier (a single thread per unknown would chain `agg` LDS reads)
	if (!PRE && t < 8 * CD)
	{
		const int u = t % CD, h = t / CD, a = u / 6, c = u - 6 * a;
		Scalar s3 = 0;
		for (int il = h; 6 * il + c < ownN; il += 8)
			s3 += (a == 0 ? Scalar(1) : agg_weight_local(I, il, sys, g.Pf)) * rown[6 * il + c];
		part[t] = s3;
	}
	__syncthreads();
	if (!PRE && t >= 64 && t < 64 + CD)
	{
		const int u = t - 64;
		Scalar s3 = 0;
#pragma unroll
		for (int h = 0; h < 8; h++) s3 += part[CD * h + u];
		rcout[CD * I + u] = s3;
	}
	if (threadIdx.x == 0)
	{
		Scalar s2 = 0;
#pragma unroll
		for (int w = 0; w < PCG2_T / 64; w++) s2 += wsum[24 + w];
		rz_slot_w(sys, kOut)[blockIdx.x] = s2;
		if (!doUpdate) sys.rz[blockIdx.x] = s2;          // r_0.z_0: kept in slot 0 for the stop test
		if (doUpdate && blockIdx.x == 0) *sys.iters = kabs + 1;
	}
	TRACE_MARK();
	if (doUpdate) TRACE_FLUSH(1, blockIdx.x * (PCG2_T / 64) + wv);
}

template <int CL, int AC, int AC2, int W, bool PRE = false>
__global__ __launch_bounds__(PCG2_T) void pcg2_fused_kernel(DeviceGraph g, DeviceSystem sys, int k, int kOut, int maxIter, Scalar tol2, int doUpdate)
{
	pcg2_fused_body<CL, AC, AC2, W, PRE>(g, sys, k, kOut, maxIter, tol2, doUpdate);
}

template <int CL, int AC, int AC2, int W>
__global__ __launch_bounds__(PCG2_T) void pcg2_fused_batch_kernel(const BatchEntry* __restrict__ tab, int k, int kOut, Scalar tol2, int doUpdate)
{
	const BatchEntry& e = tab[blockIdx.y];
	// (a graph whose solve is not part of the iterations -- it went to the exact solver -- has gridSpmv = 0; doUpdate = 0: the first
	// application, flagged graphs only)
	if ((int)blockIdx.x >= e.sys.nc || e.gridSpmv == 0 || (doUpdate == 0 && e.t.fusedOn == 0)) return;
	pcg2_fused_body<CL, AC, AC2, W, false>(e.g, e.sys, k, kOut, e.maxIter, tol2, doUpdate);
}

template <bool PRE>
static void* pcg2_kernel_sel(const DeviceSystem& sys)
{
	const int Nc = 6 * sys.cl * sys.nc;
	if (sys.acinv32 && sizeof(Scalar) == 8)
	{
		// fp32 storage of the coarse inverse: a 16-byte load carries 4 columns, 3 / 6 / 6 + 3 loads per lane and row cover 768 / 1536 / 2304
		if (sys.cl == 2) return Nc <= 768 ? (void*)pcg2_fused_kernel<2, 3, 0, 4, PRE> : Nc <= 1536 ? (void*)pcg2_fused_kernel<2, 6, 0, 4, PRE> : (void*)pcg2_fused_kernel<2, 6, 3, 4, PRE>;
		return Nc <= 768 ? (void*)pcg2_fused_kernel<1, 3, 0, 4, PRE> : Nc <= 1536 ? (void*)pcg2_fused_kernel<1, 6, 0, 4, PRE> : (void*)pcg2_fused_kernel<1, 6, 3, 4, PRE>;
	}
	const bool small = Nc <= 768;
	if (sys.cl == 2) return small ? (void*)pcg2_fused_kernel<2, 6, 0, 2, PRE> : (void*)pcg2_fused_kernel<2, 12, 6, 2, PRE>;
	return small ? (void*)pcg2_fused_kernel<1, 6, 0, 2, PRE> : (void*)pcg2_fused_kernel<1, 12, 6, 2, PRE>;
}
static void* pcg2_kernel_for(const DeviceSystem& sys, bool precondOnly = false) { return precondOnly ? pcg2_kernel_sel<true>(sys) : pcg2_kernel_sel<false>(sys); }

static size_t pcg2_lds_bytes(const DeviceSystem& sys)
{
	const size_t cd = 6 * (size_t)sys.cl;
	const size_t ncp = (cd * sys.nc + 3) & ~(size_t)3;
	return sizeof(Scalar) * (2 * ncp + 8 * cd + cd + 32 + 12 * (size_t)sys.agg);
}

void launch_pcg2_fused(const DeviceGraph& g, const DeviceSystem& sys, int k, int kOut, int maxIter, Scalar tol2, int doUpdate, hipStream_t s)
{
	const size_t lds = pcg2_lds_bytes(sys);
	hipLaunchKernelGGL((void (*)(DeviceGraph, DeviceSystem, int, int, int, Scalar, int))pcg2_kernel_for(sys), dim3(sys.nc), dim3(PCG2_T), lds, s, g, sys, k, kOut, maxIter, tol2, doUpdate);
}

// ---------------------------------------------------------------------------------------------------
// Upper-triangle iteration (large graphs: sys.upper != 0).  There the two PCG kernels above are bound by bytes, and half of
// them are redundant: the SpMV reads both triangles of the symmetric matrix from its row-ordered copy (G4M: 104 MB moved per launch for
// 48 MB of matrix), and every workgroup of the two-level kernel adds up all row-sum partials of q = A p (43 MB) -- which it needs only
// because a kernel that also updates r cannot know P^T r_{k+1} of the OTHER aggregates.  Three launches instead of two:
//   pcg_spmv_upper_kernel   one wave per block row i, straight from the upper-triangular BSR storage (diagonal block first, blocks
//                           contiguous): for every block (i, j) the product B p_j goes into q_i and the transposed product B^T p_i
//                           -- which belongs to row j -- into tq[block] (48 bytes per block); p.Ap = sum_i p_i.(D p_i + 2 sum_j B p_j)
//                           needs no completed row
//   pcg_rows_kernel         one workgroup per aggregate: q_j = ap_j + sum of the tq of row j's lower neighbours (fixed order: the
//                           adjacency list), x += alpha p, r -= alpha q, and P^T r_{k+1} of the aggregate from the updated rows
//   pcg2_fused_kernel<PRE>  z = blockdiag^-1 r + P Ac^-1 P^T r, r.z
// Every matrix byte is read once, no row-sum partials exist, the row-ordered copy (and its 9-us expand per solve) is gone.
// ---------------------------------------------------------------------------------------------------
// sums over the lanes 0..31 and over the lanes 32..63 of a wave, each handed to its own half (DPP row reductions; fixed order)
__device__ __forceinline__ Scalar half_wave_sum(Scalar v, bool upperHalf)
{
	v += dpp_shift<0x111, 0xf>(v);   // row_shr:1
	v += dpp_shift<0x112, 0xf>(v);   // row_shr:2
	v += dpp_shift<0x114, 0xf>(v);   // row_shr:4
	v += dpp_shift<0x118, 0xf>(v);   // row_shr:8
	v += dpp_shift<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
	union { Scalar s; int w[sizeof(Scalar) / 4]; } u, lo, hi;
	u.s = v;
#pragma unroll
	for (int i = 0; i < (int)(sizeof(Scalar) / 4); i++) { lo.w[i] = __builtin_amdgcn_readlane(u.w[i], 31); hi.w[i] = __builtin_amdgcn_readlane(u.w[i], 63); }
	return upperHalf ? hi.s : lo.s;
}

// NS steps of five blocks each of one block row: lane (grp, c) of the row's half-wave takes column c of block s0 + 5 s + grp
template <int NS>
__device__ __forceinline__ void upper_steps(const DeviceStructure& st, const DeviceSystem& sys, const Scalar* pold, int s0, int b0, int b1, int grp, int c,
	Scalar beta, const Scalar (&pi)[6], Scalar (&acc)[6], Scalar& dotT)
{
	int bb[NS], j[NS], lp[NS]; bool on[NS];               // (32-bit indices: 36 x nblk stays below 2^31 for every graph the handle accepts)
	Scalar2 v[NS][3]; Scalar zj[NS], pjo[NS];
#pragma unroll
	for (int s = 0; s < NS; s++)
	{
		const int b = s0 + 5 * s + grp;
		on[s] = grp < 5 && b < b1;
		bb[s] = on[s] ? b : b0;
	}
#pragma unroll
	for (int s = 0; s < NS; s++) { j[s] = st.hsc_colind[bb[s]]; lp[s] = sys.lowpos[bb[s]]; }
#pragma unroll
	for (int s = 0; s < NS; s++)
	{
		const Scalar2* B2 = reinterpret_cast<const Scalar2*>(sys.hsc + (36 * (size_t)bb[s] + 6 * c));      // column c of the block: B[0..5][c]
		v[s][0] = B2[0]; v[s][1] = B2[1]; v[s][2] = B2[2];
	}
#pragma unroll
	for (int s = 0; s < NS; s++) { zj[s] = sys.z[6 * (size_t)j[s] + c]; pjo[s] = pold[6 * (size_t)j[s] + c]; }
#pragma unroll
	for (int s = 0; s < NS; s++)
	{
		const Scalar pj = on[s] ? zj[s] + beta * pjo[s] : Scalar(0);
		const Scalar Bc[6] = { v[s][0].x, v[s][0].y, v[s][1].x, v[s][1].y, v[s][2].x, v[s][2].y };
		Scalar t = 0;
#pragma unroll
		for (int r = 0; r < 6; r++) { acc[r] += Bc[r] * pj; t += Bc[r] * pi[r]; }
		if (on[s] && bb[s] != b0)                            // (the diagonal block is its own transpose: its product is the gather above)
		{
			sys.tq[6 * (size_t)lp[s] + c] = t;               // (where row j will look for it: position of this block among ITS lower neighbours)
			dotT += pj * t;                                   // p_j . B^T p_i = p_i . B p_j: the mirrored block's share of p . A p
		}
	}
}

// TWO block rows per wave (lanes 0..31 / 32..63; five blocks of six lanes per step and row), UPPER_WAVES waves per workgroup: half the
// waves of a row-per-wave kernel -- the launch is bound by wave dispatch (~1 wave per ns) and by the dependent round trips of a wave's
// life (row pointers + reduction scalars, column indices, operands), not by the bytes per wave -- and one set of reduction scalars
// serves two rows.  Twenty blocks per row in flight (rows of the large shapes hold ~17), the rest in further trips; four waves per SIMD (fifteen blocks in
// flight at five waves per SIMD, every wave of G4M resident at once, is slower: 18.5 vs 17.2 us -- the operand phase is bound by the
// memory-side cache's ~4.5 TB/s, not by latency: profiles/r05h_*).
constexpr int UPPER_WAVES = 4;
constexpr int UPPER_ROWS = 2 * UPPER_WAVES;       // block rows per workgroup (= number of p.Ap partials: sys.npq)
__global__ __launch_bounds__(64 * UPPER_WAVES, 4) void pcg_spmv_upper_kernel(DeviceGraph g, DeviceStructure st, DeviceSystem sys, int k, int maxIter, Scalar tol2)
{
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const bool hi = lane >= 32;
	const int hl = lane & 31;
	const Scalar* pold = (k & 1) ? sys.p1 : sys.p0;
	Scalar* pnew = (k & 1) ? sys.p0 : sys.p1;
	const int lr = 2 * wv + (hi ? 1 : 0);
	const int row = blockIdx.x * UPPER_ROWS + lr;
	const int kb_v = vector_load_flag(sys.kbase);
	const int failed_v = vector_load_flag(sys.fail) | vector_load_flag(sys.done);
	const Scalar s_k = load_parts(rz_slot(sys, k), sys.nrz, lane);
	const Scalar s_0 = load_parts(sys.rz, sys.nrz0, lane);
	const Scalar s_m = load_parts(rz_slot(sys, k - 1), sys.nrz, lane);
	const bool rowOn = row < g.Pf;
	const int grp = hl / 6, c = hl - 6 * grp;              // lanes 30, 31 of a half: grp 5 -> idle
	TRACE_DECL
	TRACE_MARK();
	int b0 = 0, b1 = 0;
	if (rowOn) { b0 = st.hsc_rowptr[row]; b1 = st.hsc_rowptr[row + 1]; }
	Scalar zi[6], pio[6];
#pragma unroll
	for (int r = 0; r < 6; r++)
	{
		zi[r] = rowOn ? sys.z[6 * (size_t)row + r] : Scalar(0);
		pio[r] = rowOn ? pold[6 * (size_t)row + r] : Scalar(0);
	}
	TRACE_MARK();
	k += __builtin_amdgcn_readfirstlane(kb_v);
	const int failed = __builtin_amdgcn_readfirstlane(failed_v);
	const Scalar rzk = to_uniform(wave_sum(s_k)), rz0 = to_uniform(wave_sum(s_0)), rzm = to_uniform(wave_sum(s_m));
	if (!(k < maxIter && failed == 0 && rzk > tol2 * rz0 && rzk == rzk))   // uniform over the grid
	{
		if (blockIdx.x == 0 && threadIdx.x == 0) { *sys.done = 1; if (!(rzk == rzk)) *sys.fail = 3; }   // NaN: reported as a failed solve
		return;
	}
	const Scalar beta = k > 0 ? rzk / rzm : Scalar(0);
	Scalar pi[6];
#pragma unroll
	for (int r = 0; r < 6; r++) pi[r] = zi[r] + beta * pio[r];
	Scalar acc[6] = { 0, 0, 0, 0, 0, 0 };
	Scalar dotT = 0;
	// (trip counts of the wave = those of its longer row: the other half idles through masked steps)
	const int nb = b1 - b0;
	const int nbMax = max(__builtin_amdgcn_readlane(nb, 0), __builtin_amdgcn_readlane(nb, 32));
	int done = 0;
	for (; done + 10 < nbMax; done += 20) upper_steps<4>(st, sys, pold, b0 + done, b0, b1, grp, c, beta, pi, acc, dotT);
	if (done + 5 < nbMax) upper_steps<2>(st, sys, pold, b0 + done, b0, b1, grp, c, beta, pi, acc, dotT);
	else if (done < nbMax) upper_steps<1>(st, sys, pold, b0 + done, b0, b1, grp, c, beta, pi, acc, dotT);
	TRACE_MARK();
	Scalar qv[6];
#pragma unroll
	for (int r = 0; r < 6; r++) qv[r] = half_wave_sum(acc[r], hi);
	Scalar dot = half_wave_sum(dotT, hi);
#pragma unroll
	for (int r = 0; r < 6; r++) dot += pi[r] * qv[r];      // (uniform over the half-wave)
	if (rowOn && hl < 6)
	{
		Scalar q = qv[0], pr = pi[0];
#pragma unroll
		for (int r = 1; r < 6; r++) { q = hl == r ? qv[r] : q; pr = hl == r ? pi[r] : pr; }
		pnew[6 * (size_t)row + hl] = pr;
		sys.ap[6 * (size_t)row + hl] = q;
	}
	__shared__ Scalar part[UPPER_ROWS];
	if (hl == 0) part[lr] = rowOn ? dot : Scalar(0);
	__syncthreads();
	if (threadIdx.x == 0)
	{
		Scalar s2 = 0;
#pragma unroll
		for (int w = 0; w < UPPER_ROWS; w++) s2 += part[w];
		pq_slot(sys, k)[blockIdx.x] = s2;
	}
	TRACE_MARK();
	TRACE_FLUSH(0, blockIdx.x * UPPER_WAVES + wv);
}

// second launch of the upper-triangle iteration: one workgroup per aggregate
// (EPT = row entries per thread: an aggregate of `agg` poses has 6 agg of them -- 2 per thread up to 170 poses, 4 up to 341, 8 up to 682;
// publishStructure switches the upper-triangle iteration off beyond ROWS_MAX_AGG)
constexpr int ROWS_T = 512;
constexpr int ROWS_MAX_AGG = 8 * ROWS_T / 6;
template <int EPT>
__global__ __launch_bounds__(ROWS_T) void pcg_rows_kernel(DeviceGraph g, DeviceStructure st, DeviceSystem sys, int k, int maxIter, Scalar tol2)
{
	extern __shared__ __align__(16) unsigned char rows_lds[];
	Scalar* rown = reinterpret_cast<Scalar*>(rows_lds);      // [6 agg] updated residual of the own rows
	Scalar* part = rown + 6 * sys.agg;                       // [8][12]
	Scalar* wsum = part + 96;                                // [3][8]
	const int t = threadIdx.x, lane = t & 63, wv = t >> 6, I = blockIdx.x;
	const int CD = 6 * sys.cl, Nc = CD * sys.nc;
	const Scalar* p = (k & 1) ? sys.p0 : sys.p1;              // what the SpMV of this iteration wrote
	const Scalar* rin = (k & 1) ? sys.r2 : sys.r;
	Scalar* rout = (k & 1) ? sys.r : sys.r2;
	Scalar* rcout = sys.rc + ((k & 1) ? 0 : Nc);
	TRACE_DECL
	TRACE_MARK();
	const int kb_v = vector_load_flag(sys.kbase);
	const int failed_v = vector_load_flag(sys.fail) | vector_load_flag(sys.done);
	Scalar a_k = 0, a_0 = 0, a_q = 0;
	for (int u = t; u < sys.nrz; u += ROWS_T) a_k += rz_slot(sys, k)[u];
	for (int u = t; u < sys.nrz0; u += ROWS_T) a_0 += sys.rz[u];
	{
		// (G4M: 2500 partials of p.Ap -- four independent loads per trip)
		const Scalar* pq = pq_slot(sys, k);
		int u = t;
		Scalar q0 = 0, q1 = 0, q2 = 0, q3 = 0;
		for (; u + 3 * ROWS_T < sys.npq; u += 4 * ROWS_T) { const Scalar x0 = pq[u], x1 = pq[u + ROWS_T], x2 = pq[u + 2 * ROWS_T], x3 = pq[u + 3 * ROWS_T]; q0 += x0; q1 += x1; q2 += x2; q3 += x3; }
		for (; u < sys.npq; u += ROWS_T) q0 += pq[u];
		a_q = (q0 + q1) + (q2 + q3);
	}
	const int own0 = 6 * I * sys.agg;
	const int ownN = min(6 * g.Pf, own0 + 6 * sys.agg) - own0;
	// completed rows of q = A p: the SpMV's share + the transposed products of the lower neighbours, in adjacency order
	Scalar qrow[EPT], rrow[EPT], prow[EPT], xrow[EPT];
#pragma unroll
	for (int e = 0; e < EPT; e++) { qrow[e] = 0; rrow[e] = 0; prow[e] = 0; xrow[e] = 0; }
#pragma unroll
	for (int e = 0; e < EPT; e++)
	{
		const int w = t + e * ROWS_T;
		if (w >= ownN) break;
		const size_t gi = (size_t)own0 + w;
		const int pose = (int)(gi / 6), comp = (int)(gi - 6 * (size_t)pose);
		const int a0 = st.adj_ptr[pose];
		const int nlow = (st.adj_ptr[pose + 1] - a0) - (st.hsc_rowptr[pose + 1] - st.hsc_rowptr[pose]);
		Scalar q = sys.ap[gi];
		rrow[e] = rin[gi]; prow[e] = p[gi]; xrow[e] = sys.xp[gi];
		// (the SpMV left them in this row's own order: one contiguous range, no index list; 16 per trip -- one trip for most rows)
		const Scalar* tin = sys.tq + 6 * (size_t)(a0 - st.hsc_rowptr[pose]) + comp;
		for (int a = 0; a < nlow; a += 16)
		{
			Scalar tv[16];
#pragma unroll
			for (int m = 0; m < 16; m++) tv[m] = tin[6 * (size_t)min(a + m, nlow - 1)];
#pragma unroll
			for (int m = 0; m < 16; m++) q += a + m < nlow ? tv[m] : Scalar(0);
		}
		qrow[e] = q;
	}
	TRACE_MARK();
	a_k = wave_sum(a_k); a_0 = wave_sum(a_0); a_q = wave_sum(a_q);
	if (lane == 0) { wsum[wv] = a_k; wsum[8 + wv] = a_0; wsum[16 + wv] = a_q; }
	__syncthreads();
	TRACE_MARK();
	const int kabs = k + __builtin_amdgcn_readfirstlane(kb_v);
	const int failed = __builtin_amdgcn_readfirstlane(failed_v);
	Scalar rzk = 0, rz0 = 0, pqk = 0;
#pragma unroll
	for (int w = 0; w < ROWS_T / 64; w++) { rzk += wsum[w]; rz0 += wsum[8 + w]; pqk += wsum[16 + w]; }
	if (!(kabs < maxIter && failed == 0 && rzk > tol2 * rz0 && rzk == rzk))
	{
		if (blockIdx.x == 0 && threadIdx.x == 0) { *sys.done = 1; if (!(rzk == rzk)) *sys.fail = 3; }
		return;
	}
	if (!(pqk > 0))
	{
		if (blockIdx.x == 0 && threadIdx.x == 0) *sys.fail = 2;   // not positive definite along p
		return;
	}
	const Scalar alpha = rzk / pqk;
#pragma unroll
	for (int e = 0; e < EPT; e++)
	{
		const int w = t + e * ROWS_T;
		if (w >= ownN) break;
		const Scalar r = rrow[e] - alpha * qrow[e];
		rown[w] = r;
		rout[own0 + w] = r;
		sys.xp[own0 + w] = xrow[e] + alpha * prow[e];
	}
	__syncthreads();
	// P^T r_{k+1} of the own aggregate: 8 interleaved partial sums per coarse unknown, folded after the barrier
	if (t < 8 * CD)
	{
		const int u = t % CD, h = t / CD, a = u / 6, c = u - 6 * a;
		Scalar s3 = 0;
		for (int il = h; 6 * il + c < ownN; il += 8)
			s3 += (a == 0 ? Scalar(1) : agg_weight_local(I, il, sys, g.Pf)) * rown[6 * il + c];
		part[t] = s3;
	}
	__syncthreads();
	if (t < CD)
	{
		Scalar s3 = 0;
#pragma unroll
		for (int h = 0; h < 8; h++) s3 += part[CD * h + t];
		rcout[CD * I + t] = s3;
	}
	TRACE_MARK();
	TRACE_FLUSH(2, blockIdx.x * (ROWS_T / 64) + wv);
}

typedef void (*PcgRowsKernel)(DeviceGraph, DeviceStructure, DeviceSystem, int, int, Scalar);
static PcgRowsKernel pcg_rows_kernel_for(const DeviceSystem& sys)
{
	const int entries = 6 * sys.agg;
	return entries <= 2 * ROWS_T ? pcg_rows_kernel<2> : entries <= 4 * ROWS_T ? pcg_rows_kernel<4> : pcg_rows_kernel<8>;
}
int pcg_rows_max_aggregate() { return ROWS_MAX_AGG; }

// lowpos[b] for every off-diagonal block b = (i, j): its position among the lower neighbours of row j, counted over all rows (the
// SpMV parks B^T p_i there, pcg_rows_kernel reads the range of row j)
__global__ __launch_bounds__(256) void build_lowpos_kernel(DeviceStructure st, int Pf, int* __restrict__ lowpos)
{
	const int row = blockIdx.x * 256 + threadIdx.x;
	if (row >= Pf) return;
	const int a0 = st.adj_ptr[row];
	const int nlow = (st.adj_ptr[row + 1] - a0) - (st.hsc_rowptr[row + 1] - st.hsc_rowptr[row]);
	const int base = a0 - st.hsc_rowptr[row];
	for (int a = 0; a < nlow; a++) lowpos[st.adj_blk[a0 + a] & 0x7fffffff] = base + a;
}

void launch_build_lowpos(const DeviceGraph& g, const DeviceStructure& st, int* lowpos, hipStream_t s)
{
	if (g.Pf > 0) hipLaunchKernelGGL(build_lowpos_kernel, dim3((g.Pf + 255) / 256), dim3(256), 0, s, st, g.Pf, lowpos);
}

static void* spmv_upper_kernel_for(const DeviceSystem&) { return (void*)pcg_spmv_upper_kernel; }
int spmv_upper_grid(int Pf) { return (Pf + UPPER_ROWS - 1) / UPPER_ROWS; }
static size_t rows_lds_bytes(const DeviceSystem& sys) { return sizeof(Scalar) * (6 * (size_t)sys.agg + 96 + 24); }

void launch_pcg_upper_iteration(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, int k, int maxIter, Scalar tol2, hipStream_t s, int which)
{
	typedef void (*K3)(DeviceGraph, DeviceStructure, DeviceSystem, int, int, Scalar);
	if (which & 1) hipLaunchKernelGGL((K3)spmv_upper_kernel_for(sys), dim3(spmv_upper_grid(g.Pf)), dim3(64 * UPPER_WAVES), 0, s, g, st, sys, k, maxIter, tol2);
	if (which & 2) hipLaunchKernelGGL(pcg_rows_kernel_for(sys), dim3(sys.nc), dim3(ROWS_T), rows_lds_bytes(sys), s, g, st, sys, k, maxIter, tol2);
	if (which & 4)
		hipLaunchKernelGGL((void (*)(DeviceGraph, DeviceSystem, int, int, int, Scalar, int))pcg2_kernel_for(sys, true), dim3(sys.nc), dim3(PCG2_T), pcg2_lds_bytes(sys), s,
			g, sys, k, k + 1, maxIter, tol2, 1);
}

static bool spmv_wants_occupancy(const DeviceGraph& g) { return 2 * (long long)g.Pf > 3 * 1024; }   // two waves per row vs 1024 SIMDs x occupancy 3
int spmv_rows_for(int Pf)
{
	return 2 * (long long)Pf > 3 * 1024 ? 4 : 2;
}                            // (the 4-row workgroup needs the 128-VGPR instantiation)
// large graphs (4 rows per workgroup): one wave per block row; small ones: two waves per row, 2 rows per workgroup
// (measured: S2M 17.0 -> 15.6 us, G4M 28.5 -> 25.0 us with the row-per-wave kernel; at KITTI-00 size 7.1 vs 5.9 us)
static bool spmv_row_per_wave(const DeviceSystem& sys) { return sys.spmv_rows >= 4; }
static void* spmv_kernel_for(const DeviceGraph& g, const DeviceSystem& sys)
{
	if (sys.spmv_rows == 4) return (void*)pcg_spmv_row_kernel<4>;
	return spmv_wants_occupancy(g) ? (void*)pcg_spmv_kernel<2, 4> : (void*)pcg_spmv_kernel<2, 1>;
}
static dim3 spmv_block_for(const DeviceSystem& sys) { return dim3((spmv_row_per_wave(sys) ? 64 : 128) * sys.spmv_rows); }

void launch_pcg_spmv(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, int k, int maxIter, Scalar tol2, hipStream_t s)
{
	const dim3 grid((g.Pf + sys.spmv_rows - 1) / sys.spmv_rows);
	hipLaunchKernelGGL((void (*)(DeviceGraph, DeviceStructure, DeviceSystem, int, int, Scalar))spmv_kernel_for(g, sys), grid, spmv_block_for(sys), 0, s, g, st, sys, k, maxIter, tol2);
}

void launch_pcg_update(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, int k, int maxIter, Scalar tol2, hipStream_t s)
{
	hipLaunchKernelGGL(pcg_update_kernel, dim3((g.Pf + 39) / 40), dim3(256), 0, s, g, st, sys, k, maxIter, tol2);
}

// last node of an iteration graph: advance the iteration offset and report the solver's flags straight into the
// device-mapped host block the host looks at after synchronising (no copy kernels on the way)
// (one wave.  tol2 >= 0: the node also runs the stop test on the residual the chunk's last iteration left -- r.z of iteration
// kbase + n lives in the ring slot of chunk-local index 0, chunk lengths being multiples of 4 -- so that a batch of exactly as many
// iterations as the solve needs is recognised as converged without a further iteration launch)
__device__ __forceinline__ void pcg_advance_body(const DeviceSystem& sys, int n, int report, Scalar tol2)
{
	if (tol2 >= 0 && n > 0)
	{
		const int lane = threadIdx.x;
		const Scalar rzk = wave_sum(load_parts(rz_slot(sys, 0), sys.nrz, lane)), rz0 = wave_sum(load_parts(sys.rz, sys.nrz0, lane));
		if (lane == 0 && *sys.done == 0 && *sys.fail == 0 && !(rzk > tol2 * rz0))
		{
			*sys.done = 1;
			if (!(rzk == rzk)) *sys.fail = 3;
		}
	}
	if (threadIdx.x != 0) return;
	*sys.kbase += n;
	if (report && sys.host_flags) publish_report(sys);      // the host spins on the ticket instead of paying a stream-synchronise round trip
}

__global__ __launch_bounds__(64) void pcg_advance_kernel(DeviceSystem sys, int n, int report, Scalar tol2) { pcg_advance_body(sys, n, report, tol2); }

__global__ __launch_bounds__(64) void pcg_advance_batch_kernel(const BatchEntry* __restrict__ tab, int n, int report, Scalar tol2)
{
	if (tab[blockIdx.x].gridSpmv == 0) return;
	pcg_advance_body(tab[blockIdx.x].sys, n, report, tol2);
}

// {chi2, landmark part of the gain-ratio denominator, pose part} of the evaluation just enqueued -> three device scalars
// (the multi-GPU driver all-reduces the first two in-stream instead of reading them back rank by rank)
__global__ void collect_eval_kernel(const Scalar* slots, Scalar* out)
{
	if (threadIdx.x == 0) { out[0] = slots[0]; out[1] = slots[NSLOT]; out[2] = slots[3 * NSLOT]; }
}

void launch_collect_eval(const DeviceSystem& sys, Scalar* out3, hipStream_t s)
{
	hipLaunchKernelGGL(collect_eval_kernel, dim3(1), dim3(64), 0, s, sys.slots, out3);
}

void launch_pcg_report(const DeviceSystem& sys, hipStream_t s)
{
	hipLaunchKernelGGL(pcg_advance_kernel, dim3(1), dim3(64), 0, s, sys, 0, 1, Scalar(-1));
}

void launch_pcg_advance(const DeviceSystem& sys, int n, hipStream_t s, Scalar tol2)
{
	hipLaunchKernelGGL(pcg_advance_kernel, dim3(1), dim3(64), 0, s, sys, n, 1, tol2);
}

// ---- batched execution: one launch chain for the PCG iterations of several graphs (cuba_hip_optimize_batch) ---------------------------
// Every graph of a batch must use the same instantiations of the two iteration kernels (they are chosen by the graph's size class:
// coarse dimension, storage of the coarse inverse, SpMV occupancy) and the two-launch iteration; batch_kernel_class() names the class.
template <int CL, int AC, int AC2, int W>
static void* pcg2_batch_fn() { return (void*)pcg2_fused_batch_kernel<CL, AC, AC2, W>; }
static void* pcg2_batch_kernel_for(const DeviceSystem& sys)
{
	const int Nc = 6 * sys.cl * sys.nc;
	if (sys.acinv32 && sizeof(Scalar) == 8)
	{
		if (sys.cl == 2) return Nc <= 768 ? pcg2_batch_fn<2, 3, 0, 4>() : Nc <= 1536 ? pcg2_batch_fn<2, 6, 0, 4>() : pcg2_batch_fn<2, 6, 3, 4>();
		return Nc <= 768 ? pcg2_batch_fn<1, 3, 0, 4>() : Nc <= 1536 ? pcg2_batch_fn<1, 6, 0, 4>() : pcg2_batch_fn<1, 6, 3, 4>();
	}
	const bool small = Nc <= 768;
	if (sys.cl == 2) return small ? pcg2_batch_fn<2, 6, 0, 2>() : pcg2_batch_fn<2, 12, 6, 2>();
	return small ? pcg2_batch_fn<1, 6, 0, 2>() : pcg2_batch_fn<1, 12, 6, 2>();
}

int batch_kernel_class(const DeviceGraph& g, const DeviceSystem& sys)
{
	if (sys.agg <= 0 || sys.upper || sys.spmv_rows != 2) return -1;          // (block-Jacobi-only, upper-triangle and row-per-wave iterations run one graph at a time)
	const int Nc = 6 * sys.cl * sys.nc;
	const int cls = (sys.acinv32 && sizeof(Scalar) == 8) ? (Nc <= 768 ? 0 : Nc <= 1536 ? 1 : 2) : (Nc <= 768 ? 3 : 4);
	return (cls * 2 + (sys.cl == 2 ? 1 : 0)) * 2 + (spmv_wants_occupancy(g) ? 1 : 0);
}

size_t batch_pcg2_lds_bytes(const DeviceSystem& sys) { return pcg2_lds_bytes(sys); }

void launch_pcg_batch_iteration(const BatchEntry* tab, int n, const DeviceGraph& g0, const DeviceSystem& sys0, int gridSpmvMax, int ncMax, size_t ldsMax, int k, Scalar tol2, hipStream_t s)
{
	// (the 128-register instantiation holds four workgroups per compute unit instead of one or two: it is the one to use as soon as the
	// batch's rows need more than one round of workgroups -- same source, same arithmetic, same bits)
	void* spmv = 2LL * gridSpmvMax * n * 2 > 3 * 1024 || spmv_wants_occupancy(g0) ? (void*)pcg_spmv_batch_kernel<2, 4> : (void*)pcg_spmv_batch_kernel<2, 1>;
	hipLaunchKernelGGL((void (*)(const BatchEntry*, int, Scalar))spmv, dim3(gridSpmvMax, n), dim3(256), 0, s, tab, k, tol2);
	hipLaunchKernelGGL((void (*)(const BatchEntry*, int, int, Scalar, int))pcg2_batch_kernel_for(sys0), dim3(ncMax, n), dim3(PCG2_T), ldsMax, s, tab, k, k + 1, tol2, 1);
}

void launch_batch_first_precond(const BatchEntry* tab, int n, const DeviceSystem& sys0, int ncMax, size_t ldsMax, Scalar tol2, hipStream_t s)
{
	hipLaunchKernelGGL((void (*)(const BatchEntry*, int, int, Scalar, int))pcg2_batch_kernel_for(sys0), dim3(ncMax, n), dim3(PCG2_T), ldsMax, s, tab, 0, 0, tol2, 0);
}

void launch_pcg_batch_advance(const BatchEntry* tab, int n, int iters, hipStream_t s, Scalar tol2)
{
	hipLaunchKernelGGL(pcg_advance_batch_kernel, dim3(n), dim3(64), 0, s, tab, iters, 1, tol2);
}

void launch_pcg_iteration(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, int k, int maxIter, Scalar tol2, hipStream_t s)
{
	launch_pcg_spmv(g, st, sys, k, maxIter, tol2, s);
	launch_pcg_update(g, st, sys, k, maxIter, tol2, s);
}

// ---------------------------------------------------------------------------------------------------
// hipGraph of `chunk` PCG iterations + the kbase advance, built node by node (no stream capture: captures are
// invalidated by unrelated work other host threads put on the legacy stream meanwhile, e.g. a second solver handle).
// ---------------------------------------------------------------------------------------------------
template <typename... Args>
static hipError_t add_kernel_node(hipGraph_t graph, hipGraphNode_t& last, void* fn, dim3 grid, dim3 block, unsigned lds, Args... args)
{
	void* ptrs[] = { (void*)&args... };
	hipKernelNodeParams p = {};
	p.func = fn; p.gridDim = grid; p.blockDim = block; p.sharedMemBytes = lds; p.kernelParams = ptrs; p.extra = nullptr;
	hipGraphNode_t node = nullptr;
	const hipError_t e = hipGraphAddKernelNode(&node, graph, last ? &last : nullptr, last ? 1 : 0, &p);
	last = node;
	return e;
}

hipError_t graph_add_pcg_chunk(hipGraph_t graph, const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, int chunk, int maxIter, Scalar tol2, int report)
{
	hipGraphNode_t last = nullptr;
	hipError_t e = hipSuccess;
	for (int k = 0; k < chunk && e == hipSuccess; k++)
	{
		if (sys.upper && sys.agg > 0)
		{
			e = add_kernel_node(graph, last, spmv_upper_kernel_for(sys), dim3(spmv_upper_grid(g.Pf)), dim3(64 * UPPER_WAVES), 0, g, st, sys, k, maxIter, tol2);
			if (e == hipSuccess) e = add_kernel_node(graph, last, (void*)pcg_rows_kernel_for(sys), dim3(sys.nc), dim3(ROWS_T), (unsigned)rows_lds_bytes(sys), g, st, sys, k, maxIter, tol2);
			if (e == hipSuccess) e = add_kernel_node(graph, last, pcg2_kernel_for(sys, true), dim3(sys.nc), dim3(PCG2_T), (unsigned)pcg2_lds_bytes(sys), g, sys, k, k + 1, maxIter, tol2, 1);
			continue;
		}
		e = add_kernel_node(graph, last, spmv_kernel_for(g, sys), dim3((g.Pf + sys.spmv_rows - 1) / sys.spmv_rows), spmv_block_for(sys), 0, g, st, sys, k, maxIter, tol2);
		if (e != hipSuccess) break;
		if (sys.agg > 0)
		{
			e = add_kernel_node(graph, last, pcg2_kernel_for(sys), dim3(sys.nc), dim3(PCG2_T),
				(unsigned)pcg2_lds_bytes(sys), g, sys, k, k + 1, maxIter, tol2, 1);
		}
		else e = add_kernel_node(graph, last, (void*)pcg_update_kernel, dim3((g.Pf + 39) / 40), dim3(256), 0, g, st, sys, k, maxIter, tol2);
	}
	if (e == hipSuccess) e = add_kernel_node(graph, last, (void*)pcg_advance_kernel, dim3(1), dim3(64), 0, sys, chunk, report, tol2);
	return e;
}


}  // namespace cubahip


#ifdef CUBA_HIP_TRACE
extern "C" int cuba_hip_debug_read_trace(unsigned long long* out)   // 3 x 8192 x 8 timestamps (100 MHz)
{
	return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cubahip::cuba_trace_buf), sizeof(unsigned long long) * 3 * 8192 * 8);
}
#endif

