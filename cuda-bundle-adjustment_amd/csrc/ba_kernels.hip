// ba_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the bundle-adjustment hot path.
//
// What the reference does with 9 edge/landmark/pair-parallel kernels and materialised per-edge blocks
// (/root/reference/src/cuda_block_solver.cu:733-1091: computeActiveErrors, constructQuadraticForm,
// computeBschure, initializeHschur, computeHschure, schurComplementPost, update*, computeScale) is done
// here with landmark-major wavefront kernels that keep every per-edge quantity in registers / LDS:
//
//   linearize_kernel      one lane per edge, a wave owns whole landmarks (edges sorted by landmark):
//                         residual, Jacobians, IRLS weight, Hll/bl reduced through LDS inside the wave,
//                         3x3 inverse, W = Hpl Hll^-1, and all Schur products W_i Hpl_j^T with the Hpl
//                         tiles of the wave staged in LDS; only pose-side targets use fp64 atomics.
//   back_substitute_kernel same lane/landmark mapping; recomputes the Jacobians instead of re-reading
//                         144 B/edge of Hpl, reduces Hpl^T xp in LDS, one xl store per landmark.
//   residual_chi2_kernel  edge-parallel robust chi2 with wave-shuffle + slot atomics.
//   pcg_*                 block-Jacobi preconditioned CG on the upper-BSR reduced system, one wave per
//                         block row, symmetric half read transposed, two kernels per iteration.
//
// Large landmarks (> 64 observations) take the big_* kernels: one 256-thread workgroup per landmark.

#include "ba_kernels.hpp"

#include <type_traits>

namespace cubahip
{

// Stage timestamps for latency studies (scripts/trace_pcg.py): only in the separate libcuba_hip_trace.so build.
#ifdef CUBA_HIP_TRACE
__device__ unsigned long long cuba_trace_buf[3][8192 * 8];
#define TRACE_DECL unsigned long long tr_[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }; int trn_ = 0; \
	{ unsigned long long t0_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0_)); tr_[7] = t0_; }
#define TRACE_MARK() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); tr_[trn_++] = wall_clock64(); } while (0)
#define TRACE_FLUSH(kid, wave) do { if ((threadIdx.x & 63) == 0 && (wave) < 8192) for (int t_ = 0; t_ < 8; t_++) cuba_trace_buf[kid][(wave) * 8 + t_] = tr_[t_]; } while (0)
#else
#define TRACE_DECL
#define TRACE_MARK() do { } while (0)
#define TRACE_FLUSH(kid, wave) do { } while (0)
#endif


// ---------------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void atomic_add(Scalar* p, Scalar v)
{
	__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void atomic_max_nonneg(unsigned long long* p, Scalar v)
{
	// IEEE-754 ordering of non-negative doubles equals the ordering of their bit patterns
	if (v > 0) atomicMax(p, (unsigned long long)__double_as_longlong((double)v));   // always compared as doubles
}

// Wave-wide reductions on the DPP data path (a few cycles per step) instead of __shfl_xor (ds_bpermute: an LDS-crossbar
// round trip per step and per 32-bit half).  row_shr 1/2/4/8 leave each 16-lane row's total in its last lane,
// row_bcast:15 / row_bcast:31 carry the totals across rows into lane 63, which is then broadcast through a scalar
// register.  Lanes without a source lane receive the identity 0.  The summation order is fixed.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ Scalar dpp_shift(Scalar v)
{
	union { Scalar s; int w[sizeof(Scalar) / 4]; } a, b;
	a.s = v;
#pragma unroll
	for (int i = 0; i < (int)(sizeof(Scalar) / 4); i++) b.w[i] = __builtin_amdgcn_update_dpp(0, a.w[i], CTRL, ROW_MASK, 0xf, false);
	return b.s;
}

__device__ __forceinline__ Scalar broadcast_lane63(Scalar v)
{
	union { Scalar s; int w[sizeof(Scalar) / 4]; } u;
	u.s = v;
#pragma unroll
	for (int i = 0; i < (int)(sizeof(Scalar) / 4); i++) u.w[i] = __builtin_amdgcn_readlane(u.w[i], 63);
	return u.s;
}

__device__ __forceinline__ Scalar wave_sum(Scalar v)
{
	v += dpp_shift<0x111, 0xf>(v);   // row_shr:1
	v += dpp_shift<0x112, 0xf>(v);   // row_shr:2
	v += dpp_shift<0x114, 0xf>(v);   // row_shr:4
	v += dpp_shift<0x118, 0xf>(v);   // row_shr:8
	v += dpp_shift<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
	v += dpp_shift<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
	return broadcast_lane63(v);
}

// Load of a uniform flag through the VECTOR memory path. A plain `*p` of a uniform address becomes an s_load, and the
// next kernel-argument use then waits for lgkmcnt(0), i.e. for this load's full memory round trip, before the first
// vector load of the kernel can even be issued.
__device__ __forceinline__ int vector_load_flag(const int* p)
{
	int zero;
	asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
	return p[zero];
}

// 1/x for a normal positive x: hardware reciprocal estimate + two Newton steps (full double precision, none of the
// scaling / fix-up steps of the IEEE division sequence)
__device__ __forceinline__ Scalar fast_rcp(Scalar x)
{
	Scalar r = __builtin_amdgcn_rcp(x);
	r = r * (Scalar(2) - x * r);
	r = r * (Scalar(2) - x * r);
	return r;
}

// weight of pose `pose` in the linear coarse function of its aggregate: -1 .. 1 across the aggregate, 0 in the middle
// (a trailing aggregate of a single pose has no linear function: weight 0 there, and coarse_assemble_kernel puts an
// identity block on its diagonal so that the coarse matrix stays regular)
__device__ __forceinline__ Scalar agg_weight(int pose, int agg, int Pf)
{
	if (pose == Pf - 1 && Pf % agg == 1) return Scalar(0);
	return Scalar(2 * (pose % agg) + 1 - agg) / Scalar(agg);
}
// the same for a pose given by its position inside aggregate I (no integer division / modulo)
__device__ __forceinline__ Scalar agg_weight_local(int I, int il, const DeviceSystem& sys, int Pf)
{
	if (I * sys.agg + il == Pf - 1 && il == 0) return Scalar(0);
	return Scalar(2 * il + 1 - sys.agg) * sys.inv_agg;
}

// wave-uniform value -> scalar registers (frees the vector registers a long-lived uniform would occupy)
__device__ __forceinline__ Scalar to_uniform(Scalar v)
{
	union { Scalar s; int w[sizeof(Scalar) / 4]; } u;
	u.s = v;
#pragma unroll
	for (int i = 0; i < (int)(sizeof(Scalar) / 4); i++) u.w[i] = __builtin_amdgcn_readfirstlane(u.w[i]);
	return u.s;
}

__device__ __forceinline__ Scalar wave_max(Scalar v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
	return v;
}

// lanes of one wave exchange data through LDS: keep the compiler from moving LDS accesses across
__device__ __forceinline__ void wave_lds_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ Scalar sum_slots(const Scalar* s, int lane)
{
	Scalar v = lane < NSLOT ? s[lane] : Scalar(0);
	return wave_sum(v);
}

// lane-strided partial sum of n per-workgroup partials (finish with wave_sum)
__device__ __forceinline__ Scalar load_parts(const Scalar* p, int n, int lane)
{
	// four independent loads per trip (a runtime-trip loop is not unrolled by the compiler and would pay one memory
	// round trip per element); the summation order is fixed, so the result is reproducible
	Scalar v0 = 0, v1 = 0, v2 = 0, v3 = 0;
	int i = lane;
	for (; i + 192 < n; i += 256)
	{
		const Scalar a = p[i], b = p[i + 64], c = p[i + 128], d = p[i + 192];
		v0 += a; v1 += b; v2 += c; v3 += d;
	}
	const Scalar e0 = i < n ? p[i] : Scalar(0);
	const Scalar e1 = i + 64 < n ? p[i + 64] : Scalar(0);
	const Scalar e2 = i + 128 < n ? p[i + 128] : Scalar(0);
	return ((v0 + e0) + (v1 + e1)) + ((v2 + e2) + v3);
}

// ring slots depend on k & 3 only: graph chunks are multiples of 4, so the chunk-local k of a captured launch selects
// the same slot as the absolute iteration number and no address has to wait for the kbase load
__device__ __forceinline__ const Scalar* rz_slot(const DeviceSystem& sys, int k) { return sys.rz + (size_t)(1 + (k & 3)) * sys.rzStride; }
__device__ __forceinline__ Scalar* rz_slot_w(const DeviceSystem& sys, int k) { return sys.rz + (size_t)(1 + (k & 3)) * sys.rzStride; }
__device__ __forceinline__ int rz_count(const DeviceSystem& sys, int) { return sys.nrz; }
__device__ __forceinline__ Scalar* pq_slot(const DeviceSystem& sys, int k) { return sys.pq + (size_t)(k & 3) * sys.pqStride; }

// Deterministic second stage of every global sum (chi2, gain-ratio denominator): one workgroup adds the
// per-workgroup partials in a fixed order and writes the total to out[0] (out[1..NSLOT) = 0, so hosts that add up a
// slot group keep working).  No atomics anywhere => results are reproducible bit for bit.
// per-thread share of a partial-sum array (1024 threads, fixed strides => fixed summation order)
__device__ __forceinline__ Scalar parts_thread_sum(const Scalar* __restrict__ parts, int n)
{
	Scalar v0 = 0, v1 = 0, v2 = 0, v3 = 0;
	int i = threadIdx.x;
	for (; i + 3072 < n; i += 4096)
	{
		const Scalar a = parts[i], b = parts[i + 1024], c = parts[i + 2048], d = parts[i + 3072];
		v0 += a; v1 += b; v2 += c; v3 += d;
	}
	for (; i < n; i += 1024) v0 += parts[i];
	return (v0 + v1) + (v2 + v3);
}

// 1024 thread values -> their sum, in every lane of wave 0: a wave reduction (DPP, fixed order), 16 numbers through LDS, one more
// wave reduction -- one barrier instead of the ten of a shared-memory tree
__device__ __forceinline__ Scalar block_sum_1024(Scalar v, Scalar* sh16)
{
	v = wave_sum(v);
	if ((threadIdx.x & 63) == 0) sh16[threadIdx.x >> 6] = v;
	__syncthreads();
	Scalar t = threadIdx.x < 16 ? sh16[threadIdx.x] : Scalar(0);
	if (threadIdx.x < 64) t = wave_sum(t);
	return t;
}

__device__ __forceinline__ void store_slot_group(Scalar* out, Scalar total)
{
	if (threadIdx.x < NSLOT) out[threadIdx.x] = threadIdx.x == 0 ? total : Scalar(0);
}

__global__ __launch_bounds__(1024) void reduce_parts_kernel(const Scalar* __restrict__ parts, int n, Scalar* out)
{
	__shared__ Scalar sh[16];
	store_slot_group(out, block_sum_1024(parts_thread_sum(parts, n), sh));
}

static void launch_reduce_parts(const Scalar* parts, int n, Scalar* out, hipStream_t s)
{
	hipLaunchKernelGGL(reduce_parts_kernel, dim3(1), dim3(1024), 0, s, parts, n, out);
}

// Everything a lane knows about its edge after loading + linearising it.
struct LaneEdge
{
	int ip, il;
	bool stereo;
	Scalar wr;       // omega * rho'(omega |r|^2)
	EdgeLin lin;
};

__device__ __forceinline__ void load_pose(const DeviceGraph& g, int ip, Scalar q[4], Scalar t[3], Scalar cam[5])
{
#pragma unroll
	for (int i = 0; i < 4; i++) q[i] = g.q[4 * (size_t)ip + i];
#pragma unroll
	for (int i = 0; i < 3; i++) t[i] = g.t[3 * (size_t)ip + i];
#pragma unroll
	for (int i = 0; i < 5; i++) cam[i] = g.cam[5 * (size_t)ip + i];
}

// Load edge e and linearise it at the current estimate.
__device__ __forceinline__ void linearize_edge(const DeviceGraph& g, int e, LaneEdge& out)
{
	const int pe = g.e_pose[e];
	out.stereo = (pe & STEREO_BIT) != 0;
	out.ip = pe & ~STEREO_BIT;
	out.il = g.e_lm[e];
	Scalar q[4], t[3], cam[5], Xw[3], meas[3], Xc[3];
	load_pose(g, out.ip, q, t, cam);
#pragma unroll
	for (int i = 0; i < 3; i++) Xw[i] = g.Xw[3 * (size_t)out.il + i];
	meas[0] = g.e_mu[e]; meas[1] = g.e_mv[e]; meas[2] = g.e_mr[e];
	const Scalar w = g.e_w[e];
	const Scalar s = edge_residual(q, t, cam, Xw, meas, out.stereo, out.lin.r, Xc);
	const int kind = out.stereo ? g.rk[1].kind : g.rk[0].kind;
	const Scalar delta = out.stereo ? g.rk[1].delta : g.rk[0].delta;
	out.wr = w * robust_weight(kind, delta, w * s);
	const Rot3 R = quat_to_rot(q[0], q[1], q[2], q[3]);
	edge_jacobians(Xc, R, cam, out.stereo, out.lin);
}

// ---------------------------------------------------------------------------------------------------
// robust chi2 (and optional per-edge non-robust chi2).
// Replaces computeActiveErrorsKernel / computeChiSquaresKernel (cuda_block_solver.cu:733-786, 841-875):
// no errors/Xcs are stored -- later kernels recompute them from 40 B/edge instead of re-reading 48 B/edge.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void residual_chi2_body(const DeviceGraph& g, Scalar* parts, Scalar* per_edge, int bid, int nb)
{
	Scalar acc = 0;
	for (int e = g.e_begin + bid * 256 + threadIdx.x; e < g.e_end; e += nb * 256)
	{
		const int pe = g.e_pose[e];
		const bool stereo = (pe & STEREO_BIT) != 0;
		const int ip = pe & ~STEREO_BIT;
		const int il = g.e_lm[e];
		Scalar q[4], t[3], cam[5], Xw[3], meas[3], r[3], Xc[3];
		load_pose(g, ip, q, t, cam);
#pragma unroll
		for (int i = 0; i < 3; i++) Xw[i] = g.Xw[3 * (size_t)il + i];
		meas[0] = g.e_mu[e]; meas[1] = g.e_mv[e]; meas[2] = g.e_mr[e];
		const Scalar ee = g.e_w[e] * edge_residual(q, t, cam, Xw, meas, stereo, r, Xc);
		const int kind = stereo ? g.rk[1].kind : g.rk[0].kind;
		const Scalar delta = stereo ? g.rk[1].delta : g.rk[0].delta;
		acc += robust_rho(kind, delta, ee);
		if (per_edge) per_edge[e] = ee;
	}
	acc = wave_sum(acc);
	__shared__ Scalar part[4];
	if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x == 0) parts[bid] = part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(256) void residual_chi2_kernel(DeviceGraph g, Scalar* parts, Scalar* per_edge)
{
	residual_chi2_body(g, parts, per_edge, blockIdx.x, gridDim.x);
}

void launch_residual_chi2(const DeviceGraph& g, Scalar* parts, Scalar* slots, Scalar* per_edge, hipStream_t st)
{
	const int n = g.e_end - g.e_begin;
	const int grid = n > 0 ? min((n + 255) / 256, 2048) : 0;
	if (grid > 0) hipLaunchKernelGGL(residual_chi2_kernel, dim3(grid), dim3(256), 0, st, g, parts, per_edge);
	launch_reduce_parts(parts, grid, slots, st);
}

// ---------------------------------------------------------------------------------------------------
// Per-lane products of one linearised edge
// ---------------------------------------------------------------------------------------------------
struct EdgeBlocks
{
	Scalar hpp[21];   // upper triangle of JP^T w JP, packed column by column: (r<=c) at c*(c+1)/2 + r
	Scalar bp[6];     // JP^T w r
	Scalar hll[6];    // JL^T w JL, packing 00,01,02,11,12,22
	Scalar bl[3];     // JL^T w r
	Scalar hpl[6][3]; // JP^T w JL
};

__device__ __forceinline__ void edge_products(const LaneEdge& le, EdgeBlocks& b)
{
	const EdgeLin& L = le.lin;
	const Scalar w = le.wr;
#pragma unroll
	for (int c = 0; c < 6; c++)
	{
#pragma unroll
		for (int r = 0; r <= c; r++)
			b.hpp[c * (c + 1) / 2 + r] = w * (L.JP[0][r] * L.JP[0][c] + L.JP[1][r] * L.JP[1][c] + L.JP[2][r] * L.JP[2][c]);
		b.bp[c] = w * (L.JP[0][c] * L.r[0] + L.JP[1][c] * L.r[1] + L.JP[2][c] * L.r[2]);
#pragma unroll
		for (int k = 0; k < 3; k++)
			b.hpl[c][k] = w * (L.JP[0][c] * L.JL[0][k] + L.JP[1][c] * L.JL[1][k] + L.JP[2][c] * L.JL[2][k]);
	}
#pragma unroll
	for (int i = 0; i < 3; i++)
	{
#pragma unroll
		for (int j = i; j < 3; j++)
			b.hll[sym3_idx(i, j)] = w * (L.JL[0][i] * L.JL[0][j] + L.JL[1][i] * L.JL[1][j] + L.JL[2][i] * L.JL[2][j]);
		b.bl[i] = w * (L.JL[0][i] * L.r[0] + L.JL[1][i] * L.r[1] + L.JL[2][i] * L.r[2]);
	}
}

constexpr int PAIR_DUP_BIT = 0x40000000;   // both edges of the pair observe the same pose -> T + T^T on the diagonal block
constexpr int LDS_PER_LANE = 18;           // doubles of LDS per lane (one 6x3 Hpl tile)

// ---------------------------------------------------------------------------------------------------
// linearise + assemble (+ Schur).  MODE 0: assemble only.  MODE 1: damped Schur reduction.
// Replaces constructQuadraticFormKernel + computeBschureKernel + initializeHschurKernel +
// computeHschureKernel (cuda_block_solver.cu:788-839, 933-977) and the addLambda kernels (:906-918).
// ---------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(LIN_BLOCK) void linearize_kernel(DeviceGraph g, DeviceStructure st, DeviceSystem sys, Scalar lambda)
{
	__shared__ Scalar lds_all[(LIN_BLOCK / WAVE) * WAVE * LDS_PER_LANE];
	const int lane = threadIdx.x & 63;
	const int wv = threadIdx.x >> 6;
	const int wave = blockIdx.x * (LIN_BLOCK / WAVE) + wv;
	if (wave >= st.nWaves) return;
	Scalar* lds = lds_all + wv * WAVE * LDS_PER_LANE;

	const int lm0 = st.wave_lm[2 * wave], lm1 = st.wave_lm[2 * wave + 1];
	const int e0 = g.lm_ptr[lm0], e1 = g.lm_ptr[lm1];
	const int e = e0 + lane;
	const bool valid = e < e1;

	LaneEdge le;
	EdgeBlocks b;
	le.ip = 0; le.il = lm0; le.stereo = false; le.wr = 0;
	int seg0 = 0, seg1 = 0;
	if (valid)
	{
		linearize_edge(g, e, le);
		edge_products(le, b);
		seg0 = g.lm_ptr[le.il] - e0;
		seg1 = g.lm_ptr[le.il + 1] - e0;
	}
	const bool poseFree = valid && le.ip < g.Pf;
	const bool lmFree = valid && le.il < g.Lf;

	// ---- landmark side: reduce Hll / bl over the lanes of each landmark through LDS ---------------
	Scalar Hll[6] = { 0, 0, 0, 0, 0, 0 }, bl[3] = { 0, 0, 0 };
	if (lmFree)
	{
#pragma unroll
		for (int k = 0; k < 6; k++) lds[lane * LDS_PER_LANE + k] = b.hll[k];
#pragma unroll
		for (int k = 0; k < 3; k++) lds[lane * LDS_PER_LANE + 6 + k] = b.bl[k];
	}
	wave_lds_sync();
	if (lmFree)
	{
		for (int j = seg0; j < seg1; j++)
		{
#pragma unroll
			for (int k = 0; k < 6; k++) Hll[k] += lds[j * LDS_PER_LANE + k];
#pragma unroll
			for (int k = 0; k < 3; k++) bl[k] += lds[j * LDS_PER_LANE + 6 + k];
		}
	}
	wave_lds_sync();
	const bool head = lmFree && lane == seg0;
	Scalar* diagBlk = poseFree ? sys.hsc + 36 * (size_t)st.hsc_rowptr[le.ip] : nullptr;

	if (MODE == 0)
	{
		if (head)
		{
#pragma unroll
			for (int k = 0; k < 6; k++) sys.lm_sys[9 * (size_t)le.il + k] = Hll[k];
#pragma unroll
			for (int k = 0; k < 3; k++) sys.lm_sys[9 * (size_t)le.il + 6 + k] = bl[k];
		}
		Scalar m = head ? fmax(Hll[0], fmax(Hll[3], Hll[5])) : Scalar(0);
		m = wave_max(m);
		if (lane == 0) atomic_max_nonneg(sys.maxdiag, m);
		if (poseFree)
		{
#pragma unroll
			for (int c = 0; c < 6; c++)
			{
#pragma unroll
				for (int r = 0; r <= c; r++) atomic_add(diagBlk + c * 6 + r, b.hpp[c * (c + 1) / 2 + r]);
				atomic_add(sys.bp + 6 * (size_t)le.ip + c, b.bp[c]);
			}
		}
		return;
	}

	// ---- MODE 1: damped inverse, W = Hpl (Hll + lambda I)^-1, Schur products -------------------------
	Scalar inv[6] = { 0, 0, 0, 0, 0, 0 };
	Scalar W[6][3];
	if (lmFree)
	{
		Hll[0] += lambda; Hll[3] += lambda; Hll[5] += lambda;
		sym3_inverse(Hll, inv);
		if (head)
		{
#pragma unroll
			for (int k = 0; k < 6; k++) sys.lm_sys[9 * (size_t)le.il + k] = inv[k];
#pragma unroll
			for (int k = 0; k < 3; k++) sys.lm_sys[9 * (size_t)le.il + 6 + k] = bl[k];
		}
	}
	const bool both = poseFree && lmFree;
	if (poseFree)
	{
		Scalar ibl[3] = { 0, 0, 0 };
		if (lmFree)
		{
#pragma unroll
			for (int i = 0; i < 3; i++)
				ibl[i] = inv[sym3_idx(i, 0)] * bl[0] + inv[sym3_idx(i, 1)] * bl[1] + inv[sym3_idx(i, 2)] * bl[2];
#pragma unroll
			for (int r = 0; r < 6; r++)
#pragma unroll
				for (int k = 0; k < 3; k++)
					W[r][k] = b.hpl[r][0] * inv[sym3_idx(0, k)] + b.hpl[r][1] * inv[sym3_idx(1, k)] + b.hpl[r][2] * inv[sym3_idx(2, k)];
		}
		else
		{
#pragma unroll
			for (int r = 0; r < 6; r++) { W[r][0] = 0; W[r][1] = 0; W[r][2] = 0; }
		}
#pragma unroll
		for (int c = 0; c < 6; c++)
		{
#pragma unroll
			for (int r = 0; r <= c; r++)
			{
				const Scalar d = b.hpp[c * (c + 1) / 2 + r] - (W[r][0] * b.hpl[c][0] + W[r][1] * b.hpl[c][1] + W[r][2] * b.hpl[c][2]);
				atomic_add(diagBlk + c * 6 + r, d);
			}
			atomic_add(sys.bp + 6 * (size_t)le.ip + c, b.bp[c]);
			atomic_add(sys.bsc + 6 * (size_t)le.ip + c, b.bp[c] - (b.hpl[c][0] * ibl[0] + b.hpl[c][1] * ibl[1] + b.hpl[c][2] * ibl[2]));
		}
	}
	// stage this wave's Hpl tiles in LDS
	if (both)
	{
#pragma unroll
		for (int c = 0; c < 6; c++)
#pragma unroll
			for (int k = 0; k < 3; k++) lds[lane * LDS_PER_LANE + c * 3 + k] = b.hpl[c][k];
	}
	wave_lds_sync();
	// all products W_i Hpl_j^T, i < j inside one landmark (edges are sorted by pose, free poses first)
	const int nfree = both ? st.lm_nfree[le.il] : 0;
	const int o = lane - seg0;
	const long long pbase = both ? st.lm_pair_base[le.il] + (long long)o * (nfree - 1) - (long long)o * (o - 1) / 2 - 1 : 0;
	for (int d = 1;; d++)
	{
		const bool act = both && (o + d < nfree);
		if (!__any(act)) break;
		if (act)
		{
			const Scalar* Hj = lds + (lane + d) * LDS_PER_LANE;
			const int pb = st.pair_blk[pbase + d];
			Scalar* dst = sys.hsc + 36 * (size_t)(pb & ~PAIR_DUP_BIT);
			if (!(pb & PAIR_DUP_BIT))
			{
#pragma unroll
				for (int c = 0; c < 6; c++)
				{
					const Scalar h0 = Hj[c * 3 + 0], h1 = Hj[c * 3 + 1], h2 = Hj[c * 3 + 2];
#pragma unroll
					for (int r = 0; r < 6; r++) atomic_add(dst + c * 6 + r, -(W[r][0] * h0 + W[r][1] * h1 + W[r][2] * h2));
				}
			}
			else
			{
				// same pose observed twice by one landmark: symmetric contribution to the diagonal block
				Scalar T[6][6];
#pragma unroll
				for (int c = 0; c < 6; c++)
#pragma unroll
					for (int r = 0; r < 6; r++) T[r][c] = W[r][0] * Hj[c * 3 + 0] + W[r][1] * Hj[c * 3 + 1] + W[r][2] * Hj[c * 3 + 2];
#pragma unroll
				for (int c = 0; c < 6; c++)
#pragma unroll
					for (int r = 0; r <= c; r++) atomic_add(dst + c * 6 + r, -(T[r][c] + T[c][r]));
			}
		}
	}
}

// ---- big landmarks (> 64 observations): one workgroup per landmark, Hpl tiles in a global scratch ----
template <int MODE>
__global__ __launch_bounds__(256) void big_linearize_kernel(DeviceGraph g, DeviceStructure st, DeviceSystem sys, Scalar lambda)
{
	__shared__ Scalar red[4][9];
	__shared__ Scalar tot[9];
	const int il = st.big_lm[blockIdx.x];
	const int e0 = g.lm_ptr[il], e1 = g.lm_ptr[il + 1];
	const int n = e1 - e0;
	const bool lmFree = il < g.Lf;
	Scalar* scratch = st.big_hpl + 18 * st.big_scratch_ofs[blockIdx.x];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;

	// pass 1: landmark sums
	Scalar acc[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
	if (lmFree)
	{
		for (int i = threadIdx.x; i < n; i += 256)
		{
			LaneEdge le; EdgeBlocks b;
			linearize_edge(g, e0 + i, le);
			edge_products(le, b);
#pragma unroll
			for (int k = 0; k < 6; k++) acc[k] += b.hll[k];
#pragma unroll
			for (int k = 0; k < 3; k++) acc[6 + k] += b.bl[k];
		}
	}
#pragma unroll
	for (int k = 0; k < 9; k++) acc[k] = wave_sum(acc[k]);
	if (lane == 0)
#pragma unroll
		for (int k = 0; k < 9; k++) red[wv][k] = acc[k];
	__syncthreads();
	if (threadIdx.x < 9) tot[threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
	__syncthreads();
	Scalar Hll[6], bl[3], inv[6] = { 0, 0, 0, 0, 0, 0 };
#pragma unroll
	for (int k = 0; k < 6; k++) Hll[k] = tot[k];
#pragma unroll
	for (int k = 0; k < 3; k++) bl[k] = tot[6 + k];

	if (MODE == 0)
	{
		if (lmFree && threadIdx.x == 0)
		{
#pragma unroll
			for (int k = 0; k < 6; k++) sys.lm_sys[9 * (size_t)il + k] = Hll[k];
#pragma unroll
			for (int k = 0; k < 3; k++) sys.lm_sys[9 * (size_t)il + 6 + k] = bl[k];
			atomic_max_nonneg(sys.maxdiag, fmax(Hll[0], fmax(Hll[3], Hll[5])));
		}
	}
	else if (lmFree)
	{
		Hll[0] += lambda; Hll[3] += lambda; Hll[5] += lambda;
		sym3_inverse(Hll, inv);
		if (threadIdx.x == 0)
		{
#pragma unroll
			for (int k = 0; k < 6; k++) sys.lm_sys[9 * (size_t)il + k] = inv[k];
#pragma unroll
			for (int k = 0; k < 3; k++) sys.lm_sys[9 * (size_t)il + 6 + k] = bl[k];
		}
	}
	Scalar ibl[3] = { 0, 0, 0 };
#pragma unroll
	for (int i = 0; i < 3; i++)
		ibl[i] = inv[sym3_idx(i, 0)] * bl[0] + inv[sym3_idx(i, 1)] * bl[1] + inv[sym3_idx(i, 2)] * bl[2];

	// pass 2: pose-side accumulation, Hpl tiles to scratch
	for (int i = threadIdx.x; i < n; i += 256)
	{
		LaneEdge le; EdgeBlocks b;
		linearize_edge(g, e0 + i, le);
		if (le.ip >= g.Pf) continue;
		edge_products(le, b);
		Scalar* diagBlk = sys.hsc + 36 * (size_t)st.hsc_rowptr[le.ip];
		Scalar W[6][3];
#pragma unroll
		for (int r = 0; r < 6; r++)
#pragma unroll
			for (int k = 0; k < 3; k++)
				W[r][k] = (MODE == 1 && lmFree) ? b.hpl[r][0] * inv[sym3_idx(0, k)] + b.hpl[r][1] * inv[sym3_idx(1, k)] + b.hpl[r][2] * inv[sym3_idx(2, k)] : Scalar(0);
#pragma unroll
		for (int c = 0; c < 6; c++)
		{
#pragma unroll
			for (int r = 0; r <= c; r++)
				atomic_add(diagBlk + c * 6 + r, b.hpp[c * (c + 1) / 2 + r] - (W[r][0] * b.hpl[c][0] + W[r][1] * b.hpl[c][1] + W[r][2] * b.hpl[c][2]));
			atomic_add(sys.bp + 6 * (size_t)le.ip + c, b.bp[c]);
			if (MODE == 1)
				atomic_add(sys.bsc + 6 * (size_t)le.ip + c, b.bp[c] - (b.hpl[c][0] * ibl[0] + b.hpl[c][1] * ibl[1] + b.hpl[c][2] * ibl[2]));
		}
		if (MODE == 1 && lmFree)
#pragma unroll
			for (int c = 0; c < 6; c++)
#pragma unroll
				for (int k = 0; k < 3; k++) scratch[18 * (size_t)i + c * 3 + k] = b.hpl[c][k];
	}
	if (MODE == 0 || !lmFree) return;
	__syncthreads();   // workgroup-scope visibility of this workgroup's scratch stores
	// pass 3: pair products
	const int nfree = st.lm_nfree[il];
	const long long npairs = (long long)nfree * (nfree - 1) / 2;
	const long long base = st.lm_pair_base[il];
	for (long long pidx = threadIdx.x; pidx < npairs; pidx += 256)
	{
		// invert the row-major triangular numbering: pidx = o (nfree-1) - o (o-1)/2 + (d-1)
		int o = (int)(((2.0 * nfree - 1) - sqrt((2.0 * nfree - 1) * (2.0 * nfree - 1) - 8.0 * (double)pidx)) * 0.5);
		while ((long long)o * (nfree - 1) - (long long)o * (o - 1) / 2 > pidx) o--;
		while ((long long)(o + 1) * (nfree - 1) - (long long)(o + 1) * o / 2 <= pidx) o++;
		const int j = o + 1 + (int)(pidx - ((long long)o * (nfree - 1) - (long long)o * (o - 1) / 2));
		const Scalar* Hi = scratch + 18 * (size_t)o;
		const Scalar* Hj = scratch + 18 * (size_t)j;
		Scalar W[6][3];
#pragma unroll
		for (int r = 0; r < 6; r++)
#pragma unroll
			for (int k = 0; k < 3; k++)
				W[r][k] = Hi[r * 3 + 0] * inv[sym3_idx(0, k)] + Hi[r * 3 + 1] * inv[sym3_idx(1, k)] + Hi[r * 3 + 2] * inv[sym3_idx(2, k)];
		const int pb = st.pair_blk[base + pidx];
		Scalar* dst = sys.hsc + 36 * (size_t)(pb & ~PAIR_DUP_BIT);
		if (!(pb & PAIR_DUP_BIT))
		{
#pragma unroll
			for (int c = 0; c < 6; c++)
#pragma unroll
				for (int r = 0; r < 6; r++)
					atomic_add(dst + c * 6 + r, -(W[r][0] * Hj[c * 3 + 0] + W[r][1] * Hj[c * 3 + 1] + W[r][2] * Hj[c * 3 + 2]));
		}
		else
		{
#pragma unroll
			for (int c = 0; c < 6; c++)
#pragma unroll
				for (int r = 0; r <= c; r++)
				{
					const Scalar trc = W[r][0] * Hj[c * 3 + 0] + W[r][1] * Hj[c * 3 + 1] + W[r][2] * Hj[c * 3 + 2];
					const Scalar tcr = W[c][0] * Hj[r * 3 + 0] + W[c][1] * Hj[r * 3 + 1] + W[c][2] * Hj[r * 3 + 2];
					atomic_add(dst + c * 6 + r, -(trc + tcr));
				}
		}
	}
}

void launch_linearize(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, int mode, Scalar lambda, hipStream_t s)
{
	if (st.nWaves > 0)
	{
		const int grid = (st.nWaves + (LIN_BLOCK / WAVE) - 1) / (LIN_BLOCK / WAVE);
		if (mode == 0) hipLaunchKernelGGL(linearize_kernel<0>, dim3(grid), dim3(LIN_BLOCK), 0, s, g, st, sys, lambda);
		else hipLaunchKernelGGL(linearize_kernel<1>, dim3(grid), dim3(LIN_BLOCK), 0, s, g, st, sys, lambda);
	}
	if (st.nBig > 0)
	{
		if (mode == 0) hipLaunchKernelGGL(big_linearize_kernel<0>, dim3(st.nBig), dim3(256), 0, s, g, st, sys, lambda);
		else hipLaunchKernelGGL(big_linearize_kernel<1>, dim3(st.nBig), dim3(256), 0, s, g, st, sys, lambda);
	}
}

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
__global__ __launch_bounds__(LIN_BLOCK) void lm_pass_kernel(DeviceGraph g, DeviceStructure st, DeviceSystem sys, Scalar lambda,
	unsigned nLmGroups, const Scalar* __restrict__ backupSrc, Scalar* __restrict__ backupDst, size_t backupCount)
{
	__shared__ Scalar lds_all[(LIN_BLOCK / WAVE) * WAVE * 9];
	if (blockIdx.x >= nLmGroups)
	{
		const size_t stride = (size_t)(gridDim.x - nLmGroups) * LIN_BLOCK;
		for (size_t i = (size_t)(blockIdx.x - nLmGroups) * LIN_BLOCK + threadIdx.x; i < backupCount; i += stride) backupDst[i] = backupSrc[i];
		return;
	}
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

// landmarks with more than 64 observations: one workgroup each
template <int MODE, typename ET>
__global__ __launch_bounds__(256) void big_lm_pass_kernel(DeviceGraph g, DeviceStructure st, DeviceSystem sys, Scalar lambda)
{
	__shared__ Scalar red[4][9];
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
static void launch_linearize_dm_t(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, int mode, Scalar lambda, hipStream_t s,
	const Scalar* backupSrc, Scalar* backupDst, size_t backupCount)
{
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
	static const bool separate = std::getenv("CUBA_HIP_SEPARATE_SCHUR_PASSES") != nullptr;     // A/B knob
	const int nbp = block_pass_groups(st.nOd, st.nHeavy);
	if (mode == 1 && g.Pf > 0 && st.nOd > 0 && st.nDiagProd == 0 && !separate)     // (duplicate observations: the block pass updates diagonal blocks after the pose pass)
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
	const Scalar* backupSrc, Scalar* backupDst, size_t backupCount)
{
	if (st.mixed && sizeof(Scalar) == 8) launch_linearize_dm_t<float>(g, st, sys, mode, lambda, s, backupSrc, backupDst, backupCount);
	else launch_linearize_dm_t<Scalar>(g, st, sys, mode, lambda, s, backupSrc, backupDst, backupCount);
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

// ---------------------------------------------------------------------------------------------------
// back substitution xl = inv(Hll + lambda I) (bl - sum_e Hpl_e^T xp[pose(e)]) and the landmark part of
// sum x (lambda x + b).  Replaces schurComplementPostKernel (:1029-1043) + half of computeScaleKernel (:1070-1091).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void edge_hplT_x(const LaneEdge& le, const Scalar* xp, Scalar c[3])
{
	// Hpl^T x = JL^T w (JP x)
	const EdgeLin& L = le.lin;
	Scalar v[3];
#pragma unroll
	for (int m = 0; m < 3; m++)
	{
		Scalar s = 0;
#pragma unroll
		for (int r = 0; r < 6; r++) s += L.JP[m][r] * xp[r];
		v[m] = le.wr * s;
	}
#pragma unroll
	for (int k = 0; k < 3; k++) c[k] = L.JL[0][k] * v[0] + L.JL[1][k] * v[1] + L.JL[2][k] * v[2];
}

__device__ __forceinline__ Scalar finish_landmark(const DeviceSystem& sys, int il, const Scalar csum[3], Scalar lambda)
{
	const Scalar* ls = sys.lm_sys + 9 * (size_t)il;
	Scalar inv[6], bl[3], cl[3], xl[3];
#pragma unroll
	for (int k = 0; k < 6; k++) inv[k] = ls[k];
#pragma unroll
	for (int k = 0; k < 3; k++) { bl[k] = ls[6 + k]; cl[k] = bl[k] - csum[k]; }
	Scalar sc = 0;
#pragma unroll
	for (int i = 0; i < 3; i++)
	{
		xl[i] = inv[sym3_idx(i, 0)] * cl[0] + inv[sym3_idx(i, 1)] * cl[1] + inv[sym3_idx(i, 2)] * cl[2];
		sys.xl[3 * (size_t)il + i] = xl[i];
		sc += xl[i] * (lambda * xl[i] + bl[i]);
	}
	return sc;
}

__global__ __launch_bounds__(LIN_BLOCK) void back_substitute_kernel(DeviceGraph g, DeviceStructure st, DeviceSystem sys, Scalar lambda)
{
	__shared__ Scalar lds_all[(LIN_BLOCK / WAVE) * WAVE * 3];
	const int lane = threadIdx.x & 63;
	const int wv = threadIdx.x >> 6;
	const int wave = blockIdx.x * (LIN_BLOCK / WAVE) + wv;
	if (wave >= st.nWaves) return;
	Scalar* lds = lds_all + wv * WAVE * 3;
	const int lm0 = st.wave_lm[2 * wave], lm1 = st.wave_lm[2 * wave + 1];
	const int e0 = g.lm_ptr[lm0], e1 = g.lm_ptr[lm1];
	const int e = e0 + lane;
	const bool valid = e < e1;
	int il = lm0, seg0 = 0, seg1 = 0;
	Scalar c[3] = { 0, 0, 0 };
	if (valid)
	{
		il = g.e_lm[e];
		if (il < g.Lf)
		{
			seg0 = g.lm_ptr[il] - e0;
			seg1 = g.lm_ptr[il + 1] - e0;
			const int ip = g.e_pose[e] & ~STEREO_BIT;
			if (ip < g.Pf)
			{
				LaneEdge le;
				linearize_edge(g, e, le);
				Scalar xp[6];
#pragma unroll
				for (int r = 0; r < 6; r++) xp[r] = sys.xp[6 * (size_t)ip + r];
				edge_hplT_x(le, xp, c);
			}
		}
	}
	const bool lmFree = valid && il < g.Lf;
#pragma unroll
	for (int k = 0; k < 3; k++) lds[lane * 3 + k] = c[k];
	wave_lds_sync();
	Scalar sc = 0;
	if (lmFree && lane == seg0)
	{
		Scalar cs[3] = { 0, 0, 0 };
		for (int j = seg0; j < seg1; j++)
		{
			cs[0] += lds[j * 3 + 0]; cs[1] += lds[j * 3 + 1]; cs[2] += lds[j * 3 + 2];
		}
		sc = finish_landmark(sys, il, cs, lambda);
	}
	sc = wave_sum(sc);
	if (lane == 0) sys.parts[wave] = sc;     // one partial per wave, summed by reduce_parts_kernel
}

__global__ __launch_bounds__(256) void big_back_substitute_kernel(DeviceGraph g, DeviceStructure st, DeviceSystem sys, Scalar lambda)
{
	__shared__ Scalar red[4][3];
	const int il = st.big_lm[blockIdx.x];
	if (il >= g.Lf)
	{
		if (threadIdx.x == 0) sys.parts[st.nWaves + blockIdx.x] = 0;      // (a fixed landmark has no increment, but its partial is summed)
		return;
	}
	const int e0 = g.lm_ptr[il], e1 = g.lm_ptr[il + 1];
	Scalar acc[3] = { 0, 0, 0 };
	for (int e = e0 + threadIdx.x; e < e1; e += 256)
	{
		const int ip = g.e_pose[e] & ~STEREO_BIT;
		if (ip >= g.Pf) continue;
		LaneEdge le;
		linearize_edge(g, e, le);
		Scalar xp[6], c[3];
#pragma unroll
		for (int r = 0; r < 6; r++) xp[r] = sys.xp[6 * (size_t)ip + r];
		edge_hplT_x(le, xp, c);
		acc[0] += c[0]; acc[1] += c[1]; acc[2] += c[2];
	}
#pragma unroll
	for (int k = 0; k < 3; k++) acc[k] = wave_sum(acc[k]);
	if ((threadIdx.x & 63) == 0)
#pragma unroll
		for (int k = 0; k < 3; k++) red[threadIdx.x >> 6][k] = acc[k];
	__syncthreads();
	if (threadIdx.x == 0)
	{
		Scalar cs[3];
#pragma unroll
		for (int k = 0; k < 3; k++) cs[k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
		sys.parts[st.nWaves + blockIdx.x] = finish_landmark(sys, il, cs, lambda);
	}
}

static void launch_back_substitute_kernels(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, Scalar lambda, hipStream_t s)
{
	if (st.nWaves > 0)
	{
		const int grid = (st.nWaves + (LIN_BLOCK / WAVE) - 1) / (LIN_BLOCK / WAVE);
		hipLaunchKernelGGL(back_substitute_kernel, dim3(grid), dim3(LIN_BLOCK), 0, s, g, st, sys, lambda);
	}
	if (st.nBig > 0)
		hipLaunchKernelGGL(big_back_substitute_kernel, dim3(st.nBig), dim3(256), 0, s, g, st, sys, lambda);
}

void launch_back_substitute(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, Scalar lambda, hipStream_t s)
{
	if (g.Lf <= 0) return;
	launch_back_substitute_kernels(g, st, sys, lambda, s);
	launch_reduce_parts(sys.parts, st.nWaves + st.nBig, sys.slots + NSLOT, s);
}

// sum x (lambda x + b), pose part and (stage API only) landmark part.  Ref: computeScaleKernel :1070-1091.
__device__ __forceinline__ void pose_scale_body(const DeviceGraph& g, const DeviceSystem& sys, Scalar lambda, Scalar* parts, int bid, int nb)
{
	Scalar acc = 0;
	for (int i = bid * 256 + threadIdx.x; i < g.Pf * 6; i += nb * 256)
	{
		const Scalar x = sys.xp[i];
		acc += x * (lambda * x + sys.bp[i]);
	}
	acc = wave_sum(acc);
	if ((threadIdx.x & 63) == 0) parts[bid * 4 + (threadIdx.x >> 6)] = acc;
}

__global__ __launch_bounds__(256) void pose_scale_kernel(DeviceGraph g, DeviceSystem sys, Scalar lambda, Scalar* parts)
{
	pose_scale_body(g, sys, lambda, parts, blockIdx.x, gridDim.x);
}

// Evaluation of an LM trial in one launch: the first nRes workgroups sum the robust chi2 at the updated estimate, the others the
// pose part of the gain-ratio denominator (same partials, in the same places of their arrays, as the two separate kernels).
__global__ __launch_bounds__(256) void eval_trial_kernel(DeviceGraph g, DeviceSystem sys, Scalar lambda, Scalar* resParts, int nRes, Scalar* scaleParts, int nScale)
{
	if ((int)blockIdx.x < nRes) residual_chi2_body(g, resParts, nullptr, blockIdx.x, nRes);
	else pose_scale_body(g, sys, lambda, scaleParts, blockIdx.x - nRes, nScale);
}

// Second stage of the three sums of a trial (landmark part of the denominator from the back-substitution, chi2, pose part) and
// the report to the host in one launch: each sum is added exactly as reduce_parts_kernel adds it; the results go into the
// mapped host block, the ticket follows them.
__global__ __launch_bounds__(1024) void reduce_report_kernel(DeviceSystem sys, const Scalar* pA, int nA, Scalar* oA, const Scalar* pB, int nB, Scalar* oB,
	const Scalar* pC, int nC, Scalar* oC)
{
	__shared__ Scalar sh[3][16];
	// the three sums side by side: thread shares first, then one barrier for all of them (each is added exactly as
	// reduce_parts_kernel adds it)
	const Scalar vA = wave_sum(parts_thread_sum(pA, nA)), vB = wave_sum(parts_thread_sum(pB, nB)), vC = wave_sum(parts_thread_sum(pC, nC));
	if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = vA; sh[1][threadIdx.x >> 6] = vB; sh[2][threadIdx.x >> 6] = vC; }
	__syncthreads();
	if (threadIdx.x < 64)
	{
		const bool in = threadIdx.x < 16;
		const Scalar tA = wave_sum(in ? sh[0][threadIdx.x] : Scalar(0)), tB = wave_sum(in ? sh[1][threadIdx.x] : Scalar(0)), tC = wave_sum(in ? sh[2][threadIdx.x] : Scalar(0));
		store_slot_group(oA, tA); store_slot_group(oB, tB); store_slot_group(oC, tC);
	}
	__threadfence_system();          // every writer's results before the ticket
	__syncthreads();
	if (threadIdx.x == 0 && sys.host_flags)
	{
		sys.host_flags[0] = *sys.fail; sys.host_flags[1] = *sys.iters; sys.host_flags[2] = *sys.done;
		__threadfence_system();
		sys.host_flags[3] = ++(*sys.ticket);
	}
}

__global__ __launch_bounds__(256) void landmark_scale_kernel(DeviceGraph g, DeviceSystem sys, Scalar lambda, Scalar* parts)
{
	Scalar acc = 0;
	for (int i = blockIdx.x * 256 + threadIdx.x; i < g.Lf * 3; i += gridDim.x * 256)
	{
		const Scalar x = sys.xl[i];
		acc += x * (lambda * x + sys.lm_sys[9 * (size_t)(i / 3) + 6 + (i % 3)]);
	}
	acc = wave_sum(acc);
	if ((threadIdx.x & 63) == 0) parts[blockIdx.x * 4 + (threadIdx.x >> 6)] = acc;
}

void launch_pose_scale(const DeviceGraph& g, const DeviceSystem& sys, Scalar lambda, Scalar* slots, hipStream_t s)
{
	const int grid = g.Pf > 0 ? min((g.Pf * 6 + 255) / 256, 256) : 0;
	if (grid > 0) hipLaunchKernelGGL(pose_scale_kernel, dim3(grid), dim3(256), 0, s, g, sys, lambda, sys.parts);
	launch_reduce_parts(sys.parts, grid * 4, slots, s);
}

void launch_landmark_scale(const DeviceGraph& g, const DeviceSystem& sys, Scalar lambda, Scalar* slots, hipStream_t s)
{
	const int grid = g.Lf > 0 ? min((g.Lf * 3 + 255) / 256, 1024) : 0;
	if (grid > 0) hipLaunchKernelGGL(landmark_scale_kernel, dim3(grid), dim3(256), 0, s, g, sys, lambda, sys.parts);
	launch_reduce_parts(sys.parts, grid * 4, slots, s);
}

// ---------------------------------------------------------------------------------------------------
// manifold update.  Ref: updatePosesKernel / updateLandmarksKernel :1045-1068.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void update_poses_kernel(DeviceGraph g, DeviceSystem sys)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= g.Pf) return;
	Scalar upd[6], q[4], t[3];
#pragma unroll
	for (int k = 0; k < 6; k++) upd[k] = sys.xp[6 * (size_t)i + k];
#pragma unroll
	for (int k = 0; k < 4; k++) q[k] = g.q[4 * (size_t)i + k];
#pragma unroll
	for (int k = 0; k < 3; k++) t[k] = g.t[3 * (size_t)i + k];
	pose_exp_update(upd, q, t);
#pragma unroll
	for (int k = 0; k < 4; k++) g.q[4 * (size_t)i + k] = q[k];
#pragma unroll
	for (int k = 0; k < 3; k++) g.t[3 * (size_t)i + k] = t[k];
}

__global__ __launch_bounds__(256) void update_landmarks_kernel(DeviceGraph g, DeviceSystem sys)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i < g.Lf * 3) g.Xw[i] += sys.xl[i];
}

// both updates in one launch: the first workgroups take the poses, the rest the landmark coordinates
__global__ __launch_bounds__(256) void update_state_kernel(DeviceGraph g, DeviceSystem sys, int poseBlocks)
{
	if ((int)blockIdx.x < poseBlocks)
	{
		const int i = blockIdx.x * 256 + threadIdx.x;
		if (i >= g.Pf) return;
		Scalar upd[6], q[4], t[3];
#pragma unroll
		for (int k = 0; k < 6; k++) upd[k] = sys.xp[6 * (size_t)i + k];
#pragma unroll
		for (int k = 0; k < 4; k++) q[k] = g.q[4 * (size_t)i + k];
#pragma unroll
		for (int k = 0; k < 3; k++) t[k] = g.t[3 * (size_t)i + k];
		pose_exp_update(upd, q, t);
#pragma unroll
		for (int k = 0; k < 4; k++) g.q[4 * (size_t)i + k] = q[k];
#pragma unroll
		for (int k = 0; k < 3; k++) g.t[3 * (size_t)i + k] = t[k];
		return;
	}
	const int i = (blockIdx.x - poseBlocks) * 256 + threadIdx.x;
	if (i < g.Lf * 3) g.Xw[i] += sys.xl[i];
}

void launch_update_state(const DeviceGraph& g, const DeviceSystem& sys, hipStream_t s)
{
	const int pb = (g.Pf + 255) / 256, lb = (g.Lf * 3 + 255) / 256;
	if (pb + lb > 0) hipLaunchKernelGGL(update_state_kernel, dim3(pb + lb), dim3(256), 0, s, g, sys, pb);
}

// ---------------------------------------------------------------------------------------------------
// Fused tail of an LM trial (optimize() only): back-substitution, update and evaluation of the trial in ONE pass over the edges.
// A landmark's wave computes xl from the PRE-update estimate (read from the state backup the landmark pass of this trial made,
// so the pose-update workgroups of the same launch may overwrite the live state meanwhile), stores xl and Xw + xl, and then every
// lane evaluates its own edge at the updated estimate: its pose is updated in registers by the very code the pose-update
// workgroups run (pose_exp_update on the same inputs), its landmark comes through LDS from the head lane.  Replaces
// schurComplementPostKernel + updatePosesKernel + updateLandmarksKernel + computeActiveErrorsKernel + computeScaleKernel
// (cuda_block_solver.cu:1029-1091, 733-786) for one trial: the edge stream is read twice per trial instead of three times.
// Roles by workgroup index: [0, nLmGroups) landmark waves, then poseBlocks pose-update workgroups, then nScale workgroups for the
// pose part of the gain-ratio denominator.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ Scalar finish_landmark_x(const DeviceSystem& sys, int il, const Scalar csum[3], Scalar lambda, Scalar xl[3])
{
	const Scalar* ls = sys.lm_sys + 9 * (size_t)il;
	Scalar inv[6], bl[3], cl[3];
#pragma unroll
	for (int k = 0; k < 6; k++) inv[k] = ls[k];
#pragma unroll
	for (int k = 0; k < 3; k++) { bl[k] = ls[6 + k]; cl[k] = bl[k] - csum[k]; }
	Scalar sc = 0;
#pragma unroll
	for (int i = 0; i < 3; i++)
	{
		xl[i] = inv[sym3_idx(i, 0)] * cl[0] + inv[sym3_idx(i, 1)] * cl[1] + inv[sym3_idx(i, 2)] * cl[2];
		sys.xl[3 * (size_t)il + i] = xl[i];
		sc += xl[i] * (lambda * xl[i] + bl[i]);
	}
	return sc;
}

// robust chi2 term of an edge at the UPDATED estimate: (q0, t0) is the pre-update pose, upd its increment (poseFree = false: fixed pose)
__device__ __forceinline__ Scalar updated_edge_rho(const DeviceGraph& g, const Scalar q0[4], const Scalar t0[3], const Scalar cam[5], const Scalar upd[6], bool poseFree,
	const Scalar Xn[3], const Scalar meas[3], Scalar w, bool stereo)
{
	Scalar q[4] = { q0[0], q0[1], q0[2], q0[3] }, t[3] = { t0[0], t0[1], t0[2] }, r[3], Xc[3];
	if (poseFree) pose_exp_update(upd, q, t);
	const Scalar ee = w * edge_residual(q, t, cam, Xn, meas, stereo, r, Xc);
	return robust_rho(stereo ? g.rk[1].kind : g.rk[0].kind, stereo ? g.rk[1].delta : g.rk[0].delta, ee);
}

__device__ __forceinline__ void update_pose_rows(const DeviceGraph& g, const DeviceSystem& sys, const Scalar* __restrict__ old, int i)
{
	Scalar upd[6], q[4], t[3];
#pragma unroll
	for (int k = 0; k < 6; k++) upd[k] = sys.xp[6 * (size_t)i + k];
#pragma unroll
	for (int k = 0; k < 4; k++) q[k] = old[4 * (size_t)i + k];
#pragma unroll
	for (int k = 0; k < 3; k++) t[k] = old[4 * (size_t)g.Pt + 3 * (size_t)i + k];
	pose_exp_update(upd, q, t);
#pragma unroll
	for (int k = 0; k < 4; k++) g.q[4 * (size_t)i + k] = q[k];
#pragma unroll
	for (int k = 0; k < 3; k++) g.t[3 * (size_t)i + k] = t[k];
}

__global__ __launch_bounds__(LIN_BLOCK) void trial_tail_kernel(DeviceGraph g, DeviceStructure st, DeviceSystem sys, Scalar lambda,
	const Scalar* __restrict__ old, Scalar* __restrict__ scParts, Scalar* __restrict__ chiParts, int nLmGroups, int poseBlocks, Scalar* __restrict__ scaleParts, int nScale)
{
	if ((int)blockIdx.x >= nLmGroups)
	{
		const int b = blockIdx.x - nLmGroups;
		if (b < poseBlocks)
		{
			const int i = b * LIN_BLOCK + threadIdx.x;
			if (i < g.Pf) update_pose_rows(g, sys, old, i);
		}
		else pose_scale_body(g, sys, lambda, scaleParts, b - poseBlocks, nScale);
		return;
	}
	__shared__ Scalar lds_all[(LIN_BLOCK / WAVE) * WAVE * 3];
	__shared__ Scalar wpart[2][LIN_BLOCK / WAVE];
	const int lane = threadIdx.x & 63;
	const int wv = threadIdx.x >> 6;
	const int wave = blockIdx.x * (LIN_BLOCK / WAVE) + wv;
	const bool waveOn = wave < st.nWaves;
	Scalar* lds = lds_all + wv * WAVE * 3;
	const Scalar* qo = old; const Scalar* to = old + 4 * (size_t)g.Pt; const Scalar* Xo = old + 7 * (size_t)g.Pt;
	const int lm0 = waveOn ? st.wave_lm[2 * wave] : 0, lm1 = waveOn ? st.wave_lm[2 * wave + 1] : 0;
	const int e0 = waveOn ? g.lm_ptr[lm0] : 0, e1 = waveOn ? g.lm_ptr[lm1] : 0;
	const int e = e0 + lane;
	const bool valid = waveOn && e < e1;
	int il = lm0, ip = 0, seg0 = 0, seg1 = 0;
	bool stereo = false, poseFree = false;
	Scalar q[4] = { 0, 0, 0, 1 }, t[3] = { 0, 0, 0 }, cam[5] = { 1, 1, 0, 0, 0 }, Xw[3] = { 0, 0, 1 }, meas[3] = { 0, 0, 0 }, xp[6] = { 0, 0, 0, 0, 0, 0 }, w = 0;
	Scalar c[3] = { 0, 0, 0 };
	if (valid)
	{
		const int pe = g.e_pose[e];
		stereo = (pe & STEREO_BIT) != 0;
		ip = pe & ~STEREO_BIT;
		il = g.e_lm[e];
		poseFree = ip < g.Pf;
#pragma unroll
		for (int i = 0; i < 4; i++) q[i] = qo[4 * (size_t)ip + i];
#pragma unroll
		for (int i = 0; i < 3; i++) t[i] = to[3 * (size_t)ip + i];
#pragma unroll
		for (int i = 0; i < 5; i++) cam[i] = g.cam[5 * (size_t)ip + i];
#pragma unroll
		for (int i = 0; i < 3; i++) Xw[i] = Xo[3 * (size_t)il + i];
		meas[0] = g.e_mu[e]; meas[1] = g.e_mv[e]; meas[2] = g.e_mr[e];
		w = g.e_w[e];
		if (poseFree)
		{
#pragma unroll
			for (int r = 0; r < 6; r++) xp[r] = sys.xp[6 * (size_t)ip + r];
		}
		if (il < g.Lf)
		{
			seg0 = g.lm_ptr[il] - e0;
			seg1 = g.lm_ptr[il + 1] - e0;
			if (poseFree)
			{
				// Hpl^T xp at the linearisation point (the pre-update estimate), exactly as back_substitute_kernel forms it
				LaneEdge le;
				Scalar Xc[3];
				const Scalar ss = edge_residual(q, t, cam, Xw, meas, stereo, le.lin.r, Xc);
				le.wr = w * robust_weight(stereo ? g.rk[1].kind : g.rk[0].kind, stereo ? g.rk[1].delta : g.rk[0].delta, w * ss);
				const Rot3 R = quat_to_rot(q[0], q[1], q[2], q[3]);
				edge_jacobians(Xc, R, cam, stereo, le.lin);
				edge_hplT_x(le, xp, c);
			}
		}
	}
	const bool lmFree = valid && il < g.Lf;
#pragma unroll
	for (int k = 0; k < 3; k++) lds[lane * 3 + k] = c[k];
	wave_lds_sync();
	Scalar sc = 0, xl[3] = { 0, 0, 0 };
	if (lmFree && lane == seg0)
	{
		Scalar cs[3] = { 0, 0, 0 };
		for (int j = seg0; j < seg1; j++)
		{
			cs[0] += lds[j * 3 + 0]; cs[1] += lds[j * 3 + 1]; cs[2] += lds[j * 3 + 2];
		}
		sc = finish_landmark_x(sys, il, cs, lambda, xl);
#pragma unroll
		for (int k = 0; k < 3; k++) g.Xw[3 * (size_t)il + k] = Xw[k] + xl[k];     // (= update_landmarks: Xw += xl)
	}
	wave_lds_sync();                     // every segment sum has been read: the head lanes may reuse their own LDS slots
	if (lmFree && lane == seg0)
	{
#pragma unroll
		for (int k = 0; k < 3; k++) lds[lane * 3 + k] = xl[k];
	}
	wave_lds_sync();
	Scalar rho = 0;
	if (valid)
	{
		Scalar Xn[3] = { Xw[0], Xw[1], Xw[2] };
		if (lmFree)
		{
#pragma unroll
			for (int k = 0; k < 3; k++) Xn[k] = Xw[k] + lds[seg0 * 3 + k];
		}
		rho = updated_edge_rho(g, q, t, cam, xp, poseFree, Xn, meas, w, stereo);
	}
	sc = wave_sum(sc); rho = wave_sum(rho);
	if (lane == 0) { wpart[0][wv] = sc; wpart[1][wv] = rho; }
	__syncthreads();
	if (threadIdx.x == 0)
	{
		scParts[blockIdx.x] = (wpart[0][0] + wpart[0][1]) + (wpart[0][2] + wpart[0][3]);
		chiParts[blockIdx.x] = (wpart[1][0] + wpart[1][1]) + (wpart[1][2] + wpart[1][3]);
	}
}

// landmarks with more than 64 observations: one workgroup each (free or fixed: their edges are evaluated either way)
__global__ __launch_bounds__(256) void big_trial_tail_kernel(DeviceGraph g, DeviceStructure st, DeviceSystem sys, Scalar lambda,
	const Scalar* __restrict__ old, Scalar* __restrict__ scParts, Scalar* __restrict__ chiParts)
{
	__shared__ Scalar red[4][3];
	__shared__ Scalar xsh[3];
	__shared__ Scalar rsh[4];
	DeviceGraph go = g;                   // the pre-update estimate
	go.q = const_cast<Scalar*>(old); go.t = go.q + 4 * (size_t)g.Pt; go.Xw = go.q + 7 * (size_t)g.Pt;
	const int il = st.big_lm[blockIdx.x];
	const int e0 = g.lm_ptr[il], e1 = g.lm_ptr[il + 1];
	const bool lmFree = il < g.Lf;
	Scalar acc[3] = { 0, 0, 0 };
	if (lmFree)
	{
		for (int e = e0 + threadIdx.x; e < e1; e += 256)
		{
			const int ip = g.e_pose[e] & ~STEREO_BIT;
			if (ip >= g.Pf) continue;
			LaneEdge le;
			linearize_edge(go, e, le);
			Scalar xp[6], c[3];
#pragma unroll
			for (int r = 0; r < 6; r++) xp[r] = sys.xp[6 * (size_t)ip + r];
			edge_hplT_x(le, xp, c);
			acc[0] += c[0]; acc[1] += c[1]; acc[2] += c[2];
		}
	}
#pragma unroll
	for (int k = 0; k < 3; k++) acc[k] = wave_sum(acc[k]);
	if ((threadIdx.x & 63) == 0)
#pragma unroll
		for (int k = 0; k < 3; k++) red[threadIdx.x >> 6][k] = acc[k];
	__syncthreads();
	if (threadIdx.x == 0)
	{
		Scalar xl[3] = { 0, 0, 0 }, sc = 0;
		if (lmFree)
		{
			Scalar cs[3];
#pragma unroll
			for (int k = 0; k < 3; k++) cs[k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
			sc = finish_landmark_x(sys, il, cs, lambda, xl);
#pragma unroll
			for (int k = 0; k < 3; k++) g.Xw[3 * (size_t)il + k] = go.Xw[3 * (size_t)il + k] + xl[k];
		}
		scParts[blockIdx.x] = sc;
#pragma unroll
		for (int k = 0; k < 3; k++) xsh[k] = xl[k];
	}
	__syncthreads();
	Scalar Xn[3];
#pragma unroll
	for (int k = 0; k < 3; k++) Xn[k] = go.Xw[3 * (size_t)il + k] + xsh[k];
	Scalar rho = 0;
	for (int e = e0 + threadIdx.x; e < e1; e += 256)
	{
		const int pe = g.e_pose[e];
		const bool stereo = (pe & STEREO_BIT) != 0;
		const int ip = pe & ~STEREO_BIT;
		Scalar q[4], t[3], cam[5], meas[3], xp[6];
		load_pose(go, ip, q, t, cam);
		meas[0] = g.e_mu[e]; meas[1] = g.e_mv[e]; meas[2] = g.e_mr[e];
		const bool poseFree = ip < g.Pf;
#pragma unroll
		for (int r = 0; r < 6; r++) xp[r] = poseFree ? sys.xp[6 * (size_t)ip + r] : Scalar(0);
		rho += updated_edge_rho(g, q, t, cam, xp, poseFree, Xn, meas, g.e_w[e], stereo);
	}
	rho = wave_sum(rho);
	if ((threadIdx.x & 63) == 0) rsh[threadIdx.x >> 6] = rho;
	__syncthreads();
	if (threadIdx.x == 0) chiParts[blockIdx.x] = (rsh[0] + rsh[1]) + (rsh[2] + rsh[3]);
}

size_t trial_tail_parts(const DeviceGraph& g, const DeviceStructure& st)
{
	const size_t nA = ((size_t)(st.nWaves + LIN_BLOCK / WAVE - 1) / (LIN_BLOCK / WAVE) + st.nBig + 63) / 64 * 64;
	return 2 * nA + 4 * 256 + 64;
}

void launch_trial_tail_fused(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, Scalar lambda, const Scalar* old, hipStream_t s)
{
	const int nLm = (st.nWaves + LIN_BLOCK / WAVE - 1) / (LIN_BLOCK / WAVE);
	const int nA = nLm + st.nBig;
	Scalar* scParts = sys.parts;
	Scalar* chiParts = sys.parts + (size_t)(nA + 63) / 64 * 64;
	Scalar* scaleParts = chiParts + (size_t)(nA + 63) / 64 * 64;
	const int poseBlocks = (g.Pf + LIN_BLOCK - 1) / LIN_BLOCK;
	const int nScale = g.Pf > 0 ? min((g.Pf * 6 + 255) / 256, 256) : 0;
	if (nLm + poseBlocks + nScale > 0)
		hipLaunchKernelGGL(trial_tail_kernel, dim3(nLm + poseBlocks + nScale), dim3(LIN_BLOCK), 0, s, g, st, sys, lambda, old, scParts, chiParts, nLm, poseBlocks, scaleParts, nScale);
	if (st.nBig > 0) hipLaunchKernelGGL(big_trial_tail_kernel, dim3(st.nBig), dim3(256), 0, s, g, st, sys, lambda, old, scParts + nLm, chiParts + nLm);
	hipLaunchKernelGGL(reduce_report_kernel, dim3(1), dim3(1024), 0, s, sys, scParts, nA, sys.slots + NSLOT, chiParts, nA, sys.slots, scaleParts, 4 * nScale, sys.slots + 3 * NSLOT);
}

// Everything between a converged reduced solve and the LM decision in four launches: back-substitution, update, evaluation of
// the trial (chi2 + pose part of the gain-ratio denominator), then the second stage of the three sums with the report to the
// host.  The stage API runs the same kernels one call at a time (eight launches); per LM trial that is ~26 us more.
// The partial sums share sys.parts: [0, nWaves + nBig) back-substitution, then 2048 for chi2, then 1024 for the pose part.
void launch_trial_tail(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, Scalar lambda, hipStream_t s)
{
	const int nA = g.Lf > 0 ? st.nWaves + st.nBig : 0;
	Scalar* resParts = sys.parts + (nA + 63) / 64 * 64;
	Scalar* scaleParts = resParts + 2048;
	if (g.Lf > 0) launch_back_substitute_kernels(g, st, sys, lambda, s);
	launch_update_state(g, sys, s);
	const int n = g.e_end - g.e_begin;
	const int nRes = n > 0 ? min((n + 255) / 256, 2048) : 0;
	const int nScale = g.Pf > 0 ? min((g.Pf * 6 + 255) / 256, 256) : 0;
	if (nRes + nScale > 0) hipLaunchKernelGGL(eval_trial_kernel, dim3(nRes + nScale), dim3(256), 0, s, g, sys, lambda, resParts, nRes, scaleParts, nScale);
	hipLaunchKernelGGL(reduce_report_kernel, dim3(1), dim3(1024), 0, s, sys, sys.parts, nA, sys.slots + NSLOT, resParts, nRes, sys.slots, scaleParts, 4 * nScale, sys.slots + 3 * NSLOT);
}

void launch_update_poses(const DeviceGraph& g, const DeviceSystem& sys, hipStream_t s)
{
	if (g.Pf > 0) hipLaunchKernelGGL(update_poses_kernel, dim3((g.Pf + 255) / 256), dim3(256), 0, s, g, sys);
}

void launch_update_landmarks(const DeviceGraph& g, const DeviceSystem& sys, hipStream_t s)
{
	if (g.Lf > 0) hipLaunchKernelGGL(update_landmarks_kernel, dim3((g.Lf * 3 + 255) / 256), dim3(256), 0, s, g, sys);
}

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
		if (ROWCOPY)
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

void launch_hsc_expand(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, hipStream_t s)
{
	const size_t total = (size_t)g.Pf * st.ell_m * 20 * 36;
	if (total) hipLaunchKernelGGL(hsc_expand_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, st, sys, total);
}

void launch_pcg_setup_expand(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, Scalar lambda, hipStream_t s,
	const Scalar* copySrc, Scalar* copyDst, size_t copyCount)
{
	if (g.Pf <= 0) return;
	const size_t total = (size_t)g.Pf * st.ell_m * 20 * 36;
	const int nSetup = (g.Pf + PCG_SETUP_POSES - 1) / PCG_SETUP_POSES;
	const unsigned nExpand = (unsigned)((total + 255) / 256);
	const size_t pairs = copySrc ? copyCount / 2 : 0;            // (coarse dimensions are even)
	const unsigned nCopy = pairs ? (unsigned)std::min<size_t>(1024, (pairs + 255) / 256) : 0;
	hipLaunchKernelGGL(pcg_setup_expand_kernel, dim3(nSetup + nExpand + nCopy), dim3(256), 0, s, g, st, sys, lambda, nSetup, total, nExpand,
		reinterpret_cast<const Scalar2*>(copySrc), reinterpret_cast<Scalar2*>(copyDst), pairs);
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
template <int ROWS, int MIN_WAVES>
__global__ __launch_bounds__(128 * ROWS, MIN_WAVES) void pcg_spmv_kernel(DeviceGraph g, DeviceStructure st, DeviceSystem sys, int k, int maxIter, Scalar tol2)
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

// One wave per block row (large graphs).  With two waves per row S2M / G4M need 10 000 / 20 000 waves of ~5 KB each, i.e.
// 2.5 / 5 rounds of resident waves whose life is two dependent round trips + a barrier: the launch is bound by
// wave slots x latency, not by bytes.  Here lane (slot, r2) = (lane / 3, lane % 3) takes block rows 2 r2 and 2 r2 + 1 of one of
// the 20 entry slots: half the waves, twice the bytes per wave, no cross-wave exchange.  Two levels of the fixed-width row are
// in flight at once (96 VGPRs of operands: occupancy 3); the third level, which few rows have, follows.
template <int N>
__device__ __forceinline__ void spmv_batch2(const DeviceSystem& sys, const Scalar* pold, const int2* e, const Scalar* Arow, int r2,
	Scalar& az0, Scalar& ap0, Scalar& az1, Scalar& ap1)
{
	Scalar2 a0v[N][3], a1v[N][3], zv[N][3], pv[N][3];
#pragma unroll
	for (int n = 0; n < N; n++)
	{
		const size_t j = e[n].y >= 0 ? e[n].y : 0;
		const Scalar2* A2 = reinterpret_cast<const Scalar2*>(Arow + (size_t)n * (20 * 36) + 12 * r2);
		const Scalar2* z2 = reinterpret_cast<const Scalar2*>(sys.z + 6 * j);
		const Scalar2* p2 = reinterpret_cast<const Scalar2*>(pold + 6 * j);
		const bool on = e[n].y >= 0;       // padding slots: matrix entry not fetched
#pragma unroll
		for (int c = 0; c < 3; c++)
		{
			a0v[n][c] = on ? A2[c] : Scalar2{ 0, 0 }; a1v[n][c] = on ? A2[3 + c] : Scalar2{ 0, 0 };
			zv[n][c] = z2[c]; pv[n][c] = p2[c];
		}
	}
#pragma unroll
	for (int n = 0; n < N; n++)
	{
#pragma unroll
		for (int c = 0; c < 3; c++)
		{
			az0 += a0v[n][c].x * zv[n][c].x; ap0 += a0v[n][c].x * pv[n][c].x;
			az0 += a0v[n][c].y * zv[n][c].y; ap0 += a0v[n][c].y * pv[n][c].y;
			az1 += a1v[n][c].x * zv[n][c].x; ap1 += a1v[n][c].x * pv[n][c].x;
			az1 += a1v[n][c].y * zv[n][c].y; ap1 += a1v[n][c].y * pv[n][c].y;
		}
	}
}

template <int ROWS>
__global__ __launch_bounds__(64 * ROWS) void pcg_spmv_row_kernel(DeviceGraph g, DeviceStructure st, DeviceSystem sys, int k, int maxIter, Scalar tol2)
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
		if (cnt >= 2) spmv_batch2<2>(sys, pold, e, Arow, r2, az0, ap0, az1, ap1);
		else if (cnt == 1) spmv_batch2<1>(sys, pold, e, Arow, r2, az0, ap0, az1, ap1);
		if (cnt == 3) spmv_batch2<1>(sys, pold, e + 2, Arow + 2 * (20 * 36), r2, az0, ap0, az1, ap1);
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

// 16 x 16 x 4 matrix-core step in the library's Scalar: v_mfma_f64_16x16x4_f64 (fp64 build) / v_mfma_f32_16x16x4_f32 (fp32 build).
// Lane l feeds A[l & 15][l >> 4] and B[l >> 4][l & 15]; it receives 4 results of column l & 15, in rows (l >> 4) + 4 q (f64)
// or 4 (l >> 4) + q (f32), q = 0..3.
#ifdef CUBA_HIP_FLOAT32
typedef float MfmaAcc __attribute__((ext_vector_type(4)));
__device__ __forceinline__ MfmaAcc mfma_16x16x4(float a, float b, MfmaAcc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int mfma_row(int lane, int q) { return 4 * (lane >> 4) + q; }
#else
typedef double MfmaAcc __attribute__((ext_vector_type(4)));
__device__ __forceinline__ MfmaAcc mfma_16x16x4(double a, double b, MfmaAcc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int mfma_row(int lane, int q) { return (lane >> 4) + 4 * q; }
#endif
__device__ __forceinline__ MfmaAcc mfma_zero() { return MfmaAcc{ 0, 0, 0, 0 }; }
__device__ __forceinline__ Scalar mfma_get(const MfmaAcc& v, int q) { return q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w; }

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
	const int bkPad = (bk + 1) & ~1;
	Scalar d[4];
#pragma unroll
	for (int u = 0; u < 4; u++) d[u] = Dcur[rb + 8 * u][c];
	for (int p = 0; p < bkPad; p += 2)
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
			if (r < bkPad && c < bkPad) { d[u] = v; Dnext[r][c] = v; }
		}
		__syncthreads();
		Scalar (*tmp)[GJ_B + 1] = Dcur; Dcur = Dnext; Dnext = tmp;
	}
	return Dcur;
}

// Inverse of the first pivot block (rows / columns [0, bk)) of the sweep -> pivOut[c * 32 + r]; every later pivot block is
// inverted by the step before it (below).
__global__ __launch_bounds__(256) void dense_gj_first_pivot_kernel(const Scalar* __restrict__ src, int n, int bk, Scalar* __restrict__ pivOut)
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

// One blocked Gauss-Jordan step with pivot rows/cols [p0, p0+bk), p0 a multiple of GJ_B: dst = GJ_step(src). After the
// last step dst = A^-1.  One 256-thread workgroup per 32x32 output tile (or per colsPerGroup tiles of one tile row); thread
// (r, cb) owns the elements (r, cb + 8u), u = 0..3, of every 32x32 array.  The step time is latency, not flops (n/32
// dependent launches), so:
//   * every global load of the kernel is issued before the first use (clamped addresses, selected afterwards);
//   * the inverse of the pivot block comes in ready-made (pivIn): the workgroup that produced the NEXT pivot block in the
//     previous step -- tile (p0/32 + 1, p0/32 + 1) is final for this purpose once step p0 has updated it -- inverted it right
//     away (look-ahead).  One workgroup runs the 16-step chain per launch instead of all of them (twice as slow when two
//     workgroups share a CU), and it has its CU nearly to itself by then: 15.9 -> ~10 us per step;
//   * the two 32x32x32 products run on the matrix cores.
__global__ __launch_bounds__(256) void dense_gj_step_kernel(const Scalar* __restrict__ src, Scalar* __restrict__ dst, int n, int p0, int bk, int colsPerGroup,
	const Scalar* __restrict__ pivIn, Scalar* __restrict__ pivOut)
{
	__shared__ Scalar D[GJ_B][GJ_B + 1];
	__shared__ Scalar D2[GJ_B][GJ_B + 1];
	__shared__ Scalar Apj[GJ_B][GJ_B + 1];
	__shared__ Scalar R[GJ_B][GJ_B + 1];
	__shared__ Scalar F[GJ_B][GJ_B + 1];
	const int tid = threadIdx.x;
	TRACE_DECL
	TRACE_MARK();
	const int tiles = (n + GJ_B - 1) / GJ_B;
	const int jt0 = blockIdx.x * colsPerGroup, jt1 = min(tiles, jt0 + colsPerGroup);
	const int i0 = blockIdx.y * GJ_B;
	int j0 = jt0 * GJ_B;
	const int r = tid & 31, cb = tid >> 5;
	const bool rowTile = i0 == p0;                          // this workgroup's tiles lie in the pivot rows
	const int pNext = p0 + GJ_B;                            // look-ahead: the tile (pNext, pNext) is the next pivot block
	const bool aheadRow = i0 == pNext && pNext < n;
	Scalar dv[4], av[4], fv[4], sv[4], keep[4] = { 0, 0, 0, 0 };
#pragma unroll
	for (int u = 0; u < 4; u++)
	{
		const int c = cb + 8 * u;
		const size_t pr = (size_t)min(p0 + r, n - 1), pc = (size_t)min(p0 + c, n - 1);
		const size_t gi = (size_t)min(i0 + r, n - 1), gj = (size_t)min(j0 + c, n - 1);
		dv[u] = pivIn[c * GJ_B + r];   // D[r][c]   = inverse of the pivot block A[p0.., p0..]
		av[u] = src[gj * n + pr];      // Apj[r][c] = A[p0+r, j0+c]
		fv[u] = src[pc * n + gi];      // F[r][c]   = A[i0+r, p0+c]
		sv[u] = src[gj * n + gi];      // own tile
	}
#pragma unroll
	for (int u = 0; u < 4; u++)
	{
		const int c = cb + 8 * u;
		D[r][c] = dv[u];
		Apj[r][c] = (r < bk && j0 + c < n) ? av[u] : Scalar(0);
		F[r][c] = (c < bk && i0 + r < n) ? fv[u] : Scalar(0);
	}
	__syncthreads();
	TRACE_MARK();
	Scalar (*Dcur)[GJ_B + 1] = D;
	// The two 32 x 32 x 32 tile products on the matrix cores: wave w owns the 16 x 16 output tile (w >> 1, w & 1), eight
	// v_mfma_f64_16x16x4_f64 k-steps each (operands straight from LDS, one number per lane: A[i = lane & 15][k = lane >> 4],
	// B[k = lane >> 4][j = lane & 15]).  This is the one GEMM-shaped piece of the whole path.
	const int wv = tid >> 6, lane = tid & 63;
	const int ti = wv >> 1, tj = wv & 1;
	// A workgroup walks over colsPerGroup column tiles (1 up to n = 768: the sweep is latency there and the grid small;
	// more beyond).
	for (int jt = jt0; jt < jt1; jt++)
	{
		j0 = jt * GJ_B;
		const bool colTile = j0 == p0;                      // this tile lies in the pivot columns
		// next column tile of this workgroup: loads in flight under the products of the current one
		Scalar avN[4], svN[4];
		if (jt + 1 < jt1)
		{
#pragma unroll
			for (int u = 0; u < 4; u++)
			{
				const size_t pr = (size_t)min(p0 + r, n - 1), gi = (size_t)min(i0 + r, n - 1), gj = (size_t)min(j0 + GJ_B + cb + 8 * u, n - 1);
				avN[u] = src[gj * n + pr];
				svN[u] = src[gj * n + gi];
			}
		}
		// R = Dinv * Apj (not needed by the tiles of the pivot columns)
		if (!colTile)
		{
			MfmaAcc acc = mfma_zero();
#pragma unroll
			for (int s4 = 0; s4 < GJ_B; s4 += 4)
				acc = mfma_16x16x4(Dcur[16 * ti + (lane & 15)][s4 + (lane >> 4)], Apj[s4 + (lane >> 4)][16 * tj + (lane & 15)], acc);
#pragma unroll
			for (int q = 0; q < 4; q++)
			{
				const int rr = 16 * ti + mfma_row(lane, q);
				R[rr][16 * tj + (lane & 15)] = rr < bk ? mfma_get(acc, q) : Scalar(0);
			}
		}
		__syncthreads();
		Scalar out[4];
		if (rowTile && colTile)
		{
#pragma unroll
			for (int u = 0; u < 4; u++) out[u] = dv[u];
		}
		else if (rowTile)
		{
#pragma unroll
			for (int u = 0; u < 4; u++) out[u] = R[r][cb + 8 * u];
		}
		else
		{
			Scalar (*B)[GJ_B + 1] = colTile ? Dcur : R;            // pivot columns: -F Dinv; elsewhere: S - F R
			MfmaAcc acc = mfma_zero();
#pragma unroll
			for (int s4 = 0; s4 < GJ_B; s4 += 4)
				acc = mfma_16x16x4(F[16 * ti + (lane & 15)][s4 + (lane >> 4)], B[s4 + (lane >> 4)][16 * tj + (lane & 15)], acc);
			// back to the thread -> element map of the loads / stores through LDS (Apj is free by now)
			__syncthreads();
#pragma unroll
			for (int q = 0; q < 4; q++) Apj[16 * ti + mfma_row(lane, q)][16 * tj + (lane & 15)] = mfma_get(acc, q);
			__syncthreads();
#pragma unroll
			for (int u = 0; u < 4; u++) out[u] = colTile ? -Apj[r][cb + 8 * u] : sv[u] - Apj[r][cb + 8 * u];
		}
#pragma unroll
		for (int u = 0; u < 4; u++)
		{
			const int gi = i0 + r, gj = j0 + cb + 8 * u;
			if (gi < n && gj < n) dst[(size_t)gj * n + gi] = out[u];
		}
		if (aheadRow && j0 == pNext)                        // (uniform over the workgroup)
		{
#pragma unroll
			for (int u = 0; u < 4; u++) keep[u] = out[u];
		}
		if (jt + 1 < jt1)
		{
			__syncthreads();                                   // every reader of Apj / R of this tile is through
#pragma unroll
			for (int u = 0; u < 4; u++)
			{
				sv[u] = svN[u];
				Apj[r][cb + 8 * u] = (r < bk && j0 + GJ_B + cb + 8 * u < n) ? avN[u] : Scalar(0);
			}
			__syncthreads();
		}
	}
	TRACE_MARK();
	// look-ahead: this workgroup produced the next pivot block -> invert it for the next launch
	if (aheadRow && jt0 * GJ_B <= pNext && pNext < jt1 * GJ_B)
	{
		const int bkN = min(GJ_B, n - pNext);
		__syncthreads();                                       // D (the current inverse) is no longer an operand
#pragma unroll
		for (int u = 0; u < 4; u++)
		{
			const int c = cb + 8 * u;
			keep[u] = (r < bkN && c < bkN) ? keep[u] : (r == c ? Scalar(1) : Scalar(0));     // identity padding of a short last block
			D[r][c] = keep[u];
			D2[r][c] = r == c ? Scalar(1) : Scalar(0);
		}
		__syncthreads();
		Scalar (*res)[GJ_B + 1] = gj_pivot_inverse(D, D2, tid, bkN);
#pragma unroll
		for (int u = 0; u < 4; u++) pivOut[(cb + 8 * u) * GJ_B + r] = res[r][cb + 8 * u];
		TRACE_MARK();
		TRACE_FLUSH(2, 8000 + (threadIdx.x >> 6));          // (kept apart: the last launch of a sweep has no look-ahead workgroup)
		return;
	}
	TRACE_MARK();
	TRACE_FLUSH(2, (blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6));
}

// blocked Gauss-Jordan sweep: work0 holds the matrix on entry; returns the buffer (work0 or work1) holding the inverse.
// pivots: 2 x 32 x 32 numbers of scratch (the inverse of the current / of the next pivot block)
Scalar* launch_dense_inverse(Scalar* work0, Scalar* work1, int n, Scalar* pivots, hipStream_t s)
{
	Scalar* src = work0; Scalar* dst = work1;
	const int tiles = (n + GJ_B - 1) / GJ_B;
	// column tiles per workgroup: enough workgroups to fill 256 CUs once (3 fit a CU by their LDS), no second round
	int cols = 1;
	if (const char* e = std::getenv("CUBA_HIP_GJ_COLS")) cols = std::max(1, std::atoi(e));
	else while (tiles * ((tiles + cols - 1) / cols) > 768) cols++;
	const int groups = (tiles + cols - 1) / cols;
	Scalar* pivIn = pivots; Scalar* pivOut = pivots + GJ_B * GJ_B;
	hipLaunchKernelGGL(dense_gj_first_pivot_kernel, dim3(1), dim3(256), 0, s, src, n, min(GJ_B, n), pivIn);
	for (int p0 = 0; p0 < n; p0 += GJ_B)
	{
		hipLaunchKernelGGL(dense_gj_step_kernel, dim3(groups, tiles), dim3(256), 0, s, src, dst, n, p0, min(GJ_B, n - p0), cols, pivIn, pivOut);
		Scalar* tmp = src; src = dst; dst = tmp;
		tmp = pivIn; pivIn = pivOut; pivOut = tmp;
	}
	return src;
}

// Assemble P^T A P from the (already damped) reduced matrix and invert it; returns the buffer (work0 or work1) holding the inverse.
Scalar* launch_coarse_setup(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, Scalar* work0, Scalar* work1, hipStream_t s, hipEvent_t assembled)
{
	const int Nc = 6 * sys.cl * sys.nc;
	(void)hipMemsetAsync(work0, 0, sizeof(Scalar) * (size_t)Nc * Nc, s);
	if (st.nCb) hipLaunchKernelGGL(coarse_assemble_kernel, dim3(st.nCb), dim3(256), 0, s, st, sys, work0, g.Pf);
	if (assembled) (void)hipEventRecord(assembled, s);      // from here on the sweep no longer reads the reduced matrix
	return launch_dense_inverse(work0, work1, Nc, sys.gj_pivots, s);
}

// Fused B(k) of the two-level PCG: [x += alpha p; r -= alpha q;]  rc = P^T r;  z = Minv r + P (Ac^-1 rc);
// rz[kOut] = r.z.  One 512-thread workgroup per aggregate; only the owner of an aggregate stores x, r, z.
// Every workgroup needs the WHOLE restricted residual rc = P^T r_k - alpha P^T q_k (6*nc values). Neither term is
// rebuilt from the full vectors: P^T r_k was stored by the owners one iteration earlier (sys.rc, ping-pong), and
// P^T q_k is summed from the per-workgroup row sums the SpMV kernel leaves in sys.qpart.  Together with the local-k
// slot addressing every global load of the kernel is issued in its first instructions (one memory round trip).
// doUpdate = 0 (once per solve: z_0 = M^-1 r_0) still restricts r directly.
constexpr int PCG2_T = 512;

__device__ __forceinline__ Scalar block_strided_sum(const Scalar* p, int n)
{
	Scalar v0 = 0, v1 = 0;
	int t = threadIdx.x;
	for (; t + PCG2_T < n; t += 2 * PCG2_T)
	{
		const Scalar a = p[t], b = p[t + PCG2_T];
		v0 += a; v1 += b;
	}
	const Scalar e = t < n ? p[t] : Scalar(0);
	return (v0 + e) + v1;
}

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

template <int CL, int AC, int AC2, int W>
__global__ __launch_bounds__(PCG2_T) void pcg2_fused_kernel(DeviceGraph g, DeviceSystem sys, int k, int kOut, int maxIter, Scalar tol2, int doUpdate)
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
	const Scalar* rin = (k & 1) ? sys.r2 : sys.r;
	Scalar* rout = (k & 1) ? sys.r : sys.r2;
	const Scalar* rcin = sys.rc + ((k & 1) ? Nc : 0);
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
		if (t < sys.npq) e_q0 = pq_slot(sys, k)[t];
		if (t + PCG2_T < sys.npq) e_q1 = pq_slot(sys, k)[t + PCG2_T];
	}
	if (t < ownN)
	{
		const size_t gi = (size_t)own0 + t;
		pre_r = rin[gi];
		if (doUpdate) { pre_q = sys.ap[gi]; pre_p = p[gi]; pre_x = sys.xp[gi]; }
		const size_t pose = gi / 6; const int comp = (int)(gi - 6 * pose);
#pragma unroll
		for (int c = 0; c < 6; c++) pre_m[c] = sys.minv[36 * pose + c * 6 + comp];
	}
	if (doUpdate && 2 * t < Nc)
	{
		sr = *reinterpret_cast<const Scalar2*>(rcin + 2 * t);
#pragma unroll
		for (int m = 0; m < QV; m++)      // (m < per is uniform over the grid; sets of workgroups that do not exist stay zero)
			if (m < per) qv[m] = *reinterpret_cast<const Scalar2*>(sys.qpart + (size_t)m * Nc + 2 * t);
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
		for (int u = t + 2 * PCG2_T; u < sys.npq; u += PCG2_T) a_q += pq_slot(sys, k)[u];
	}
	if (t < ownN) { rown[t] = pre_r; qown[t] = pre_q; }
	for (int w = t + PCG2_T; w < ownN; w += PCG2_T)
	{
		rown[w] = rin[own0 + w];
		qown[w] = doUpdate ? sys.ap[own0 + w] : Scalar(0);
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
			for (int m0 = pj == t ? QV : 0; m0 < per; m0 += QV)      // further unknowns of this thread (large graphs): QV loads per trip
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
		if (!(pqk > 0))
		{
			if (blockIdx.x == 0 && threadIdx.x == 0) *sys.fail = 2;
			return;
		}
		alpha = rzk / pqk;
	}
	TRACE_MARK();
	// ---- own rows: r_{k+1}, x_{k+1} ---------------------------------------------------------------------------
	const int ow = t;
	for (int w = ow; w < ownN; w += PCG2_T)
	{
		const Scalar r = rown[w] - alpha * qown[w];      // rown[w] / qown[w] were written by this very thread
		rown[w] = r;
		if (doUpdate)
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
				for (int i = 0; i < W; i++) acc += (Scalar)ainv[a][m][i] * (sR[j + i] - alpha * sQ[j + i]);
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
					for (int i = 0; i < W; i++) acc += (Scalar)ainv2[a][m][i] * (sR[j + i] - alpha * sQ[j + i]);
				}
			}
		}
		if (row < CD)
			for (int j = lane + 64 * W * (AC + AC2); j < Nc; j += 64) acc += (Scalar)acinv[(size_t)(CD * I + row) * ld + j] * (sR[j] - alpha * sQ[j]);
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
	if (t < 8 * CD)
	{
		const int u = t % CD, h = t / CD, a = u / 6, c = u - 6 * a;
		Scalar s3 = 0;
		for (int il = h; 6 * il + c < ownN; il += 8)
			s3 += (a == 0 ? Scalar(1) : agg_weight_local(I, il, sys, g.Pf)) * rown[6 * il + c];
		part[t] = s3;
	}
	__syncthreads();
	if (t >= 64 && t < 64 + CD)
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

static void* pcg2_kernel_for(const DeviceSystem& sys)
{
	const int Nc = 6 * sys.cl * sys.nc;
	if (sys.acinv32 && sizeof(Scalar) == 8)
	{
		// fp32 storage of the coarse inverse: a 16-byte load carries 4 columns, 3 / 6 / 6 + 3 loads per lane and row cover 768 / 1536 / 2304
		if (sys.cl == 2) return Nc <= 768 ? (void*)pcg2_fused_kernel<2, 3, 0, 4> : Nc <= 1536 ? (void*)pcg2_fused_kernel<2, 6, 0, 4> : (void*)pcg2_fused_kernel<2, 6, 3, 4>;
		return Nc <= 768 ? (void*)pcg2_fused_kernel<1, 3, 0, 4> : Nc <= 1536 ? (void*)pcg2_fused_kernel<1, 6, 0, 4> : (void*)pcg2_fused_kernel<1, 6, 3, 4>;
	}
	const bool small = Nc <= 768;
	if (sys.cl == 2) return small ? (void*)pcg2_fused_kernel<2, 6, 0, 2> : (void*)pcg2_fused_kernel<2, 12, 6, 2>;
	return small ? (void*)pcg2_fused_kernel<1, 6, 0, 2> : (void*)pcg2_fused_kernel<1, 12, 6, 2>;
}

static size_t pcg2_lds_bytes(const DeviceSystem& sys)
{
	const size_t cd = 6 * (size_t)sys.cl;
	const size_t ncp = (cd * sys.nc + 3) & ~(size_t)3;
	return sizeof(Scalar) * (2 * ncp + 8 * cd + cd + 32 + 12 * (size_t)sys.agg);
}

// fp64 coarse inverse (n x n, column-major, symmetric up to rounding) -> fp32, symmetrised exactly, rows padded with zeros to ld
__global__ __launch_bounds__(256) void coarse_to_fp32_kernel(const Scalar* __restrict__ src, float* __restrict__ dst, int n, int ld)
{
	const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (t >= (size_t)n * ld) return;
	const int row = (int)(t / ld), j = (int)(t - (size_t)row * ld);
	dst[t] = j < n ? (float)(Scalar(0.5) * (src[(size_t)row * n + j] + src[(size_t)j * n + row])) : 0.0f;
}

void launch_coarse_to_fp32(const Scalar* src, float* dst, int n, hipStream_t s)
{
	const int ld = (n + 3) & ~3;
	const size_t total = (size_t)n * ld;
	if (total) hipLaunchKernelGGL(coarse_to_fp32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, dst, n, ld);
}

void launch_pcg2_fused(const DeviceGraph& g, const DeviceSystem& sys, int k, int kOut, int maxIter, Scalar tol2, int doUpdate, hipStream_t s)
{
	const size_t lds = pcg2_lds_bytes(sys);
	hipLaunchKernelGGL((void (*)(DeviceGraph, DeviceSystem, int, int, int, Scalar, int))pcg2_kernel_for(sys), dim3(sys.nc), dim3(PCG2_T), lds, s, g, sys, k, kOut, maxIter, tol2, doUpdate);
}

// ===================================================================================================
// Single-kernel PCG iteration (round 3).  The two-kernel iteration above pays two dependent launches and two round trips of
// freshly written cross-XCD data per iteration (~15.6 us at KITTI-00 size, 2/3 of a run).  Here ONE launch does a whole iteration:
//   * Chronopoulos-Gear recurrences -- u = M^-1 r, w = A u, p = u + beta p, s = w + beta s, x += alpha p, r -= alpha s, with
//     gamma = r.u and delta = w.u reduced together, alpha_k = gamma_k / (delta_k - beta_k gamma_k / alpha_{k-1}) -- need one global
//     reduction per iteration, and the kernel boundary is that reduction's synchronisation;
//   * a workgroup owns one coarse aggregate.  It recomputes r_{k+1} and u_{k+1} = M^-1 r_{k+1} on its HALO (the poses its block rows
//     touch: r_k, w_k, s_{k-1} of ~50 poses from the previous launches), which needs the coarse correction of the few aggregates the
//     halo poses belong to: those rows of the explicit coarse inverse (fp32) times P^T r_{k+1}, the latter again by recurrence from
//     the restricted vectors P^T r_k, P^T w_k, P^T s_{k-1} that every workgroup left behind (12 numbers each);
//   * then w_{k+1} = A u_{k+1} on its own rows from the row-ordered matrix copy (wave = block row, as pcg_spmv_row_kernel), the
//     two dot products, and its 12 entries of the restricted vectors for the next launch.
// Every address is known at launch (ring slots by k & 3, ping-pong buffers by k & 1): index arrays first, everything else in one
// batch.  The preconditioner M^-1 = blockdiag^-1 + P Ac^-1 P^T is the same fixed SPD operator as in the two-kernel iteration and the
// stop test the same quantity (r.u = r.z), so iteration counts and results agree to rounding.
// ===================================================================================================
constexpr int PCG1_T = 512;

// NQ: 16-byte loads per lane and coarse-inverse row (covers a coarse dimension of 64 W NQ); RB: rows per wave whose loads are in
// flight together
template <int CL, int W, int NQ, int RB>
__global__ __launch_bounds__(PCG1_T) void pcg1_kernel(DeviceGraph g, DeviceStructure st, DeviceSystem sys, int k, int maxIter, Scalar tol2)
{
	constexpr int CD = 6 * CL;
	typedef typename std::conditional<W == 4, float, Scalar>::type PT;
	typedef typename InvVec<PT, W>::type AV;
	extern __shared__ __align__(16) unsigned char pcg1_lds[];
	const int Nc = CD * sys.nc, NcP = (Nc + 3) & ~3, ld = W == 4 ? NcP : Nc;
	const PT* acinv = W == 4 ? reinterpret_cast<const PT*>(sys.acinv32) : reinterpret_cast<const PT*>(sys.acinv);
	Scalar* sC = reinterpret_cast<Scalar*>(pcg1_lds);    // [NcP] P^T r_{k+1}
	Scalar* sY = sC + NcP;                                // [jmax][CD] coarse correction of the halo aggregates
	Scalar* sR = sY + st.jmax * CD;                       // [hmax][6] r_{k+1} on the halo
	Scalar* sU = sR + 6 * st.hmax;                        // [hmax][6] u_{k+1} on the halo
	Scalar* sW = sU + 6 * st.hmax;                        // [agg][6]  w_{k+1} on the own rows
	Scalar* red = sW + 6 * sys.agg;                       // [4][8] per-wave partial sums
	const int t = threadIdx.x, lane = t & 63, wv = t >> 6, I = blockIdx.x;
	const int pk = k & 1;
	const Scalar* rin = pk ? sys.r2 : sys.r;       Scalar* rout = pk ? sys.r : sys.r2;
	const Scalar* win = pk ? sys.w2 : sys.ap;      Scalar* wout = pk ? sys.ap : sys.w2;
	const Scalar* sin_ = pk ? sys.s0 : sys.s1;     Scalar* sout = pk ? sys.s1 : sys.s0;        // s_{k-1} in, s_k out
	const Scalar* cRin = sys.rc + (pk ? Nc : 0);   Scalar* cRout = sys.rc + (pk ? 0 : Nc);
	const Scalar* cWin = sys.cw + (pk ? Nc : 0);   Scalar* cWout = sys.cw + (pk ? 0 : Nc);
	const Scalar* cSin = sys.cs + (pk ? 0 : Nc);   Scalar* cSout = sys.cs + (pk ? Nc : 0);     // P^T s_{k-1} in, P^T s_k out
	const int r0 = I * sys.agg, r1 = min(g.Pf, r0 + sys.agg), nOwn = r1 - r0;
	const int W20 = 20 * st.ell_m;
	TRACE_DECL
	TRACE_MARK();

	// ---- loads: flags, reduction partials, index arrays ------------------------------------------------------------------
	const int kb_v = vector_load_flag(sys.kbase);
	const int failed_v = vector_load_flag(sys.fail) | vector_load_flag(sys.done);
	// (every wave adds up all nc partials of the four sums itself: no LDS exchange, no barrier before alpha and beta are known)
	Scalar e_k[2], e_m[2], e_0[2], e_d[2];
#pragma unroll
	for (int q = 0; q < 2; q++)
	{
		const int u = lane + 64 * q;
		const bool in = u < sys.nc;
		e_k[q] = in ? rz_slot(sys, k)[u] : Scalar(0); e_m[q] = in ? rz_slot(sys, k - 1)[u] : Scalar(0);
		e_0[q] = in ? sys.rz[u] : Scalar(0); e_d[q] = in ? pq_slot(sys, k)[u] : Scalar(0);
	}
	int zero; asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
	const Scalar a_prev_v = sys.alpha[1 - pk + zero];
	const int nJ = st.hal_nj[I];
	// halo vectors: thread x = (halo pose, component), two trips cover 170 poses
	int hp[2]; Scalar hr[2] = { 0, 0 }, hw[2] = { 0, 0 }, hs[2] = { 0, 0 }, hm[2][6]; int hal[2] = { 0, 0 };
#pragma unroll
	for (int q = 0; q < 2; q++)
	{
		// (the lists are padded with -1 up to hmax / jmax: no load waits for the list lengths)
		const int x = t + q * PCG1_T;
		const bool in = x < 6 * st.hmax;
		hp[q] = in ? st.hal_pose[(size_t)I * st.hmax + x / 6] : -1;
		hal[q] = in ? st.hal_aloc[(size_t)I * st.hmax + x / 6] : 0;
	}
	// coarse vectors: thread pairs (2 t, 2 t + 1), three trips cover a coarse dimension of 3072
	Scalar2 vR[3], vW[3], vS[3];
#pragma unroll
	for (int q = 0; q < 3; q++)
	{
		const int j = 2 * (t + q * PCG1_T);
		vR[q] = vW[q] = vS[q] = Scalar2{ 0, 0 };
		if (j < Nc)
		{
			vR[q] = *reinterpret_cast<const Scalar2*>(cRin + j); vW[q] = *reinterpret_cast<const Scalar2*>(cWin + j); vS[q] = *reinterpret_cast<const Scalar2*>(cSin + j);
		}
	}
	// own rows: u_k, p_{k-1}, x_k
	Scalar pre_u = 0, pre_p = 0, pre_x = 0;
	if (t < 6 * nOwn) { const size_t gi = 6 * (size_t)r0 + t; pre_u = sys.z[gi]; pre_p = sys.p0[gi]; pre_x = sys.xp[gi]; }
	// halo vectors (second round trip: their addresses come from the static index arrays above)
#pragma unroll
	for (int q = 0; q < 2; q++)
	{
#pragma unroll
		for (int c = 0; c < 6; c++) hm[q][c] = 0;
		if (hp[q] >= 0)
		{
			const int comp = (t + q * PCG1_T) % 6;
			const size_t gi = 6 * (size_t)hp[q] + comp;
			hr[q] = rin[gi]; hw[q] = win[gi]; hs[q] = sin_[gi];
#pragma unroll
			for (int c = 0; c < 6; c++) hm[q][c] = sys.minv[36 * (size_t)hp[q] + c * 6 + comp];
		}
	}
	// first batch of coarse-inverse rows: wave wv takes rows wv, wv + 8, ... of the nJ * CD rows its halo needs
	const int nRows = nJ * CD;
	AV ainv[RB][NQ];
	int rowOf[RB];
#pragma unroll
	for (int q = 0; q < RB; q++)
	{
		const int ri = wv + 8 * q;
		const int J = ri < st.jmax * CD ? st.hagg_id[(size_t)I * st.jmax + ri / CD] : -1;
		rowOf[q] = J >= 0 ? CD * J + ri % CD : -1;
	}
#pragma unroll
	for (int q = 0; q < RB; q++)
	{
		const PT* Arow = acinv + (size_t)max(rowOf[q], 0) * ld;
#pragma unroll
		for (int m = 0; m < NQ; m++) ainv[q][m] = rowOf[q] >= 0 ? *reinterpret_cast<const AV*>(Arow + min(W * lane + 64 * W * m, ld - W)) : AV(0);
	}
	TRACE_MARK();

	// ---- scalars ---------------------------------------------------------------------------------------------------------------
	Scalar a_k = e_k[0] + e_k[1], a_m = e_m[0] + e_m[1], a_0 = e_0[0] + e_0[1], a_d = e_d[0] + e_d[1];
	for (int u = lane + 128; u < sys.nc; u += 64) { a_k += rz_slot(sys, k)[u]; a_m += rz_slot(sys, k - 1)[u]; a_0 += sys.rz[u]; a_d += pq_slot(sys, k)[u]; }
	const Scalar gk = to_uniform(wave_sum(a_k)), gm = to_uniform(wave_sum(a_m)), g0 = to_uniform(wave_sum(a_0)), dk = to_uniform(wave_sum(a_d));
	const int kabs = k + __builtin_amdgcn_readfirstlane(kb_v);
	const int failed = __builtin_amdgcn_readfirstlane(failed_v);
	if (!(kabs < maxIter && failed == 0 && gk > tol2 * g0 && gk == gk))          // uniform over the grid
	{
		if (blockIdx.x == 0 && threadIdx.x == 0) { *sys.done = 1; if (!(gk == gk)) *sys.fail = 3; }
		return;
	}
	const Scalar a_prev = to_uniform(a_prev_v);
	const Scalar beta = kabs > 0 ? gk / gm : Scalar(0);
	const Scalar den = kabs > 0 ? dk - beta * gk / a_prev : dk;
	if (!(den > 0))
	{
		if (blockIdx.x == 0 && threadIdx.x == 0) *sys.fail = 2;      // not positive definite along the search direction
		return;
	}
	const Scalar alpha = gk / den;

	// ---- restricted vectors: P^T s_k, P^T r_{k+1} ----------------------------------------------------------------------------------
#pragma unroll
	for (int q = 0; q < 3; q++)
	{
		const int j = 2 * (t + q * PCG1_T);
		if (j < Nc)
		{
			const Scalar2 cs = vW[q] + beta * vS[q];
			const Scalar2 cr = vR[q] - alpha * cs;
			*reinterpret_cast<Scalar2*>(sC + j) = cr;
			if (j >= CD * I && j < CD * I + CD)        // (CD is even: a pair never straddles two aggregates)
			{
				*reinterpret_cast<Scalar2*>(cSout + j) = cs; *reinterpret_cast<Scalar2*>(cRout + j) = cr;
			}
		}
	}
	if (t < NcP - Nc) sC[Nc + t] = 0;
	__syncthreads();
	TRACE_MARK();

	// ---- coarse correction of the halo aggregates: yc = Ac^-1[rows] (P^T r_{k+1}) -----------------------------------------------------
	for (int b = 0; 8 * RB * b < nRows; b++)
	{
		if (b > 0)
		{
#pragma unroll
			for (int q = 0; q < RB; q++)
			{
				const int ri = wv + 8 * (RB * b + q);
				const int J = ri < nRows ? st.hagg_id[(size_t)I * st.jmax + ri / CD] : -1;
				rowOf[q] = J >= 0 ? CD * J + ri % CD : -1;
			}
#pragma unroll
			for (int q = 0; q < RB; q++)
			{
				const PT* Arow = acinv + (size_t)max(rowOf[q], 0) * ld;
#pragma unroll
				for (int m = 0; m < NQ; m++) ainv[q][m] = rowOf[q] >= 0 ? *reinterpret_cast<const AV*>(Arow + min(W * lane + 64 * W * m, ld - W)) : AV(0);
			}
		}
#pragma unroll
		for (int q = 0; q < RB; q++)
		{
			Scalar acc = 0;
#pragma unroll
			for (int m = 0; m < NQ; m++)
			{
				const int j = W * lane + 64 * W * m;
				if (j < Nc)
				{
#pragma unroll
					for (int i = 0; i < W; i++) acc += (Scalar)ainv[q][m][i] * sC[j + i];
				}
			}
			if (rowOf[q] >= 0)
				for (int j = lane + 64 * W * NQ; j < Nc; j += 64) acc += (Scalar)acinv[(size_t)rowOf[q] * ld + j] * sC[j];     // (beyond the instantiated width: never in practice)
			acc = wave_sum(acc);
			const int ri = wv + 8 * (RB * b + q);
			if (lane == 0 && ri < nRows) sY[ri] = acc;            // row ri = (local aggregate ri / CD, coarse unknown ri % CD)
		}
	}

	TRACE_MARK();
	// ---- matrix entries of the first SpMV round: requested now, needed after the two halo phases ------------------------------------------
	const int slot = lane / 3, r2 = lane - 3 * slot;
	Scalar2 aNext[3][6];
	int locNext[3] = { -1, -1, -1 };
	auto loadRound = [&](int row) {
#pragma unroll
		for (int m = 0; m < 3; m++)
		{
			const bool in = lane < 60 && m < st.ell_m;
			locNext[m] = in ? st.ell_loc[(size_t)row * W20 + m * 20 + slot] : -1;
			const Scalar2* A2 = reinterpret_cast<const Scalar2*>(sys.hrow + 36 * ((size_t)row * W20 + (in ? m * 20 + slot : 0)) + 12 * r2);
#pragma unroll
			for (int c = 0; c < 6; c++) aNext[m][c] = in ? A2[c] : Scalar2{ 0, 0 };
		}
	};
	if (wv < nOwn) loadRound(r0 + wv);

	// ---- halo: s_k, r_{k+1} ----------------------------------------------------------------------------------------------------------
#pragma unroll
	for (int q = 0; q < 2; q++)
	{
		if (hp[q] >= 0)
		{
			const int x = t + q * PCG1_T;
			const Scalar sk = hw[q] + beta * hs[q];
			const Scalar rn = hr[q] - alpha * sk;
			sR[x] = rn;
			if (hp[q] >= r0 && hp[q] < r1) { const size_t gi = 6 * (size_t)hp[q] + x % 6; sout[gi] = sk; rout[gi] = rn; }
		}
	}
	__syncthreads();
	// ---- halo: u_{k+1} = blockdiag^-1 r_{k+1} + P yc -------------------------------------------------------------------------------------
#pragma unroll
	for (int q = 0; q < 2; q++)
	{
		if (hp[q] >= 0)
		{
			const int x = t + q * PCG1_T, hl = x / 6, comp = x - 6 * hl;
			Scalar u = sY[hal[q] * CD + comp];
			if (CL == 2) u += agg_weight(hp[q], sys.agg, g.Pf) * sY[hal[q] * CD + 6 + comp];
#pragma unroll
			for (int c = 0; c < 6; c++) u += hm[q][c] * sR[6 * hl + c];
			sU[x] = u;
		}
	}
	__syncthreads();
	TRACE_MARK();

	// ---- w_{k+1} = A u_{k+1} on the own rows: wave = block row, lane = (slot, pair of block rows) --------------------------------------
	// (the matrix entries of a round -- one row per wave -- were requested a round ahead: padding slots of the row-ordered copy hold
	// zeros, so the loads need no index and no mask)
	{
		for (int il = wv; il < nOwn; il += PCG1_T / 64)
		{
			const int row = r0 + il;
			Scalar az0 = 0, az1 = 0;
			Scalar2 a0v[3][3], a1v[3][3];
			int loc[3];
#pragma unroll
			for (int m = 0; m < 3; m++)
			{
				loc[m] = locNext[m];
#pragma unroll
				for (int c = 0; c < 3; c++) { a0v[m][c] = aNext[m][c]; a1v[m][c] = aNext[m][3 + c]; }
			}
			if (il + PCG1_T / 64 < nOwn) loadRound(row + PCG1_T / 64);       // next round's entries fly while this round is summed
#pragma unroll
			for (int m = 0; m < 3; m++)
			{
				if (loc[m] >= 0)
				{
					const Scalar2* u2 = reinterpret_cast<const Scalar2*>(sU + 6 * loc[m]);
#pragma unroll
					for (int c = 0; c < 3; c++)
					{
						const Scalar2 uv = u2[c];
						az0 += a0v[m][c].x * uv.x + a0v[m][c].y * uv.y;
						az1 += a1v[m][c].x * uv.x + a1v[m][c].y * uv.y;
					}
				}
			}
			// fold the 20 slots (lanes 3 apart) onto lanes 0..2, then spread the six block rows over lanes 0..5
			az0 += __shfl_down(az0, 30); az1 += __shfl_down(az1, 30);
			az0 += __shfl_down(az0, 15); az1 += __shfl_down(az1, 15);
			Scalar t0 = az0, t1 = az1;
#pragma unroll
			for (int d = 3; d <= 12; d += 3) { t0 += __shfl_down(az0, d); t1 += __shfl_down(az1, d); }
			const Scalar wA = __shfl(t0, lane >> 1), wB = __shfl(t1, lane >> 1);
			if (lane < 6)
			{
				const Scalar wn = (lane & 1) ? wB : wA;
				sW[6 * il + lane] = wn;
				wout[6 * (size_t)row + lane] = wn;
			}
		}
	}
	__syncthreads();
	TRACE_MARK();

	// ---- own rows: p_k, x_{k+1}, u_{k+1}; gamma_{k+1}, delta_{k+1}; P^T w_{k+1} ---------------------------------------------------------
	Scalar dg = 0, dd = 0;
	if (t < 6 * nOwn)
	{
		const int il = t / 6, comp = t - 6 * il;
		const size_t gi = 6 * (size_t)r0 + t;
		const int lo = st.own_loc[r0 + il];
		const Scalar rn = sR[6 * lo + comp], un = sU[6 * lo + comp], wn = sW[t];
		const Scalar pkv = pre_u + beta * pre_p;
		sys.p0[gi] = pkv;
		sys.xp[gi] = pre_x + alpha * pkv;
		sys.z[gi] = un;
		dg = rn * un; dd = wn * un;
	}
	dg = wave_sum(dg); dd = wave_sum(dd);
	__syncthreads();                       // (red is reused)
	if (lane == 0) { red[wv] = dg; red[8 + wv] = dd; }
	if (t >= 64 && t < 64 + CD)
	{
		const int u = t - 64, a = u / 6, c = u - 6 * a;
		Scalar s3 = 0;
		for (int il = 0; il < nOwn; il++) s3 += (a == 0 ? Scalar(1) : agg_weight_local(I, il, sys, g.Pf)) * sW[6 * il + c];
		cWout[CD * I + u] = s3;
	}
	__syncthreads();
	if (t == 0)
	{
		Scalar sg = 0, sd = 0;
#pragma unroll
		for (int w = 0; w < PCG1_T / 64; w++) { sg += red[w]; sd += red[8 + w]; }
		rz_slot_w(sys, k + 1)[I] = sg;
		pq_slot(sys, k + 1)[I] = sd;
		if (I == 0) { sys.alpha[pk] = alpha; *sys.iters = kabs + 1; }
	}
	TRACE_MARK();
	TRACE_FLUSH(1, blockIdx.x * (PCG1_T / 64) + wv);
}

// per-solve start of the single-kernel iteration, after pcg_setup (r_0, blockdiag^-1, x = p = 0), the fused kernel with doUpdate = 0
// (u_0 = M^-1 r_0 -> z, P^T r_0 -> rc, gamma_0) and the SpMV with k = 0 (w_0 = A u_0 -> ap, its dot-product partials, the row sums of
// w_0 in qpart): delta_0 as nc partials, P^T w_0, and zeros for s_{-1}, P^T s_{-1}
__global__ __launch_bounds__(256) void pcg1_init_kernel(DeviceGraph g, DeviceSystem sys)
{
	const int Nc = 6 * sys.cl * sys.nc;
	const int gt = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
	const int per = sys.agg / sys.spmv_rows;
	for (int j = gt; j < Nc; j += stride)
	{
		Scalar s2 = 0;
		for (int m = 0; m < per; m++) s2 += sys.qpart[(size_t)m * Nc + j];
		sys.cw[j] = s2;               // P^T w_0 (k = 0 reads the first halves)
		sys.cs[Nc + j] = 0;           // P^T s_{-1}
	}
	for (int i = gt; i < 6 * g.Pf; i += stride) sys.s1[i] = 0;      // s_{-1}
	if (blockIdx.x == 0)
	{
		__shared__ Scalar sh[4];
		Scalar v = 0;
		for (int u = threadIdx.x; u < sys.npq; u += 256) v += pq_slot(sys, 0)[u];
		v = wave_sum(v);
		if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
		__syncthreads();
		const Scalar tot = (sh[0] + sh[1]) + (sh[2] + sh[3]);
		__syncthreads();
		for (int u = threadIdx.x; u < sys.nc; u += 256) pq_slot(sys, 0)[u] = u == 0 ? tot : Scalar(0);
		if (threadIdx.x == 0) { sys.alpha[0] = 0; sys.alpha[1] = 0; }
	}
}

struct Pcg1Variant { void* fn; int width; };       // width = coarse dimension the instantiation covers
static Pcg1Variant pcg1_kernel_for(const DeviceSystem& sys)
{
	const int Nc = 6 * sys.cl * sys.nc;
	const bool f32 = sys.acinv32 && sizeof(Scalar) == 8;
	if (sys.cl == 2)
	{
		if (f32) return Nc <= 1024 ? Pcg1Variant{ (void*)pcg1_kernel<2, 4, 4, 6>, 1024 } : Nc <= 1536 ? Pcg1Variant{ (void*)pcg1_kernel<2, 4, 6, 4>, 1536 } : Pcg1Variant{ (void*)pcg1_kernel<2, 4, 9, 3>, 2304 };
		return Nc <= 768 ? Pcg1Variant{ (void*)pcg1_kernel<2, 2, 6, 4>, 768 } : Pcg1Variant{ (void*)pcg1_kernel<2, 2, 12, 2>, 1536 };
	}
	if (f32) return Nc <= 1024 ? Pcg1Variant{ (void*)pcg1_kernel<1, 4, 4, 6>, 1024 } : Pcg1Variant{ (void*)pcg1_kernel<1, 4, 9, 3>, 2304 };
	return Nc <= 768 ? Pcg1Variant{ (void*)pcg1_kernel<1, 2, 6, 4>, 768 } : Pcg1Variant{ (void*)pcg1_kernel<1, 2, 12, 2>, 1536 };
}

static size_t pcg1_lds_bytes(const DeviceStructure& st, const DeviceSystem& sys)
{
	const size_t cd = 6 * (size_t)sys.cl;
	const size_t ncp = (cd * sys.nc + 3) & ~(size_t)3;
	return sizeof(Scalar) * (ncp + (size_t)st.jmax * cd + 12 * (size_t)st.hmax + 6 * (size_t)sys.agg + 32);
}

bool pcg1_supported(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys)
{
	if (sys.agg <= 0 || sys.nc <= 0 || st.ell_over || st.ell_m < 1 || st.ell_m > 3 || !st.hal_n) return false;
	const int Nc = 6 * sys.cl * sys.nc;
	if (6 * sys.agg > PCG1_T || 6 * st.hmax > 2 * PCG1_T || Nc > 6 * PCG1_T || sys.nc > sys.rzStride || sys.nc > sys.pqStride) return false;
	if (Nc > pcg1_kernel_for(sys).width) return false;
	return pcg1_lds_bytes(st, sys) <= 64 * 1024;
}

void launch_pcg1_init(const DeviceGraph& g, const DeviceSystem& sys, hipStream_t s)
{
	const int n = max(6 * sys.cl * sys.nc, 6 * g.Pf);
	hipLaunchKernelGGL(pcg1_init_kernel, dim3(min(256, (n + 255) / 256)), dim3(256), 0, s, g, sys);
}

void launch_pcg1(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, int k, int maxIter, Scalar tol2, hipStream_t s)
{
	hipLaunchKernelGGL((void (*)(DeviceGraph, DeviceStructure, DeviceSystem, int, int, Scalar))pcg1_kernel_for(sys).fn, dim3(sys.nc), dim3(PCG1_T),
		pcg1_lds_bytes(st, sys), s, g, st, sys, k, maxIter, tol2);
}

static bool spmv_wants_occupancy(const DeviceGraph& g) { return 2 * (long long)g.Pf > 3 * 1024; }   // two waves per row vs 1024 SIMDs x occupancy 3
int spmv_rows_for(int Pf)
{
	if (const char* e = std::getenv("CUBA_HIP_SPMV_ROWS")) { const int r = std::atoi(e); return r == 8 ? 8 : r == 4 ? 4 : 2; }     // A/B knob
	return 2 * (long long)Pf > 3 * 1024 ? 4 : 2;
}                            // (the 4-row workgroup needs the 128-VGPR instantiation)
static bool spmv_row_per_wave(const DeviceSystem& sys)
{
	static const bool off = std::getenv("CUBA_HIP_SPMV_TWO_WAVES") != nullptr;     // A/B knob
	return sys.spmv_rows >= 4 && !off;
}
static void* spmv_kernel_for(const DeviceGraph& g, const DeviceSystem& sys)
{
	if (sys.spmv_rows == 8) return spmv_row_per_wave(sys) ? (void*)pcg_spmv_row_kernel<8> : (void*)pcg_spmv_kernel<8, 1>;
	if (sys.spmv_rows == 4) return spmv_row_per_wave(sys) ? (void*)pcg_spmv_row_kernel<4> : (void*)pcg_spmv_kernel<4, 4>;
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
__global__ __launch_bounds__(64) void pcg_advance_kernel(DeviceSystem sys, int n, int report, Scalar tol2)
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
	if (report && sys.host_flags)
	{
		sys.host_flags[0] = *sys.fail; sys.host_flags[1] = *sys.iters; sys.host_flags[2] = *sys.done;
		__threadfence_system();
		sys.host_flags[3] = ++(*sys.ticket);      // the host spins on this word instead of paying a stream-synchronise round trip
	}
}

// Forcing term of the inexact Levenberg-Marquardt step (option "pcg_forcing"): one wave between the first preconditioner application
// of a solve and its first iteration.  The stop test of every iteration kernel is r_k.z_k <= tol^2 * sum(slot 0 of rz), slot 0 holding
// the per-workgroup partials of r_0.z_0; this kernel fills ONE extra partial behind them with max(0, eta^2 * ref - r_0.z_0), ref =
// r_0.z_0 of the first solve of the LM run (stored here when isFirst != 0) -- so the test becomes r_k.z_k <= tol^2 max(r_0.z_0, eta^2 ref)
// without a single extra load in the iteration kernels.  Fixed summation order: reproducible.
__global__ __launch_bounds__(64) void pcg_forcing_kernel(DeviceSystem sys, int nReal, Scalar eta2, int isFirst, Scalar* ref)
{
	const int lane = threadIdx.x;
	const Scalar rz0 = wave_sum(load_parts(sys.rz, nReal, lane));
	if (lane != 0) return;
	if (isFirst) { *ref = rz0; sys.rz[nReal] = 0; return; }
	const Scalar want = eta2 * *ref;
	sys.rz[nReal] = want > rz0 ? want - rz0 : Scalar(0);
}

void launch_pcg_forcing(const DeviceSystem& sys, Scalar eta2, int isFirst, Scalar* ref, hipStream_t s)
{
	hipLaunchKernelGGL(pcg_forcing_kernel, dim3(1), dim3(64), 0, s, sys, sys.nrz0 - 1, eta2, isFirst, ref);
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

void launch_pcg_advance(const DeviceSystem& sys, int n, hipStream_t s)
{
	hipLaunchKernelGGL(pcg_advance_kernel, dim3(1), dim3(64), 0, s, sys, n, 1, Scalar(-1));
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
	for (int k = 0; k < chunk && e == hipSuccess && sys.cg1; k++)
		e = add_kernel_node(graph, last, pcg1_kernel_for(sys).fn, dim3(sys.nc), dim3(PCG1_T), (unsigned)pcg1_lds_bytes(st, sys), g, st, sys, k, maxIter, tol2);
	for (int k = 0; k < chunk && e == hipSuccess && !sys.cg1; k++)
	{
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
