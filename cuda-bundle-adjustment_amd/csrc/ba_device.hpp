// ba_device.hpp -- device-side helpers shared by the kernel translation units (ba_edge.hip, ba_linearize.hip, ba_pcg.hip,
// ba_coarse.hip): DPP wave reductions, flag loads on the vector path, the deterministic second-stage sums, loading +
// linearising one edge.  Internal to csrc/ -- the launch interface is ba_kernels.hpp.
#pragma once

#include "ba_kernels.hpp"

#include <type_traits>

namespace cubahip
{


// Stage timestamps for latency studies (scripts/trace_pcg.py): only in the separate libcuba_hip_trace.so build.
#ifdef CUBA_HIP_TRACE
static __device__ unsigned long long cuba_trace_buf[3][8192 * 8];
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

// the damping of a launch: the kernel argument, or -- when that is negative -- what the device-resident LM decision left in sys.lam_dev
__device__ __forceinline__ Scalar launch_lambda(const DeviceSystem& sys, Scalar lambda)
{
	return lambda < Scalar(0) ? sys.lam_dev[0] : lambda;
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

static __global__ __launch_bounds__(1024) void reduce_parts_kernel(const Scalar* __restrict__ parts, int n, Scalar* out)
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

// The report a host spins on (cuba_hip_solver::waitReport): {failure code, iterations done, stop flag} and, LAST, the ticket -- all in the
// handle's coherent mapped host block; results the same thread stored before (the LM decision record, the sum slots) must have LANDED before
// the ticket does.  A system-scope release fence alone is not enough here: on this compiler the wait for the outstanding stores behind the
// fence's write-back can be dropped when the wave's store counter is provably empty at that point (microarchitecture guide, "compiler
// hazard"), and a store to another address may then reach the host after the ticket -- seen as an iteration's chi2 reported 0 or stale once
// in a few hundred short runs.  The explicit wait is invisible to that pass.
__device__ __forceinline__ void publish_report(const DeviceSystem& sys)
{
	sys.host_flags[0] = *sys.fail; sys.host_flags[1] = *sys.iters; sys.host_flags[2] = *sys.done;
	__threadfence_system();
	asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
	const int t = ++(*sys.ticket);
	__hip_atomic_store(&sys.host_flags[3], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace cubahip
