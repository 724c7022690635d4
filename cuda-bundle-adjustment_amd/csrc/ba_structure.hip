// ba_structure.hip -- kernels of the device-side set-up (see ba_structure.hpp).
#include "ba_structure.hpp"

#include <cstring>

#include <rocprim/rocprim.hpp>

#include "ba_kernels.hpp"

namespace cubahip
{
namespace topo
{

// ---------------------------------------------------------------------------------------------------------------------
// rocPRIM wrappers
// ---------------------------------------------------------------------------------------------------------------------
size_t sort_temp_bytes(size_t n)
{
	size_t b = 0;
	(void)rocprim::radix_sort_pairs(nullptr, b, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint64_t*)nullptr, (uint64_t*)nullptr, n, 0, 64, (hipStream_t)0);
	return b + 256;
}

size_t scan_temp_bytes(size_t n)
{
	size_t b = 0;
	(void)rocprim::exclusive_scan(nullptr, b, (const long long*)nullptr, (long long*)nullptr, 0LL, n, rocprim::plus<long long>(), (hipStream_t)0);
	return b + 256;
}

hipError_t sort_u64_u32(void* temp, size_t tb, const uint64_t* kin, uint64_t* kout, const uint32_t* vin, uint32_t* vout, size_t n, int endBit, hipStream_t s)
{
	return rocprim::radix_sort_pairs(temp, tb, kin, kout, vin, vout, n, 0, (unsigned)endBit, s);
}

hipError_t sort_u64_u64(void* temp, size_t tb, const uint64_t* kin, uint64_t* kout, const uint64_t* vin, uint64_t* vout, size_t n, int endBit, hipStream_t s)
{
	return rocprim::radix_sort_pairs(temp, tb, kin, kout, vin, vout, n, 0, (unsigned)endBit, s);
}

hipError_t sort_u32_u32(void* temp, size_t tb, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, size_t n, int endBit, hipStream_t s)
{
	return rocprim::radix_sort_pairs(temp, tb, kin, kout, vin, vout, n, 0, (unsigned)endBit, s);
}

hipError_t exclusive_scan_i64(void* temp, size_t tb, const long long* in, long long* out, size_t n, hipStream_t s)
{
	return rocprim::exclusive_scan(temp, tb, in, out, 0LL, n, rocprim::plus<long long>(), s);
}

hipError_t inclusive_scan_i32(void* temp, size_t tb, const int* in, int* out, size_t n, hipStream_t s)
{
	return rocprim::inclusive_scan(temp, tb, in, out, n, rocprim::plus<int>(), s);
}

namespace
{
constexpr int T = 256;
inline dim3 grid_for(size_t n) { return dim3((unsigned)((n + T - 1) / T)); }
constexpr uint64_t KEY_MAX = ~0ull;

// ---------------------------------------------------------------------------------------------------------------------
// A. edges
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(T) void edge_keys_kernel(const int* ep, const int* el, const uint8_t* dim, int E, int Pt, int Pf, int Lt, int Lf,
	uint64_t* keys, uint32_t* vals, int* counters)
{
	const int e = blockIdx.x * T + threadIdx.x;
	if (e >= E) return;
	const int p = ep[e], l = el[e], d = dim[e];
	int bad = 0;
	if (p < 0 || p >= Pt || l < 0 || l >= Lt) bad = 1;
	else if (d != 2 && d != 3) bad = 2;
	else if (p >= Pf && l >= Lf) bad = 3;
	if (bad) counters[CNT_BAD] = bad;
	keys[e] = bad == 1 ? 0ull : (((uint64_t)(uint32_t)l << 32) | (uint32_t)p);
	vals[e] = (uint32_t)e;
}

__global__ __launch_bounds__(T) void gather_edges_kernel(const uint32_t* perm, const int* ep, const int* el, const uint8_t* dim, const double* meas,
	const double* omega, int E, int* e_pose, int* e_lm, Scalar* mu, Scalar* mv, Scalar* mr, Scalar* w)
{
	const int i = blockIdx.x * T + threadIdx.x;
	if (i >= E) return;
	const size_t e = perm[i];
	const bool stereo = dim[e] == 3;
	if (e_pose) { e_pose[i] = ep[e] | (stereo ? STEREO_BIT : 0); e_lm[i] = el[e]; }
	if (!mu) return;                   // (index arrays only: the values are still on their way from the host)
	mu[i] = (Scalar)meas[3 * e]; mv[i] = (Scalar)meas[3 * e + 1];
	mr[i] = stereo ? (Scalar)meas[3 * e + 2] : Scalar(0);
	w[i] = (Scalar)omega[e];
}

__global__ __launch_bounds__(T) void scatter_values_kernel(const int* ids, const double* packed, int n, double* meas, double* omega)
{
	const int i = blockIdx.x * T + threadIdx.x;
	if (i >= n) return;
	const size_t e = ids[i];
	meas[3 * e] = packed[4 * (size_t)i]; meas[3 * e + 1] = packed[4 * (size_t)i + 1]; meas[3 * e + 2] = packed[4 * (size_t)i + 2];
	omega[e] = packed[4 * (size_t)i + 3];
}

__global__ __launch_bounds__(T) void segment_ptr_kernel(const int* keys, int n, int nSeg, int* ptr)
{
	const int i = blockIdx.x * T + threadIdx.x;
	if (i > n) return;
	// position i closes the segments (prev, cur]: prev = key before i (-1 at the start), cur = key at i (nSeg at the end)
	const int prev = i > 0 ? min(keys[i - 1], nSeg) : -1;
	const int cur = i < n ? min(keys[i], nSeg) : nSeg;
	for (int k = prev + 1; k <= cur; k++) ptr[k] = i;
}

// ---------------------------------------------------------------------------------------------------------------------
// B. structure
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(T) void lm_pairs_kernel(const int* lm_ptr, const int* e_pose, int Lf, int Pf, int* nfree, long long* pairCount, long long* freeCount)
{
	const int l = blockIdx.x * T + threadIdx.x;
	if (l > Lf) return;
	if (l == Lf) { pairCount[l] = 0; freeCount[l] = 0; return; }
	int n = 0;
	for (int i = lm_ptr[l]; i < lm_ptr[l + 1]; i++) n += (e_pose[i] & ~STEREO_BIT) < Pf;      // sorted by pose: the free ones come first
	nfree[l] = n;
	pairCount[l] = (long long)n * (n - 1) / 2;
	freeCount[l] = n;
}

__global__ __launch_bounds__(T) void pose_keys_kernel(const int* e_pose, int E, int Pf, uint32_t* keys, uint32_t* vals)
{
	const int i = blockIdx.x * T + threadIdx.x;
	if (i >= E) return;
	const int p = e_pose[i] & ~STEREO_BIT;
	keys[i] = (uint32_t)min(p, Pf);
	vals[i] = (uint32_t)i;
}

__global__ __launch_bounds__(T) void copy_u32_int_kernel(const uint32_t* in, int* out, int n)
{
	const int i = blockIdx.x * T + threadIdx.x;
	if (i < n) out[i] = (int)in[i];
}

__global__ __launch_bounds__(T) void pattern_entries_kernel(const int* lm_ptr, const int* e_pose, const int* e_lm, const int* nfree, const long long* pairBase,
	int E, int Lf, int Pf, uint64_t* keys, uint64_t* vals)
{
	const int i = blockIdx.x * T + threadIdx.x;
	if (i < Pf) { keys[i] = ((uint64_t)i << 32) | (uint32_t)i; vals[i] = 0; }       // diagonal seeds: every free pose owns its diagonal block
	if (i >= E) return;
	const int l = e_lm[i];
	if (l >= Lf) return;
	const int b0 = lm_ptr[l], a = i - b0, n = nfree[l];
	if (a >= n) return;                                                            // edge of a fixed pose
	const uint64_t pa = (uint64_t)(uint32_t)(e_pose[i] & ~STEREO_BIT);
	size_t out = (size_t)Pf + (size_t)pairBase[l] + (size_t)a * (n - 1) - (size_t)a * (a - 1) / 2;     // id of product (a, a + 1)
	for (int c = a + 1; c < n; c++, out++)
	{
		keys[out] = (pa << 32) | (uint32_t)(e_pose[b0 + c] & ~STEREO_BIT);
		vals[out] = ((uint64_t)(uint32_t)(i + 1) << 32) | (uint32_t)(b0 + c + 1);
	}
}

__global__ __launch_bounds__(T) void entry_heads_kernel(const uint64_t* keys, size_t n, int* head)
{
	const size_t j = (size_t)blockIdx.x * T + threadIdx.x;
	if (j < n) head[j] = (j == 0 || keys[j] != keys[j - 1]) ? 1 : 0;
}

__global__ __launch_bounds__(T) void blocks_from_entries_kernel(const uint64_t* keys, const uint64_t* vals, const int* blkOfEntry, size_t n, int Pf,
	int* colind, int* blkrow, int* prod_ptr, int* prod_ea, int* prod_eb)
{
	const size_t j = (size_t)blockIdx.x * T + threadIdx.x;
	if (j >= n) return;
	const uint64_t key = keys[j], val = vals[j];
	const int row = (int)(key >> 32), col = (int)(uint32_t)key;
	const int b = blkOfEntry[j] - 1;
	const bool head = j == 0 || keys[j - 1] != key;
	const bool seed = val == 0;
	// seeds: one per row, first entry of its row (stable sort, seeds first in the input) => row + 1 seeds up to and
	// including a non-seed entry of that row, row seeds before the seed of that row
	const long long ppos = (long long)j - row - 1;
	if (head)
	{
		colind[b] = col; blkrow[b] = row;
		prod_ptr[b] = (int)(seed ? ppos + 1 : ppos);
	}
	if (!seed)
	{
		prod_ea[ppos] = (int)(val >> 32) - 1;
		prod_eb[ppos] = (int)(uint32_t)val - 1;
	}
	if (j == n - 1) prod_ptr[b + 1] = (int)(n - (size_t)Pf);
}

// For every segment s of a list (items ptr[s] .. ptr[s + 1], their values ascending): the sub-range whose values lie in [vlo, vhi).
// A landmark partition keeps the global lists and walks these sub-ranges: the products of a block are in landmark order, a pose's
// edges in edge (= landmark) order.
__global__ __launch_bounds__(T) void segment_subrange_kernel(const int* __restrict__ ptr, int nseg, const int* __restrict__ vals, int vlo, int vhi,
	int* __restrict__ beg, int* __restrict__ end)
{
	const int sgm = blockIdx.x * T + threadIdx.x;
	if (sgm >= nseg) return;
	const int a0 = ptr[sgm], a1 = ptr[sgm + 1];
	int lo = a0, hi = a1;
	while (lo < hi) { const int m = (lo + hi) >> 1; if (vals[m] < vlo) lo = m + 1; else hi = m; }
	const int b = lo;
	hi = a1;
	while (lo < hi) { const int m = (lo + hi) >> 1; if (vals[m] < vhi) lo = m + 1; else hi = m; }
	beg[sgm] = b; end[sgm] = lo;
}

__global__ __launch_bounds__(T) void od_keys_kernel(const int* prod_beg, const int* prod_end, const int* blkrow, const int* colind, int nblk, int farOffset, int heavy, uint32_t* keys, uint32_t* vals, int* counters)
{
	const int k = blockIdx.x * T + threadIdx.x;
	int cnt = 0, far = 0, dup = 0;
	if (k < nblk)
	{
		far = colind[k] - blkrow[k] > farOffset;
		cnt = prod_end[k] - prod_beg[k];
		dup = cnt > 0 && colind[k] == blkrow[k];
		keys[k] = cnt > 0 ? 0x7fffffffu - (uint32_t)cnt : 0xffffffffu;          // longest list first; blocks without products last
		vals[k] = (uint32_t)k;
	}
	const int n = __popcll(__ballot(cnt > 0)), nf = __popcll(__ballot(far != 0));
	if ((threadIdx.x & 63) == 0 && n) atomicAdd(&counters[CNT_NOD], n);
	if ((threadIdx.x & 63) == 0 && nf) atomicAdd(&counters[CNT_FARBLOCKS], nf);
	const int nd = __popcll(__ballot(dup != 0));
	if ((threadIdx.x & 63) == 0 && nd) atomicAdd(&counters[CNT_DIAGPROD], nd);
	const int nh = __popcll(__ballot(cnt > heavy));
	if ((threadIdx.x & 63) == 0 && nh) atomicAdd(&counters[CNT_NHEAVY], nh);
}

__global__ __launch_bounds__(T) void remap_poses_kernel(const int* epIn, const int* newOfOld, int E, int Pf, int* epOut)
{
	const int e = blockIdx.x * T + threadIdx.x;
	if (e >= E) return;
	const int p = epIn[e];
	epOut[e] = p < Pf ? newOfOld[p] : p;
}

__global__ __launch_bounds__(T) void transpose_keys_kernel(const int* colind, const int* blkrow, int nblk, uint64_t* keys, uint32_t* vals)
{
	const int k = blockIdx.x * T + threadIdx.x;
	if (k >= nblk) return;
	const int r = blkrow[k], c = colind[k];
	keys[k] = r == c ? KEY_MAX : (((uint64_t)(uint32_t)c << 32) | (uint32_t)r);
	vals[k] = (uint32_t)k;
}

__global__ __launch_bounds__(T) void keys_hi_kernel(const uint64_t* keys, int n, int limit, int* hi)
{
	const int i = blockIdx.x * T + threadIdx.x;
	if (i < n) hi[i] = (int)min((uint64_t)limit, keys[i] >> 32);
}

__global__ __launch_bounds__(T) void adj_ptr_kernel(const int* rowptr, const int* lowerPtr, int Pf, int* adjPtr, int* counters)
{
	const int i = blockIdx.x * T + threadIdx.x;
	int len = 0;
	if (i <= Pf)
	{
		adjPtr[i] = lowerPtr[i] + rowptr[i];
		if (i < Pf) len = (lowerPtr[i + 1] + rowptr[i + 1]) - (lowerPtr[i] + rowptr[i]);
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) len = max(len, __shfl_xor(len, o));
	if ((threadIdx.x & 63) == 0 && len) atomicMax(&counters[CNT_MAXROW], len);
}

__global__ __launch_bounds__(T) void adj_fill_kernel(const int* rowptr, const int* colind, const int* blkrow, const int* lowerPtr, const uint64_t* tKeys,
	const uint32_t* tBlk, int nblk, const int* adjPtr, int* adjBlk, int* adjCol, int* adjRow)
{
	const int t = blockIdx.x * T + threadIdx.x;
	if (t >= nblk) return;
	// lower part of column c: the off-diagonal blocks (r, c), r < c, in ascending r = the sorted transposed order
	const uint64_t key = tKeys[t];
	if (key != KEY_MAX)
	{
		const int c = (int)(key >> 32), r = (int)(uint32_t)key;
		const int a = adjPtr[c] + (t - lowerPtr[c]);
		adjBlk[a] = (int)tBlk[t] | (int)0x80000000; adjCol[a] = r; adjRow[a] = c;
	}
	// the row's own blocks after its lower part
	const int i = blkrow[t];
	const int a = adjPtr[i] + (lowerPtr[i + 1] - lowerPtr[i]) + (t - rowptr[i]);
	adjBlk[a] = t; adjCol[a] = colind[t]; adjRow[a] = i;
}

__global__ __launch_bounds__(T) void ell_kernel(const int* adjPtr, const int* adjBlk, const int* adjCol, int Pf, int M, int2* ell)
{
	const size_t x = (size_t)blockIdx.x * T + threadIdx.x;
	const int W = 20 * M;
	if (x >= (size_t)Pf * W) return;
	const int i = (int)(x / W), e = (int)(x - (size_t)i * W);
	const int a0 = adjPtr[i], n = adjPtr[i + 1] - a0;
	ell[x] = e < n ? int2{ adjBlk[a0 + e], adjCol[a0 + e] } : int2{ 0, -1 };
}

__global__ __launch_bounds__(T) void coarse_keys_kernel(const int* adjRow, const int* adjCol, int nAdj, int agg, int nc, uint32_t* keys, uint32_t* vals)
{
	const int a = blockIdx.x * T + threadIdx.x;
	if (a >= nAdj) return;
	keys[a] = (uint32_t)((adjRow[a] / agg) * nc + adjCol[a] / agg);
	vals[a] = (uint32_t)a;
}

__global__ __launch_bounds__(T) void heads_u32_kernel(const uint32_t* keys, int n, int* head)
{
	const int j = blockIdx.x * T + threadIdx.x;
	if (j < n) head[j] = (j == 0 || keys[j] != keys[j - 1]) ? 1 : 0;
}

__device__ __forceinline__ Scalar coarse_weight(int pose, int agg, int Pf)      // same formula as agg_weight() of the PCG kernels
{
	if (pose == Pf - 1 && Pf % agg == 1) return Scalar(0);
	return Scalar(2 * (pose % agg) + 1 - agg) / Scalar(agg);
}

__global__ __launch_bounds__(T) void coarse_lists_kernel(const uint32_t* keys, const uint32_t* order, const int* cbOfEntry, const int* adjBlk, const int* adjRow,
	const int* adjCol, int nAdj, int agg, int nc, int Pf, int cl, int* cbI, int* cbJ, int* cbPtr, int* cbBlk, Scalar* cbWi, Scalar* cbWj, int* counters)
{
	const int x = blockIdx.x * T + threadIdx.x;
	if (x >= nAdj) return;
	const uint32_t key = keys[x];
	const int a = (int)order[x], cb = cbOfEntry[x] - 1;
	if (x == 0 || keys[x - 1] != key) { cbI[cb] = (int)(key / (uint32_t)nc); cbJ[cb] = (int)(key % (uint32_t)nc); cbPtr[cb] = x; }
	cbBlk[x] = adjBlk[a];
	if (cl == 2) { cbWi[x] = coarse_weight(adjRow[a], agg, Pf); cbWj[x] = coarse_weight(adjCol[a], agg, Pf); }
	if (x == nAdj - 1) { cbPtr[cb + 1] = nAdj; counters[CNT_NCB] = cb + 1; }
}

// ---- wave list ---------------------------------------------------------------------------------------------------------
// one thread per chunk of WAVE_CHUNK landmarks; WRITE = false: count waves / big landmarks, true: emit
template <bool WRITE>
__global__ __launch_bounds__(64) void wave_list_kernel(const int* lm_ptr, int lo, int hi, const int* chunkOfs, int* chunkCounts,
	int* wave_lm, int* big_lm)
{
	const int c = blockIdx.x * 64 + threadIdx.x;
	const int l0 = lo + c * WAVE_CHUNK;
	if (l0 >= hi) return;
	const int l1 = min(hi, l0 + WAVE_CHUNK);
	int nW = 0, nB = 0;
	int w = 0, bI = 0;
	if (WRITE) { w = chunkOfs[2 * c]; bI = chunkOfs[2 * c + 1]; }
	int start = -1, cnt = 0;
	int prev = lm_ptr[l0];
	for (int l = l0; l < l1; l++)
	{
		const int next = lm_ptr[l + 1];
		const int n = next - prev;
		prev = next;
		if (n > WAVE)
		{
			if (start >= 0 && cnt > 0) { if (WRITE) { wave_lm[2 * (w + nW)] = start; wave_lm[2 * (w + nW) + 1] = l; } nW++; }
			start = -1; cnt = 0;
			if (WRITE) big_lm[bI + nB] = l;
			nB++;
			continue;
		}
		if (n == 0) continue;
		if (start >= 0 && cnt + n > WAVE)
		{
			if (WRITE) { wave_lm[2 * (w + nW)] = start; wave_lm[2 * (w + nW) + 1] = l; }
			nW++; start = -1; cnt = 0;
		}
		if (start < 0) start = l;
		cnt += n;
	}
	if (start >= 0 && cnt > 0) { if (WRITE) { wave_lm[2 * (w + nW)] = start; wave_lm[2 * (w + nW) + 1] = l1; } nW++; }
	if (!WRITE) { chunkCounts[2 * c] = nW; chunkCounts[2 * c + 1] = nB; }
}

// exclusive scan of the 2 counts per chunk by one workgroup (in place), totals -> counters
__global__ __launch_bounds__(1024) void wave_scan_kernel(int* chunkCounts, int nChunks, int* counters)
{
	__shared__ long long sh[2][1024];
	const int t = threadIdx.x;
	const int per = (nChunks + 1023) / 1024;
	const int c0 = min(nChunks, t * per), c1 = min(nChunks, c0 + per);
	long long s[2] = { 0, 0 };
	for (int c = c0; c < c1; c++)
#pragma unroll
		for (int k = 0; k < 2; k++) s[k] += chunkCounts[2 * c + k];
#pragma unroll
	for (int k = 0; k < 2; k++) sh[k][t] = s[k];
	__syncthreads();
	for (int off = 1; off < 1024; off <<= 1)
	{
		long long v[2] = { 0, 0 };
		if (t >= off)
#pragma unroll
			for (int k = 0; k < 2; k++) v[k] = sh[k][t - off];
		__syncthreads();
#pragma unroll
		for (int k = 0; k < 2; k++) sh[k][t] += v[k];
		__syncthreads();
	}
	long long run[2];
#pragma unroll
	for (int k = 0; k < 2; k++) run[k] = sh[k][t] - s[k];          // exclusive prefix of this thread's segment
	for (int c = c0; c < c1; c++)
#pragma unroll
		for (int k = 0; k < 2; k++) { const int v = chunkCounts[2 * c + k]; chunkCounts[2 * c + k] = (int)run[k]; run[k] += v; }
	if (t == 1023) { counters[CNT_NWAVES] = (int)sh[0][1023]; counters[CNT_NBIG] = (int)sh[1][1023]; }
}

__global__ __launch_bounds__(T) void unsort_kernel(const uint32_t* perm, const Scalar* sorted, int E, double* callerOrder)
{
	const int i = blockIdx.x * T + threadIdx.x;
	if (i < E) callerOrder[perm[i]] = (double)sorted[i];
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------------------
void launch_edge_keys(const int* ep, const int* el, const uint8_t* dim, int E, int Pt, int Pf, int Lt, int Lf, uint64_t* keys, uint32_t* vals, int* counters, hipStream_t s)
{
	if (E > 0) hipLaunchKernelGGL(edge_keys_kernel, grid_for(E), dim3(T), 0, s, ep, el, dim, E, Pt, Pf, Lt, Lf, keys, vals, counters);
}

void launch_gather_edges(const uint32_t* perm, const int* ep, const int* el, const uint8_t* dim, const double* meas, const double* omega, int E,
	int* e_pose, int* e_lm, Scalar* mu, Scalar* mv, Scalar* mr, Scalar* w, hipStream_t s)
{
	if (E > 0) hipLaunchKernelGGL(gather_edges_kernel, grid_for(E), dim3(T), 0, s, perm, ep, el, dim, meas, omega, E, e_pose, e_lm, mu, mv, mr, w);
}

void launch_scatter_values(const int* ids, const double* packed, int n, double* meas, double* omega, hipStream_t s)
{
	if (n > 0) hipLaunchKernelGGL(scatter_values_kernel, grid_for(n), dim3(T), 0, s, ids, packed, n, meas, omega);
}

void launch_segment_ptr(const int* keys, int n, int nSeg, int* ptr, hipStream_t s)
{
	hipLaunchKernelGGL(segment_ptr_kernel, grid_for((size_t)n + 1), dim3(T), 0, s, keys, n, nSeg, ptr);
}

void launch_lm_pairs(const int* lm_ptr, const int* e_pose, int Lf, int Pf, int* nfree, long long* pairCount, long long* freeCount, hipStream_t s)
{
	hipLaunchKernelGGL(lm_pairs_kernel, grid_for((size_t)Lf + 1), dim3(T), 0, s, lm_ptr, e_pose, Lf, Pf, nfree, pairCount, freeCount);
}

void launch_pose_keys(const int* e_pose, int E, int Pf, uint32_t* keys, uint32_t* vals, hipStream_t s)
{
	if (E > 0) hipLaunchKernelGGL(pose_keys_kernel, grid_for(E), dim3(T), 0, s, e_pose, E, Pf, keys, vals);
}

__global__ void gather_int_kernel(const int* __restrict__ idx, const int* __restrict__ src, size_t n, int* __restrict__ dst)
{
	const size_t i = (size_t)blockIdx.x * T + threadIdx.x;
	if (i < n) dst[i] = src[idx[i]];
}

void launch_gather_int(const int* idx, const int* src, size_t n, int* dst, hipStream_t s)
{
	if (n > 0) hipLaunchKernelGGL(gather_int_kernel, grid_for(n), dim3(T), 0, s, idx, src, n, dst);
}

void launch_copy_u32_to_int(const uint32_t* in, int* out, int n, hipStream_t s)
{
	if (n > 0) hipLaunchKernelGGL(copy_u32_int_kernel, grid_for(n), dim3(T), 0, s, in, out, n);
}

void launch_pattern_entries(const int* lm_ptr, const int* e_pose, const int* e_lm, const int* nfree, const long long* pairBase, int E, int Lf, int Pf,
	uint64_t* keys, uint64_t* vals, hipStream_t s)
{
	const size_t n = (size_t)(E > Pf ? E : Pf);
	if (n > 0) hipLaunchKernelGGL(pattern_entries_kernel, grid_for(n), dim3(T), 0, s, lm_ptr, e_pose, e_lm, nfree, pairBase, E, Lf, Pf, keys, vals);
}

namespace
{
__global__ __launch_bounds__(T) void lm_first_last_init_kernel(int Lt, int* first, int* last)
{
	const int l = blockIdx.x * T + threadIdx.x;
	if (l < Lt) { first[l] = 0x7fffffff; last[l] = -1; }
}
__global__ __launch_bounds__(T) void lm_first_last_kernel(const int* __restrict__ ep, const int* __restrict__ el, int E, int* first, int* last)
{
	const int e = blockIdx.x * T + threadIdx.x;
	if (e >= E) return;
	atomicMin(&first[el[e]], ep[e]);          // (integer min / max: the result does not depend on the order of the atomics)
	atomicMax(&last[el[e]], ep[e]);
}
__global__ __launch_bounds__(T) void lm_order_keys_kernel(const int* first, const int* last, int Lf, uint64_t* keys, uint32_t* vals)
{
	const int l = blockIdx.x * T + threadIdx.x;
	if (l >= Lf) return;
	keys[l] = ((uint64_t)(uint32_t)first[l] << 32) | (uint32_t)(last[l] + 1);
	vals[l] = (uint32_t)l;
}
__global__ __launch_bounds__(T) void lm_order_map_kernel(const uint32_t* order, int Lf, int Lt, int* newOfOld)
{
	const int r = blockIdx.x * T + threadIdx.x;
	if (r < Lf) newOfOld[order[r]] = r;
	else if (r < Lt) newOfOld[r] = r;
}
__global__ __launch_bounds__(T) void remap_landmarks_kernel(const int* __restrict__ elIn, const int* __restrict__ newOfOld, int E, int* elOut)
{
	const int e = blockIdx.x * T + threadIdx.x;
	if (e < E) elOut[e] = newOfOld[elIn[e]];
}
__global__ __launch_bounds__(T) void permute_rows_kernel(const Scalar* __restrict__ src, Scalar* __restrict__ dst, const int* __restrict__ newOfOld, size_t total, int width, int toInternal)
{
	const size_t i = (size_t)blockIdx.x * T + threadIdx.x;
	if (i >= total) return;
	const size_t l = i / width; const int k = (int)(i - l * width);
	const size_t m = (size_t)newOfOld[l];
	if (toInternal) dst[m * width + k] = src[i]; else dst[i] = src[m * width + k];
}
}

void launch_lm_first_last(const int* ep, const int* el, int E, int Lt, int* first, int* last, hipStream_t s)
{
	if (Lt > 0) hipLaunchKernelGGL(lm_first_last_init_kernel, grid_for((size_t)Lt), dim3(T), 0, s, Lt, first, last);
	if (E > 0) hipLaunchKernelGGL(lm_first_last_kernel, grid_for((size_t)E), dim3(T), 0, s, ep, el, E, first, last);
}
void launch_lm_order_keys(const int* first, const int* last, int Lf, uint64_t* keys, uint32_t* vals, hipStream_t s)
{
	if (Lf > 0) hipLaunchKernelGGL(lm_order_keys_kernel, grid_for((size_t)Lf), dim3(T), 0, s, first, last, Lf, keys, vals);
}
void launch_lm_order_map(const uint32_t* order, int Lf, int Lt, int* newOfOld, hipStream_t s)
{
	if (Lt > 0) hipLaunchKernelGGL(lm_order_map_kernel, grid_for((size_t)Lt), dim3(T), 0, s, order, Lf, Lt, newOfOld);
}
void launch_remap_landmarks(const int* elIn, const int* newOfOld, int E, int* elOut, hipStream_t s)
{
	if (E > 0) hipLaunchKernelGGL(remap_landmarks_kernel, grid_for((size_t)E), dim3(T), 0, s, elIn, newOfOld, E, elOut);
}
void launch_permute_rows(const Scalar* src, Scalar* dst, const int* newOfOld, int nrows, int width, bool toInternal, hipStream_t s)
{
	const size_t total = (size_t)nrows * width;
	if (total) hipLaunchKernelGGL(permute_rows_kernel, grid_for(total), dim3(T), 0, s, src, dst, newOfOld, total, width, toInternal ? 1 : 0);
}

void launch_entry_heads(const uint64_t* keys, size_t n, int* head, hipStream_t s)
{
	if (n > 0) hipLaunchKernelGGL(entry_heads_kernel, grid_for(n), dim3(T), 0, s, keys, n, head);
}

void launch_blocks_from_entries(const uint64_t* keys, const uint64_t* vals, const int* blkOfEntry, size_t n, int Pf, int* colind, int* blkrow, int* prod_ptr,
	int* prod_ea, int* prod_eb, hipStream_t s)
{
	if (n > 0) hipLaunchKernelGGL(blocks_from_entries_kernel, grid_for(n), dim3(T), 0, s, keys, vals, blkOfEntry, n, Pf, colind, blkrow, prod_ptr, prod_ea, prod_eb);
}

void launch_segment_subrange(const int* ptr, int nseg, const int* vals, int vlo, int vhi, int* beg, int* end, hipStream_t s)
{
	if (nseg > 0) hipLaunchKernelGGL(segment_subrange_kernel, grid_for(nseg), dim3(T), 0, s, ptr, nseg, vals, vlo, vhi, beg, end);
}

void launch_od_keys(const int* prod_beg, const int* prod_end, const int* blkrow, const int* colind, int nblk, int farOffset, int heavy, uint32_t* keys, uint32_t* vals, int* counters, hipStream_t s)
{
	if (nblk > 0) hipLaunchKernelGGL(od_keys_kernel, grid_for(nblk), dim3(T), 0, s, prod_beg, prod_end, blkrow, colind, nblk, farOffset, heavy, keys, vals, counters);
}

void launch_remap_poses(const int* epIn, const int* newOfOld, int E, int Pf, int* epOut, hipStream_t s)
{
	if (E > 0) hipLaunchKernelGGL(remap_poses_kernel, grid_for(E), dim3(T), 0, s, epIn, newOfOld, E, Pf, epOut);
}

void launch_transpose_keys(const int* colind, const int* blkrow, int nblk, uint64_t* keys, uint32_t* vals, hipStream_t s)
{
	if (nblk > 0) hipLaunchKernelGGL(transpose_keys_kernel, grid_for(nblk), dim3(T), 0, s, colind, blkrow, nblk, keys, vals);
}

void launch_keys_hi(const uint64_t* keys, int n, int limit, int* hi, hipStream_t s)
{
	if (n > 0) hipLaunchKernelGGL(keys_hi_kernel, grid_for(n), dim3(T), 0, s, keys, n, limit, hi);
}

void launch_ell(const int* adjPtr, const int* adjBlk, const int* adjCol, int Pf, int M, int2* ell, hipStream_t s)
{
	const size_t n = (size_t)Pf * 20 * M;
	if (n > 0) hipLaunchKernelGGL(ell_kernel, grid_for(n), dim3(T), 0, s, adjPtr, adjBlk, adjCol, Pf, M, ell);
}

void launch_coarse_keys(const int* adjRow, const int* adjCol, int nAdj, int agg, int nc, uint32_t* keys, uint32_t* vals, hipStream_t s)
{
	if (nAdj > 0) hipLaunchKernelGGL(coarse_keys_kernel, grid_for(nAdj), dim3(T), 0, s, adjRow, adjCol, nAdj, agg, nc, keys, vals);
}

void launch_heads_u32(const uint32_t* keys, int n, int* head, hipStream_t s)
{
	if (n > 0) hipLaunchKernelGGL(heads_u32_kernel, grid_for(n), dim3(T), 0, s, keys, n, head);
}

void launch_coarse_lists(const uint32_t* keys, const uint32_t* order, const int* cbOfEntry, const int* adjBlk, const int* adjRow, const int* adjCol, int nAdj,
	int agg, int nc, int Pf, int cl, int* cbI, int* cbJ, int* cbPtr, int* cbBlk, Scalar* cbWi, Scalar* cbWj, int* counters, hipStream_t s)
{
	if (nAdj > 0) hipLaunchKernelGGL(coarse_lists_kernel, grid_for(nAdj), dim3(T), 0, s, keys, order, cbOfEntry, adjBlk, adjRow, adjCol, nAdj, agg, nc, Pf, cl,
		cbI, cbJ, cbPtr, cbBlk, cbWi, cbWj, counters);
}

void launch_wave_count(const int* lm_ptr, int lo, int hi, int* chunkCounts, hipStream_t s)
{
	const int nChunks = (hi - lo + WAVE_CHUNK - 1) / WAVE_CHUNK;
	if (nChunks > 0) hipLaunchKernelGGL(wave_list_kernel<false>, dim3((nChunks + 63) / 64), dim3(64), 0, s, lm_ptr, lo, hi, (const int*)nullptr, chunkCounts,
		(int*)nullptr, (int*)nullptr);
}

void launch_wave_scan(int* chunkCounts, int nChunks, int* counters, hipStream_t s)
{
	hipLaunchKernelGGL(wave_scan_kernel, dim3(1), dim3(1024), 0, s, chunkCounts, nChunks, counters);
}

void launch_wave_write(const int* lm_ptr, int lo, int hi, const int* chunkOfs, int* wave_lm, int* big_lm, hipStream_t s)
{
	const int nChunks = (hi - lo + WAVE_CHUNK - 1) / WAVE_CHUNK;
	if (nChunks > 0) hipLaunchKernelGGL(wave_list_kernel<true>, dim3((nChunks + 63) / 64), dim3(64), 0, s, lm_ptr, lo, hi, chunkOfs, (int*)nullptr, wave_lm, big_lm);
}

void launch_unsort(const uint32_t* perm, const Scalar* sorted, int E, double* callerOrder, hipStream_t s)
{
	if (E > 0) hipLaunchKernelGGL(unsort_kernel, grid_for(E), dim3(T), 0, s, perm, sorted, E, callerOrder);
}

// adjacency in two launches (pointer array + maximum row length, then the fill)
void launch_adj_ptr(const int* rowptr, const int* lowerPtr, int Pf, int* adjPtr, int* counters, hipStream_t s)
{
	hipLaunchKernelGGL(adj_ptr_kernel, grid_for((size_t)Pf + 1), dim3(T), 0, s, rowptr, lowerPtr, Pf, adjPtr, counters);
}

void launch_adj_fill(const int* rowptr, const int* colind, const int* blkrow, const int* lowerPtr, const uint64_t* tKeys, const uint32_t* tBlk, int nblk,
	const int* adjPtr, int* adjBlk, int* adjCol, int* adjRow, hipStream_t s)
{
	if (nblk > 0) hipLaunchKernelGGL(adj_fill_kernel, grid_for(nblk), dim3(T), 0, s, rowptr, colind, blkrow, lowerPtr, tKeys, tBlk, nblk, adjPtr, adjBlk, adjCol, adjRow);
}

}  // namespace topo
}  // namespace cubahip
