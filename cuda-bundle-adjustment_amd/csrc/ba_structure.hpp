// ba_structure.hpp -- device-side set-up: edge sort (landmark, pose) and the symbolic structure of the reduced system.
//
// Role of the reference's gpu::buildHplStructure / gpu::findHschureMulBlockIndices (thrust sorts + scan,
// /root/reference/src/cuda_block_solver.cu:1158-1190) and of HschurSparseBlockMatrix::constructFromVertices
// (src/sparse_block_matrix.cpp:55-133, host, dense P x P map) -- here everything is sort / scan / segment based and runs on
// the GPU: radix sorts of (landmark, pose) and (row, column) keys (rocPRIM, the counterpart of the reference's Thrust calls),
// head flags + scans for the block pattern, closed-form positions for the product lists.  A new topology then costs
// ~1 ms instead of the 7-8 ms of the host pipeline (ba_solver.hip keeps that one for landmark-partitioned handles).
#pragma once

#include <cstddef>
#include <cstdint>

#include <hip/hip_runtime.h>

#include "ba_math.hpp"

namespace cubahip
{
namespace topo
{

// ---- rocPRIM wrappers (temp storage is the caller's; *_temp_bytes give an upper bound for n elements) ----------------
size_t sort_temp_bytes(size_t n);
size_t scan_temp_bytes(size_t n);
hipError_t sort_u64_u32(void* temp, size_t tempBytes, const uint64_t* kin, uint64_t* kout, const uint32_t* vin, uint32_t* vout, size_t n, int endBit, hipStream_t s);
hipError_t sort_u64_u64(void* temp, size_t tempBytes, const uint64_t* kin, uint64_t* kout, const uint64_t* vin, uint64_t* vout, size_t n, int endBit, hipStream_t s);
hipError_t sort_u32_u32(void* temp, size_t tempBytes, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, size_t n, int endBit, hipStream_t s);
hipError_t exclusive_scan_i64(void* temp, size_t tempBytes, const long long* in, long long* out, size_t n, hipStream_t s);
hipError_t inclusive_scan_i32(void* temp, size_t tempBytes, const int* in, int* out, size_t n, hipStream_t s);

// counters the kernels fill (one device int each), read back by the host at its synchronisation points
enum { CNT_BAD = 0, CNT_FREE_EDGES, CNT_NOD, CNT_MAXROW, CNT_NCB, CNT_NWAVES, CNT_NBIG, CNT_FARBLOCKS, CNT_DIAGPROD, CNT_NHEAVY, CNT_COUNT = 16 };

// ---- A. edges ------------------------------------------------------------------------------------------------------
// keys[e] = landmark << 32 | pose, vals[e] = e; counters[CNT_BAD] = 1 / 2 / 3 for an index out of range / a bad dimension /
// an edge with both ends fixed
void launch_edge_keys(const int* ep, const int* el, const uint8_t* dim, int E, int Pt, int Pf, int Lt, int Lf,
	uint64_t* keys, uint32_t* vals, int* counters, hipStream_t s);
// sorted edge arrays from the caller-order ones through the sort permutation
void launch_gather_edges(const uint32_t* perm, const int* ep, const int* el, const uint8_t* dim, const double* meas, const double* omega, int E,
	int* e_pose, int* e_lm, Scalar* mu, Scalar* mv, Scalar* mr, Scalar* w, hipStream_t s);
// values of a subset of the edges into the caller-order arrays: meas[3 * ids[i] ..] = packed[4 * i .. + 2], omega[ids[i]] = packed[4 * i + 3]
void launch_scatter_values(const int* ids, const double* packed, int n, double* meas, double* omega, hipStream_t s);
// ptr[k] = first position i with keys[i] >= k, k = 0..nSeg (keys ascending; ptr[nSeg] counts the keys < nSeg)
void launch_segment_ptr(const int* keys, int n, int nSeg, int* ptr, hipStream_t s);

// ---- B. structure ----------------------------------------------------------------------------------------------------
// per free landmark: number of edges with a free pose, number of pose pairs n(n-1)/2 (pairCount has Lf + 1 entries, the
// last one 0, so that an exclusive scan leaves the total in pairBase[Lf]); freeCount likewise holds nfree as 64-bit numbers
void launch_lm_pairs(const int* lm_ptr, const int* e_pose, int Lf, int Pf, int* nfree, long long* pairCount, long long* freeCount, hipStream_t s);
// pose key of every sorted edge (Pf for edges whose pose is fixed) + identity values, for the per-pose edge lists
void launch_pose_keys(const int* e_pose, int E, int Pf, uint32_t* keys, uint32_t* vals, hipStream_t s);
void launch_copy_u32_to_int(const uint32_t* in, int* out, int n, hipStream_t s);
void launch_gather_int(const int* idx, const int* src, size_t n, int* dst, hipStream_t s);     // dst[i] = src[idx[i]]
// pattern entries: Pf diagonal seeds first, then every pair (a < c) of free-pose edges of every free landmark in
// product-id order: key = row << 32 | column, value = (edge a + 1) << 32 | (edge c + 1), 0 for a seed
void launch_pattern_entries(const int* lm_ptr, const int* e_pose, const int* e_lm, const int* nfree, const long long* pairBase,
	int E, int Lf, int Pf, uint64_t* keys, uint64_t* vals, hipStream_t s);
void launch_entry_heads(const uint64_t* keys, size_t n, int* head, hipStream_t s);
// blocks and product lists from the sorted entries (blkOfEntry = inclusive scan of the head flags)
void launch_blocks_from_entries(const uint64_t* keys, const uint64_t* vals, const int* blkOfEntry, size_t n, int Pf,
	int* colind, int* blkrow, int* prod_ptr, int* prod_ea, int* prod_eb, hipStream_t s);
// blocks with products, longest list first (stable): sort keys + values; counters[CNT_NOD] = their number
// counters[CNT_FARBLOCKS] = number of blocks more than farOffset block columns off the diagonal (pose order check)
// counters[CNT_DIAGPROD] = number of DIAGONAL blocks with products (a landmark observed twice by one pose)
// counters[CNT_NHEAVY] = number of blocks with more than `heavy` products (the first ones of the sorted list)
// (prod_beg[k] .. prod_end[k] = the product range of block k this handle evaluates: prod_ptr / prod_ptr + 1 for the whole graph)
void launch_od_keys(const int* prod_beg, const int* prod_end, const int* blkrow, const int* colind, int nblk, int farOffset, int heavy, uint32_t* keys, uint32_t* vals, int* counters, hipStream_t s);
// beg[s] .. end[s] = the items of segment s (ptr[s] .. ptr[s + 1], values ascending) whose value lies in [vlo, vhi)
void launch_segment_subrange(const int* ptr, int nseg, const int* vals, int vlo, int vhi, int* beg, int* end, hipStream_t s);
// pose indices of the caller-order edge array through a map of the free poses (fixed poses keep their index)
void launch_remap_poses(const int* epIn, const int* newOfOld, int E, int Pf, int* epOut, hipStream_t s);
// transposed view of the off-diagonal blocks: key = column << 32 | row (diagonal blocks get the largest key)
void launch_transpose_keys(const int* colind, const int* blkrow, int nblk, uint64_t* keys, uint32_t* vals, hipStream_t s);
void launch_keys_hi(const uint64_t* keys, int n, int limit, int* hi, hipStream_t s);       // hi[i] = min(keys[i] >> 32, limit)
// symmetric adjacency (lower neighbours first, then the row's own blocks): adjPtr[i] = lowerPtr[i] + rowptr[i] and
// counters[CNT_MAXROW], then the fill from the transposed (column, row)-sorted block list
void launch_adj_ptr(const int* rowptr, const int* lowerPtr, int Pf, int* adjPtr, int* counters, hipStream_t s);
void launch_adj_fill(const int* rowptr, const int* colind, const int* blkrow, const int* lowerPtr, const uint64_t* tKeys, const uint32_t* tBlk, int nblk,
	const int* adjPtr, int* adjBlk, int* adjCol, int* adjRow, hipStream_t s);
void launch_ell(const int* adjPtr, const int* adjBlk, const int* adjCol, int Pf, int M, int2* ell, hipStream_t s);
// coarse-matrix assembly lists: key = coarse block of every adjacency entry
void launch_coarse_keys(const int* adjRow, const int* adjCol, int nAdj, int agg, int nc, uint32_t* keys, uint32_t* vals, hipStream_t s);
void launch_heads_u32(const uint32_t* keys, int n, int* head, hipStream_t s);
void launch_coarse_lists(const uint32_t* keys, const uint32_t* order, const int* cbOfEntry, const int* adjBlk, const int* adjRow, const int* adjCol,
	int nAdj, int agg, int nc, int Pf, int cl, int* cbI, int* cbJ, int* cbPtr, int* cbBlk, Scalar* cbWi, Scalar* cbWj, int* counters, hipStream_t s);
// wave work list of the landmark-major kernels: whole landmarks, at most 64 edges per wave; larger landmarks apart.
// Two passes over chunks of WAVE_CHUNK landmarks (each chunk starts a new wave): count, then (after a scan by one
// workgroup) write.  chunkCounts has 2 ints per chunk.
constexpr int WAVE_CHUNK = 256;
void launch_wave_count(const int* lm_ptr, int lo, int hi, int* chunkCounts, hipStream_t s);
void launch_wave_scan(int* chunkCounts, int nChunks, int* counters, hipStream_t s);
void launch_wave_write(const int* lm_ptr, int lo, int hi, const int* chunkCounts, int* wave_lm, int* big_lm, hipStream_t s);
// ---- internal landmark order (round 5) ---------------------------------------------------------------------------------
// first[l] / last[l] = smallest / largest pose index among the edges of landmark l (first = INT_MAX, last = -1 for a landmark without edges)
void launch_lm_first_last(const int* ep, const int* el, int E, int Lt, int* first, int* last, hipStream_t s);
// keys[l] = first << 32 | last, vals[l] = l for the free landmarks l < Lf
void launch_lm_order_keys(const int* first, const int* last, int Lf, uint64_t* keys, uint32_t* vals, hipStream_t s);
// newOfOld[order[r]] = r for r < Lf, newOfOld[l] = l for Lf <= l < Lt
void launch_lm_order_map(const uint32_t* order, int Lf, int Lt, int* newOfOld, hipStream_t s);
void launch_remap_landmarks(const int* elIn, const int* newOfOld, int E, int* elOut, hipStream_t s);
// rows of `width` numbers: toInternal: dst[newOfOld[l]] = src[l]; else dst[l] = src[newOfOld[l]]  (l < nrows; src != dst)
void launch_permute_rows(const Scalar* src, Scalar* dst, const int* newOfOld, int nrows, int width, bool toInternal, hipStream_t s);

// per-edge values from sorted order back to the caller's order
void launch_unsort(const uint32_t* perm, const Scalar* sorted, int E, double* callerOrder, hipStream_t s);

}  // namespace topo
}  // namespace cubahip
