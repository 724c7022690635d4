// ba_kernels.hpp -- launch interface of the gfx950 kernels (implemented in ba_edge.hip, ba_linearize.hip, ba_pcg.hip, ba_coarse.hip).
//
// Device data model (all SoA, fp64 + int32):
//   poses      q[4*Pt] t[3*Pt] cam[5*Pt]          free poses [0,Pf) first, fixed after
//   landmarks  Xw[3*Lt]                           free [0,Lf) first
//   edges      sorted by (landmark, pose): e_pose (bit 31 = stereo), e_lm, e_mu/e_mv/e_mr, e_w
//              lm_ptr[Lt+1] = edge range of each landmark
//   wave list  wave_lm[2*nWaves]: each 64-lane wavefront owns whole landmarks with <= 64 edges in total,
//              one lane per edge -> Hll/bl are reduced inside the wave, no atomics on the landmark side
//   Hsc        upper-triangular BSR (row_ptr/col_ind, 6x6 col-major blocks, diagonal block first in
//              each row), prod_* = per block the edge pairs of its Schur products, adj_* = full symmetric
//              adjacency over the same storage for the PCG SpMV
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>
#include <vector>

#include "ba_math.hpp"

namespace cubahip
{

constexpr int PCG_SETUP_POSES = 64;   // poses per workgroup of the PCG set-up (= its partial sums of r0.z0 in block-Jacobi-only mode)
constexpr int NSLOT = 16;            // partial-sum slots for global reductions (spreads same-address atomics)
constexpr int WAVE = 64;
constexpr int LIN_BLOCK = 256;       // 4 wavefronts per workgroup in the landmark-major kernels
constexpr int STEREO_BIT = 0x40000000;

struct DeviceGraph
{
	int Pt = 0, Pf = 0, Lt = 0, Lf = 0, E = 0;
	int e_begin = 0, e_end = 0;        // sorted-edge range this handle evaluates (whole graph unless landmark-partitioned)
	Scalar *q = nullptr, *t = nullptr, *cam = nullptr, *Xw = nullptr;
	int *e_pose = nullptr, *e_lm = nullptr;
	Scalar *e_mu = nullptr, *e_mv = nullptr, *e_mr = nullptr, *e_w = nullptr;
	int* lm_ptr = nullptr;
	RobustKernel rk[2] = { { 0, 0 }, { 0, 0 } };
};

constexpr int BP_HEAVY = 64;      // blocks of Hsc with more products than this get a whole wave in the block pass

struct DeviceStructure
{
	int nWaves = 0;
	int* wave_lm = nullptr;            // [2*nWaves] first / one-past-last landmark of each wave
	int nBig = 0;
	int* big_lm = nullptr;             // landmarks with more than 64 edges (own workgroup each)
	int nblk = 0;
	int *hsc_rowptr = nullptr, *hsc_colind = nullptr;
	int *adj_ptr = nullptr, *adj_blk = nullptr, *adj_col = nullptr;  // adj_blk bit 31 = use transposed
	// the first 20*ell_m entries of every adjacency row again, padded to a fixed width and interleaved so that lane
	// (slot, m) finds its (block, column) pair at ((row*ell_m + m)*20 + slot) without reading adj_ptr first;
	// column -1 = padding. ell_over != 0 when some row has more entries (those stay in adj_* only).
	int2* ell = nullptr;
	int ell_m = 0, ell_over = 0;
	// destination-major (atomic-free) Schur assembly
	int* hsc_blkrow = nullptr;         // [nblk] block row of every block
	int nOd = 0;                       // blocks that receive at least one off-diagonal (or duplicate-pose) product
	int nDiagProd = 0;                 // of these, diagonal blocks (a landmark observed twice by one pose): the block pass then updates what the pose pass stored
	int* od_blocks = nullptr;          // [nOd] their ids, largest product count first
	int inv_rows8 = 1;                 // 1 = the landmark pass also writes inv(Hll + lambda I) as 64-byte rows (lm_inv) and the block pass gathers those;
	                                   // 0 = 48-byte gathers from the 72-byte rows of lm_sys (small graphs: the extra stores cost the landmark pass more than the block pass gains)
	int nHeavy = 0;                    // the first nHeavy of them have more than BP_HEAVY products: a whole wave each in the block pass
	int *prod_ea = nullptr, *prod_eb = nullptr;   // sorted edge ids of each product (ea: row pose, eb: column pose)
	// the ranges of the product list (per block) and of the pose edge list (per free pose) that THIS handle walks: consecutive
	// entries of the [n + 1] range arrays for a whole graph, sub-ranges (the lists are in landmark order) for a landmark
	// partition built on the device
	const int *prod_beg = nullptr, *prod_end = nullptr, *pe_beg = nullptr, *pe_end = nullptr;
	int* prod_lm = nullptr;            // landmark of each product (= e_lm[prod_ea]): the block pass then fetches inv(Hll + lambda) beside the
	                                   // two edge records instead of after them (one memory round trip per product instead of two)
	int* pe_edge = nullptr;            // per free pose: its sorted edge ids (ranges: pe_beg / pe_end)
	// coarse-matrix assembly lists: for every non-empty coarse block (I,J) the fine blocks that fall into it
	int nCb = 0;                       // non-empty coarse blocks
	int *cb_I = nullptr, *cb_J = nullptr, *cb_ptr = nullptr, *cb_blk = nullptr;   // cb_blk: adjacency-style id (bit 31 = transposed)
	Scalar *cb_wi = nullptr, *cb_wj = nullptr;                                    // weights of the fine (row, column) poses of every list entry in the linear coarse functions
	Scalar* e_rec = nullptr;           // [8*E] per-edge linearisation record {Xc[3], w' (sign bit = stereo), r[3], landmark (integer bits)}
	int mixed = 0;                     // fp64 library only: 1 = records and the per-edge arithmetic of the pose / block passes in fp32
};

int spmv_rows_for(int Pf);      // block rows per SpMV workgroup (two waves each): 2, or 4 for large graphs

struct DeviceSystem
{
	Scalar* hsc = nullptr;     // [36*nblk]  } one contiguous allocation (multi-GPU reduction buffer)
	Scalar* bsc = nullptr;     // [6*Pf]     }
	Scalar* bp = nullptr;      // [6*Pf]     }
	Scalar* lm_sys = nullptr;  // [9*Lf]  6 unique of Hll or inv(Hll+lambda I), then bl
	Scalar* lm_inv = nullptr;  // [8*Lf]  copy of inv(Hll+lambda I) in 64-byte rows for the block pass (a 48-byte gather from the 72-byte rows of lm_sys
	                           // straddles two 64-byte sectors three times out of four, and the pass pays per sector: profiles/r03x_block_pass_where.txt)
	Scalar* xp = nullptr;      // [6*Pf]
	Scalar* xl = nullptr;      // [3*Lf]
	Scalar* parts = nullptr;   // per-workgroup partial sums (first stage of the deterministic global reductions)
	Scalar* slots = nullptr;   // [4*NSLOT] chi2 | landmark scale (back_substitute) | per-edge chi2 scratch | stage scale
	unsigned long long* maxdiag = nullptr;  // bit pattern of a non-negative double
	int* fail = nullptr;       // numeric failure flag
	// PCG work
	Scalar *minv = nullptr, *r = nullptr, *z = nullptr, *p0 = nullptr, *p1 = nullptr, *ap = nullptr;
	// CG scalars as per-workgroup partial sums in small rings (no atomics => fixed summation order, nothing to zero):
	//   rz: slot 0 = r0.z0 (kept for the stop test), slot 1+(k&3) = r_k.z_k (k = 0: a second copy of slot 0, so that no
	//   address depends on the absolute iteration number);  pq: slot k&3 = p_k.A p_k
	Scalar *rz = nullptr, *pq = nullptr;
	int rzStride = 0, pqStride = 0;        // entries per ring slot
	int nrz0 = 0, nrz = 0, npq = 0;        // number of partials in slot 0 / in the ring slots of rz (nrz0 <= nrz) / in pq slots
	int* done = nullptr;                   // set once the stop test fails: later queued launches return at once
	int* iters = nullptr;      // device iteration counter
	int* host_flags = nullptr; // device-mapped host ints: {fail, iterations done, stop flag, ticket}, refreshed by the last node of every
	                           // iteration graph and by launch_pcg_report
	int* ticket = nullptr;     // device counter of those reports
	int* kbase = nullptr;      // iteration offset added to the k / kOut kernel arguments (lets one hipGraph
	                           // of `chunk` iterations be replayed: the graph's last node advances it by `chunk`)
	// two-level preconditioner: aggregates of `agg` consecutive free poses, 6 coarse dof each
	int agg = 0;               // 0 = block-Jacobi only
	int nc = 0;                // number of aggregates
	Scalar inv_agg = 0;        // 1 / agg
	int cl = 1;                // coarse functions per aggregate and pose component: 1 = constant, 2 = constant + linear in the pose
	                           // index (coarse dimension = 6*cl*nc)
	Scalar* acinv = nullptr;   // [(6nc)^2] explicit inverse of the coarse matrix P^T A P, column-major
	float* acinv32 = nullptr;  // fp64 library, option "precond_fp32": the same inverse in fp32 (symmetrised, rows padded to a multiple of 4
	                           // numbers) -- what the iteration kernels then read instead of acinv
	Scalar* gj_pivots = nullptr;   // [2][32 x 32] scratch of the Gauss-Jordan sweep (inverse of the current / next pivot block)
	Scalar* rc = nullptr;      // [2*6nc] restricted residual P^T r_k, ping-pong by the parity of k like r / r2 (each
	                           // aggregate's owner workgroup writes its 6 entries of P^T r_{k+1})
	Scalar* hrow = nullptr;    // [36 * 20 * ell_m * Pf] row-ordered copy of Hsc for the SpMV (launch_hsc_expand), entry (row, m, slot)
	int spmv_rows = 2;         // block rows per SpMV workgroup
	Scalar* qpart = nullptr;   // [agg/spmv_rows][6*cl*nc] (weighted) sums of q = A p over the block rows of each SpMV workgroup,
	                           // indexed by the workgroup's position inside its aggregate (P^T q is summed from these:
	                           // aggregates are whole multiples of spmv_rows rows)
	Scalar* r2 = nullptr;      // second residual buffer (the fused two-level kernel ping-pongs r / r2)
	// device-resident Levenberg-Marquardt decision (cuba_hip_optimize): kernels that are handed a NEGATIVE damping read it from here
	const Scalar* lam_dev = nullptr;
	// upper-triangle iteration (large graphs; ba_pcg.hip): three launches per iteration straight from the upper-triangular BSR storage
	int upper = 0;
	Scalar* tq = nullptr;      // [6 * (nblk - Pf)] transposed products B^T p_i of the off-diagonal blocks, written by the SpMV in the order of the rows that
	                           // own them: the lower neighbours of row j are one contiguous range
	int* lowpos = nullptr;     // [nblk] position of every off-diagonal block in that order (launch_build_lowpos)
};

// residual / robust chi2 over all edges -> sys.slots[0..NSLOT) (must be zeroed by the caller).
// per_edge (optional, sorted edge order): non-robust omega*|r|^2.
void launch_residual_chi2(const DeviceGraph& g, Scalar* parts, Scalar* slots, Scalar* per_edge, hipStream_t st);

// mode 0: assemble only (Hpp -> diagonal blocks of hsc, bp, Hll/bl -> lm_sys, max diagonal of Hll)
// mode 1: full linearise + Schur reduction with damping lambda (hsc, bsc, bp, inv(Hll+lambda)/bl -> lm_sys)
// launch_linearize_dm: destination-major, atomic-free and bitwise reproducible:
//   landmark pass (Hll/bl, inverse, per-edge record) -> pose pass (diagonal blocks, bp, bsc) -> block pass (off-diagonal blocks)
// backupSrc != nullptr: the landmark pass's launch also copies backupCount numbers backupSrc -> backupDst (the LM loop's push())
// range != nullptr (landmark partitions whose reduction is cut into parts, ba_setup.hip: cutReductionParts): the block pass covers the
// entries [begin, end) of st.od_blocks only, the first `heavy` of them with a whole wave each; launch_block_pass runs a further range
struct BlockPassRange { int begin, end, heavy; };
void launch_linearize_dm(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, int mode, Scalar lambda, hipStream_t s,
	const Scalar* backupSrc = nullptr, Scalar* backupDst = nullptr, size_t backupCount = 0, const BlockPassRange* range = nullptr);
void launch_block_pass(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, const BlockPassRange& range, hipStream_t s);

// max over the diagonal of the diagonal blocks of hsc (Hpp after an assemble pass) folded into sys.maxdiag
void launch_pose_maxdiag(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, hipStream_t s);

// xl = inv(Hll+lambda)(bl - Hpl^T xp); accumulates sum xl (lambda xl + bl) into slots[NSLOT..2*NSLOT)
void launch_back_substitute(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, Scalar lambda, hipStream_t s);

// pose-side part of the gain-ratio denominator: sum xp (lambda xp + bp) -> slots[0..NSLOT)
void launch_pose_scale(const DeviceGraph& g, const DeviceSystem& sys, Scalar lambda, Scalar* slots, hipStream_t s);
// back-substitution + update + evaluation of the trial + second stage of their three sums + report to the host in two launches:
// back-substitution, update and evaluation fused into one pass over the edges (`old` = copy of the state
// [q | t | Xw] made before the trial: the pass reads the pre-update estimate from it while it writes the updated one), then the sums
// + report.  trial_tail_parts(): numbers of partial-sum scratch (sys.parts) it needs.
struct LmDevice;
void launch_trial_tail_fused(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, Scalar lambda, const Scalar* old, hipStream_t s,
	const LmDevice* decide = nullptr);
size_t trial_tail_parts(const DeviceGraph& g, const DeviceStructure& st);
// Device-resident LM decision (control flow of CudaBundleAdjustmentImpl::optimize, /root/reference/src/cuda_bundle_adjustment.cpp:816-851):
// state = {F, lambda, nu, halt, trials, accepted (last trial), rejections in a row, max rejections} in device memory, lam = the damping as
// the kernels read it (sys.lam_dev), ring = device-mapped host records, LM_REC numbers per trial {Fhat, denominator, rho, next lambda,
// next F, accepted, halt, next nu}.  launch_trial_tail_fused(.., decide) lets the sums' second stage take the decision of the trial instead
// of only reporting them; launch_lm_decide_failed is the decision of a trial whose reduced solve failed (rho = -1);
// launch_restore_if_rejected copies the backup over the estimates when the last decision was a rejection (the reference's pop()).
constexpr int LM_REC = 8, LM_RING = 64;
struct LmDevice { double* state = nullptr; Scalar* lam = nullptr; double* ring = nullptr; };
void launch_lm_decide_failed(const DeviceSystem& sys, const LmDevice& lm, hipStream_t s);
void launch_restore_if_rejected(Scalar* state, const Scalar* backup, size_t count, const LmDevice& lm, hipStream_t s);
// landmark-side part recomputed from xl and the stored bl (stage API; the fused path gets it from back_substitute)
void launch_landmark_scale(const DeviceGraph& g, const DeviceSystem& sys, Scalar lambda, Scalar* slots, hipStream_t s);

void launch_update_state(const DeviceGraph& g, const DeviceSystem& sys, hipStream_t s);   // poses + landmarks in one launch

// block-Jacobi PCG on the upper-BSR reduced system
void launch_pcg_setup(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, Scalar lambda, hipStream_t s);
void launch_pcg_spmv(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, int k, int maxIter, Scalar tol2, hipStream_t s);
void launch_pcg_update(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, int k, int maxIter, Scalar tol2, hipStream_t s);
// two-level (block-Jacobi + aggregate coarse correction) variant: 3 kernels per iteration
Scalar* launch_coarse_setup(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, Scalar* work0, Scalar* work1, hipStream_t s, hipEvent_t assembled = nullptr);
// blocked Gauss-Jordan inversion of a dense SPD n x n matrix (column-major in work0; work1 = scratch of the same size);
// returns whichever of the two buffers holds the inverse
Scalar* launch_dense_inverse(Scalar* work0, Scalar* work1, int n, Scalar* pivots, hipStream_t s);   // symmetric sweep: the result's upper triangle holds -A^-1; pivots: 2 x 32 x 32 numbers of scratch
void launch_coarse_finish(const Scalar* swept, Scalar* dst, int n, hipStream_t s);    // swept buffer -> full symmetric +A^-1 (dst may be the swept buffer)
// batched sweep of several coarse matrices (cuba_hip_optimize_batch): one launch per step for all of them
struct GjJob { Scalar* buf[2]; Scalar* piv[2]; int n, tiles; };       // buf[0]: the matrix on entry; piv: 2 x (32 x 32) scratch
void launch_coarse_assemble(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, Scalar* work0, hipStream_t s);
void launch_dense_inverse_batch(const GjJob* jobs, int m, int nMax, hipStream_t s);
void launch_pcg2_fused(const DeviceGraph& g, const DeviceSystem& sys, int k, int kOut, int maxIter, Scalar tol2, int doUpdate, hipStream_t s);
// one iteration of the upper-triangle form (sys.upper): which = 1 SpMV | 2 row updates | 4 preconditioner (7 = all three, in this order)
int spmv_upper_grid(int Pf);      // workgroups of the upper-triangle SpMV (= its p.Ap partials)
int pcg_rows_max_aggregate();     // largest aggregate (poses) the row-update launch of the upper-triangle iteration handles
void launch_build_lowpos(const DeviceGraph& g, const DeviceStructure& st, int* lowpos, hipStream_t s);
void launch_pcg_upper_iteration(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, int k, int maxIter, Scalar tol2, hipStream_t s, int which = 7);
// what the last node of an iteration graph does, as a launch: advance the iteration offset by n, run the stop test on the residual the
// chunk left (tol2 >= 0), report to the host
void launch_pcg_advance(const DeviceSystem& sys, int n, hipStream_t s, Scalar tol2 = Scalar(-1));
void launch_coarse_to_fp32(const Scalar* swept, float* dst, int n, hipStream_t s);   // swept buffer -> +A^-1 in the sys.acinv32 layout
void launch_pcg_iteration(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, int k, int maxIter, Scalar tol2, hipStream_t s);

// ---- batched execution (cuba_hip_optimize_batch): the PCG iterations of several graphs as ONE launch chain ---------------------------
// Device table entry of one graph: what its kernels otherwise receive as kernel arguments.  The batched kernels take blockIdx.y as the
// graph number; a graph whose reduced solve has finished keeps its entry (its `done` flag makes its workgroups return at once).
struct LmDevice;
struct BatchTrial               // per-graph arguments of the batched launches around the iterations (a grid of 0: the graph sits this launch out)
{
	// landmark pass (+ state backup), pose + block pass
	unsigned lmGroups = 0, lmGrid = 0; const Scalar* backupSrc = nullptr; Scalar* backupDst = nullptr; size_t backupCount = 0;
	int poseGroups = 0; unsigned schurGrid = 0;
	// PCG set-up launch (+ row-ordered copy + copy of a fresh coarse inverse), first preconditioner application
	int nSetup = 0; size_t expandTotal = 0; unsigned nExpand = 0; const Scalar* copySrc = nullptr; Scalar* copyDst = nullptr; size_t copyPairs = 0; unsigned setupGrid = 0;
	int fusedOn = 0;
	// trial tail, sums + decision + report, conditional restore
	const Scalar* old = nullptr; Scalar *scParts = nullptr, *chiParts = nullptr, *scaleParts = nullptr; int nLm = 0, poseBlocks = 0, nScale = 0; unsigned tailGrid = 0; int nA = 0;
	double* lmState = nullptr; Scalar* lmLam = nullptr; double* lmRing = nullptr; int reportOn = 0;
	Scalar* state = nullptr; size_t stateCount = 0; unsigned restoreGrid = 0;
};
struct BatchEntry
{
	DeviceGraph g; DeviceStructure st; DeviceSystem sys;
	int maxIter = 0;
	int gridSpmv = 0;              // workgroups of this graph's SpMV launch (the batched grid is the maximum over the graphs)
	BatchTrial t;
};
// the batched launches of one trial (every graph's kernels, arguments and order as launch_linearize_dm / launch_pcg_setup_expand /
// launch_pcg2_fused(k = 0) / launch_trial_tail_fused + launch_restore_if_rejected issue them for one graph)
void batch_fill_linearize(const DeviceGraph& g, const DeviceStructure& st, const Scalar* backupSrc, Scalar* backupDst, size_t backupCount, BatchTrial& t);
void launch_batch_linearize(const BatchEntry* tab, int n, unsigned lmGridMax, unsigned schurGridMax, bool mixed, hipStream_t s);
void batch_fill_setup(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, const Scalar* copySrc, Scalar* copyDst, size_t copyCount, BatchTrial& t);
void launch_batch_setup(const BatchEntry* tab, int n, unsigned gridMax, hipStream_t s);
void launch_batch_first_precond(const BatchEntry* tab, int n, const DeviceSystem& sys0, int ncMax, size_t ldsMax, Scalar tol2, hipStream_t s);
void batch_fill_tail(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, const Scalar* old, const LmDevice& lm, Scalar* state, size_t stateCount, BatchTrial& t);
void launch_batch_tail(const BatchEntry* tab, int n, unsigned tailGridMax, unsigned restoreGridMax, hipStream_t s);
int batch_kernel_class(const DeviceGraph& g, const DeviceSystem& sys);       // graphs of one batch must agree on it; -1 = not batchable
size_t batch_pcg2_lds_bytes(const DeviceSystem& sys);
// one PCG iteration (chunk-local k) of all n graphs of the table: SpMV + fused two-level kernel, 2 launches
void launch_pcg_batch_iteration(const BatchEntry* tab, int n, const DeviceGraph& g0, const DeviceSystem& sys0, int gridSpmvMax, int ncMax, size_t ldsMax, int k, Scalar tol2, hipStream_t s);
// advance every graph's iteration offset by `iters`, stop test on the residual the chunk left, report to every graph's host block
void launch_pcg_batch_advance(const BatchEntry* tab, int n, int iters, hipStream_t s, Scalar tol2);

// Adds `chunk` PCG iterations (chunk-local k = 0..chunk-1) and the kbase advance to `graph` as a chain of kernel nodes.
// report != 0: the last node also copies the solver's flags and a ticket into the mapped host block (only the last graph of a
// batch needs that: the write to host memory costs several microseconds)
hipError_t graph_add_pcg_chunk(hipGraph_t graph, const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, int chunk, int maxIter, Scalar tol2, int report);

// hsc (damped, after pcg_setup) -> sys.hrow
void launch_hsc_expand(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, hipStream_t s);
// both in one launch; copySrc != nullptr: the same launch also copies copyCount (even) numbers copySrc -> copyDst (a freshly inverted
// coarse matrix into the buffer the iteration graphs read)
void launch_pcg_setup_expand(const DeviceGraph& g, const DeviceStructure& st, const DeviceSystem& sys, Scalar lambda, hipStream_t s,
	const Scalar* copySrc = nullptr, Scalar* copyDst = nullptr, size_t copyCount = 0);

// ---- exact reduced solve (ba_direct.hip): sparse tile Cholesky of the damped reduced matrix + triangular solves -------------------
// The free poses (internal order) are cut into segments of SC_TP consecutive poses = one 32 x 32 tile of the matrix (30 unknowns + 2
// identity-padded ones).  A minimum-degree elimination of the segment graph (multiple elimination: independent segments of near-minimal
// degree per round) gives the tile order, the fill and the elimination tree; columns of one tree LEVEL are independent, so the numeric
// phase is one launch per level (left-looking: every tile gathers the updates of its descendants in a fixed order -- no atomics, results
// reproducible bit for bit) and the two substitutions ride on the same levels.
constexpr int SC_T = 32;          // tile edge
constexpr int SC_TP = 5;          // poses per tile
constexpr int SC_TT = SC_T * SC_T;

struct SparseCholPlan             // host: the symbolic phase's result for one block pattern
{
	int Pf = 0, T = 0, nTiles = 0, nLevels = 0, slack = 0;
	std::vector<int> posOfSeg, segOfPos;     // segment <-> elimination position (= tile column)
	std::vector<int> colPtr;                 // [T + 1] tiles of column k: the diagonal tile first, then its rows in ascending position
	std::vector<int> rowIdx, colOfTile;      // [nTiles]
	std::vector<int> gPtr;                   // [nTiles + 1] gather list of every tile ...
	std::vector<int> gather;                 // ... 4 ints per entry {tile (i, k) or the zero tile, tile (j, k), k, 0}, k ascending, padded to a multiple of 4
	std::vector<int> wgRec;                  // [8 nTiles] per entry of lvlTiles: {tile, diagonal tile of its column, column, first gather entry, entries, 0, 0, 0}
	std::vector<int> lvlPtr, lvlTiles;       // [nLevels + 1], [nTiles]: tiles by the level of their column (work list of the factorisation)
	std::vector<int> lvlColPtr, lvlCols;     // [nLevels + 1], [T]: columns by level (work list of the backward substitution)
	std::vector<int> blkTile;                // [nblk] destination tile of every block of the upper-triangular BSR storage (bit 30: transposed)
	long long entries = 0;                   // gather entries (two tile products each at most)
	size_t tileBytes() const { return sizeof(Scalar) * (size_t)SC_TT * ((size_t)2 * nTiles + 1); }
	double secondsEstimate() const;          // what one factorisation + solve costs (a model: launches per level + tile traffic)
};
// slack < 0: automatic (the cheapest of a few multiple-elimination slacks by secondsEstimate).  false: more than maxTiles tiles of fill
bool sparse_chol_plan(int Pf, const int* rowptr, const int* colind, int slack, size_t maxTiles, SparseCholPlan& out);

struct SparseChol                 // device view
{
	Scalar* tiles = nullptr;      // [nTiles + 1][32 x 32] column-major: the matrix; the off-diagonal tiles become L; the last tile stays zero
	Scalar* tilesT = nullptr;     // [nTiles][32 x 32] the off-diagonal tiles of L once more, transposed (what the backward substitution reads); a diagonal
	                              // tile's slot: L_jj mirrored into both triangles
	Scalar* y = nullptr;          // [32 T] right-hand side -> L^-1 b -> solution, in elimination order
	Scalar* rinv = nullptr;       // [32 T] 1 / L_cc
	const int *colPtr = nullptr, *rowIdx = nullptr, *colOfTile = nullptr, *gPtr = nullptr, *gather = nullptr;
	const int *wgRec = nullptr, *lvlCols = nullptr, *blkTile = nullptr, *posOfSeg = nullptr;
	int* fail = nullptr;          // != 0 after the solve: a non-positive pivot was met (the matrix is not positive definite)
	int T = 0, Pf = 0, nTiles = 0;
};
// hsc (damped by launch_pcg_setup: full diagonal blocks) and bsc -> d.tiles / d.y; clears d.fail
void launch_sparse_chol_fill(const DeviceStructure& st, const DeviceSystem& sys, const SparseChol& d, hipStream_t s);
// factorise and solve: x[0 .. 6 Pf) = A^-1 b (internal pose order)
void launch_sparse_chol_solve(const SparseChol& d, const SparseCholPlan& plan, Scalar* x, hipStream_t s);

// out3 = {chi2 total, landmark scale part, pose scale part} gathered from the result slots of the kernels enqueued before
void launch_collect_eval(const DeviceSystem& sys, Scalar* out3, hipStream_t s);

// copies {fail, iters, done} into sys.host_flags (what the last node of an iteration graph does anyway)
void launch_pcg_report(const DeviceSystem& sys, hipStream_t s);

}  // namespace cubahip
