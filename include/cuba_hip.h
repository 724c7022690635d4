/*
 * cuba_hip.h -- C ABI of the MI355X (gfx950) bundle-adjustment hot path.
 *
 * This is the drop-in boundary: everything the reference's host layer (class CudaBlockSolver,
 * /root/reference/src/cuda_bundle_adjustment.cpp:73-673) asks of its device layer -- the 27 functions
 * of namespace cuba::gpu (src/cuda_block_solver.h:29-94) plus SparseLinearSolver::{initialize,solve}
 * (src/cuda_linear_solver.h:28-39) -- is reachable through the entry points below, one solver handle
 * per graph.  Plain pointers and sizes only; no C++/torch types.  Each entry point names the reference
 * interface it replaces.
 *
 * Conventions
 *   - every call returns a cuba_hip_status (0 = ok); nothing throws across the ABI;
 *     cuba_hip_last_error() gives the message of the last failure on that handle.
 *   - all array arguments are HOST pointers, caller-owned, only read/written during the call.
 *   - Scalar is fp64.  Quaternions are (x,y,z,w), poses are world->camera, small matrices
 *     column-major (as in the reference, src/cuda_block_solver.cu:79-105).
 *   - vertices arrive in "solver order": free poses first [0,Pf), then fixed [Pf,Pt); same for
 *     landmarks (that is what CudaBlockSolver::initialize produces, :142-200).  Edges may come in
 *     any order; per-edge outputs are returned in the caller's order.
 *   - a handle is bound to one device and one HIP stream; it is not re-entrant, distinct handles
 *     may be driven from distinct host threads.
 */
#ifndef CUBA_HIP_H_
#define CUBA_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cuba_hip_solver cuba_hip_solver;

typedef enum cuba_hip_status
{
	CUBA_HIP_OK = 0,
	CUBA_HIP_ERR_INVALID_ARGUMENT = 1,
	CUBA_HIP_ERR_RUNTIME = 2,       /* a HIP runtime call failed (reference: CUDA_CHECK only prints, src/macro.h:22-27) */
	CUBA_HIP_ERR_STATE = 3,         /* call order violated (e.g. solve before set_graph) */
	CUBA_HIP_ERR_NO_DEVICE = 4
} cuba_hip_status;

/* robust kernel kinds / edge types: include/cuda_bundle_adjustment_types.h:143-148,213-218 */
enum { CUBA_HIP_ROBUST_NONE = 0, CUBA_HIP_ROBUST_HUBER = 1, CUBA_HIP_ROBUST_TUKEY = 2 };
enum { CUBA_HIP_EDGE_MONOCULAR = 0, CUBA_HIP_EDGE_STEREO = 1 };

/* profile buckets, same order and meaning as CudaBlockSolver::ProfileItem (src/cuda_bundle_adjustment.cpp:77-88).
   Bucket 5 ("symbolic decomposition") holds the reduced-system structure analysis, bucket 6
   ("numerical decomposition") the block-PCG solve that replaces cuSOLVER's factor+solve. */
enum { CUBA_HIP_PROFILE_ITEMS = 8 };

/* arrays retrievable with cuba_hip_get_array (introspection for parity tests) */
enum
{
	CUBA_HIP_ARRAY_BP = 0,      /* 6*Pf   gradient-side vector bp                  (d_bp_)      */
	CUBA_HIP_ARRAY_BSC = 1,     /* 6*Pf   reduced right-hand side                   (d_bsc_)     */
	CUBA_HIP_ARRAY_XP = 2,      /* 6*Pf   pose increments                           (d_xp_)      */
	CUBA_HIP_ARRAY_XL = 3,      /* 3*Lf   landmark increments                       (d_xl_)      */
	CUBA_HIP_ARRAY_LM_SYS = 4,  /* 9*Lf   per landmark: 6 unique entries (00,01,02,11,12,22) of Hll
	                                       (after cuba_hip_max_diagonal) or of inv(Hll+lambda I)
	                                       (after cuba_hip_schur), then bl (3)                   */
	CUBA_HIP_ARRAY_HSC = 5,     /* 36*nblk upper-triangular BSR values of Hsc (col-major 6x6);
	                                       after cuba_hip_max_diagonal the diagonal blocks hold Hpp */
	CUBA_HIP_ARRAY_STATE = 6    /* 7*Pt+3*Lt  [q | t | Xw] estimates (cuba_hip_device_pointer only)          */
};

/* ---- lifetime -------------------------------------------------------------------------------- */

/* Replaces: CudaBundleAdjustment::create() -> CudaBlockSolver construction (src/cuda_bundle_adjustment.cpp:905-908).
   Threads: a handle is not re-entrant; distinct handles may be driven from distinct host threads at once (the reference's
   CudaBundleAdjustment objects are independent too, include/cuda_bundle_adjustment.h:34-125).  Each handle owns one work stream, one
   low-priority side stream (coarse inversions) and one upload stream.  hipGraphs are OPT-IN (option "pcg_graph" = 1 or CUBA_HIP_GRAPHS=1
   in the environment; the handle then also owns a helper thread that calls hipGraphInstantiate / hipGraphExecDestroy in the background --
   relevant for callers that fork, or that capture with hipStreamCaptureModeGlobal): on this runtime a process that has ever instantiated a
   hipGraph no longer overlaps the kernel chains of two streams (two KITTI-00 graphs from two threads: 1.5 x one graph's throughput
   without graphs, 1.0 x with), while a lone handle is only ~2 % faster with them.  With graphs on, the library
   still stops using them while more than one handle is alive. */
int cuba_hip_create(int device, cuba_hip_solver** out);
int cuba_hip_destroy(cuba_hip_solver* s);
const char* cuba_hip_last_error(const cuba_hip_solver* s);
const char* cuba_hip_version(void);
/* 8 for libcuba_hip.so, 4 for libcuba_hip_f32.so (the reference's USE_FLOAT32 build option, src/scalar.h:25-29):
   element size of the device arrays behind cuba_hip_device_pointer / cuba_hip_reduction_buffer.  Host-side
   arguments of every entry point are double in both builds. */
int cuba_hip_scalar_size(void);

/* Page-locked host memory for the caller's staging arrays (what it passes to cuba_hip_set_graph / receives from
   cuba_hip_get_solution / cuba_hip_chi_squares): transfers from it run at full PCIe rate and asynchronously.  Falls back to malloc
   when no device is visible; cuba_hip_host_free releases either kind.  Optional -- any host pointer is accepted everywhere. */
void* cuba_hip_host_alloc(size_t bytes);
void cuba_hip_host_free(void* p);

/* Run on an existing hipStream_t (e.g. torch's current stream) instead of the handle's private one. */
int cuba_hip_set_stream(cuba_hip_solver* s, void* hip_stream);

/* Options (20).  Solver: "pcg_tol" (relative preconditioned-residual tolerance of the reduced solve, default 1e-7; fp32 build 1e-4),
   "pcg_max_iter" (default 4*6*Pf capped at 32768), "pcg_accept_unconverged" (default 0, see cuba_hip_get_pcg_history),
   "direct_fallback" (default 1: a reduced solve whose PCG uses up its iteration budget, breaks down, or follows such a solve in the
   same Levenberg-Marquardt run is solved EXACTLY on the device -- sparse tile Cholesky, csrc/ba_direct.hip: minimum-degree ordering
   and symbolic analysis once per structure at the first exact solve, numeric factorisation + triangular solves per solve, the three
   phases of SparseLinearSolver, /root/reference/src/cuda_linear_solver.cpp:278-348, 147-232, 386-415 -- and fails only on a
   non-positive pivot like the reference, :406-410; 0 = the solve is reported as failed instead), "reduced_solver" (default 0 = block
   PCG with that fallback; 1 = EVERY reduced solve is the exact one, the reference's behaviour), "direct_after" (PCG iterations before
   the hand-over; default 0 = automatic: Pf / 4 clamped to 128 ... 384 -- about twice what an exact solve costs --, at most
   pcg_max_iter; a function of the graph's size, so that the decision never depends on timing), "direct_max_tiles" (default 2^20: the fill of the factor, in 32 x 32 tiles of 5 poses x 5 poses,
   beyond which the exact solver is not used and the failure report stands; a 10 000-pose trajectory with loop closures needs 19 000),
   "direct_slack" (default -1 = automatic: multiple-elimination slack of the ordering, 0 / 2 / 4 / 8 by a cost model of levels and fill),
   "pcg_aggregate" (poses per coarse aggregate of the two-level preconditioner; -1 = automatic: max(6, Pf/55) below 1320 free poses,
   max(16, Pf/min(180, max(115, Pf/32))) above; 0 = block-Jacobi only), "coarse_linear" (default 1: constant + linear-in-pose-index
   coarse functions per aggregate, 12 unknowns each; 0 = constant only, 6 unknowns), "precond_fp32" (fp64 library only, default 1: the
   explicit coarse inverse is STORED in fp32 -- symmetrised, applied with fp64 accumulation; a solve whose PCG breaks down with it is
   repeated with fp64 storage, which the handle then keeps; 0 = fp64 storage), "mixed_precision" (fp64 library only, default 0; 1 =
   per-edge linearisation records and the per-edge arithmetic of the pose / block Schur passes in fp32, every sum over edges, the
   reduced system and the PCG in fp64 -- the reference's USE_FLOAT32 idea, src/scalar.h:25-29, applied where it is second-order for
   the objective).
   The coarse matrix of a trial is assembled and inverted on a second, low-priority stream under that trial's PCG and serves from a
   later trial on (every trial up to a coarse dimension of 512, every second up to 1024, every third beyond); only the first solve
   on a structure inverts in line.
   "heuristics" (default 1: two run-to-run memories, exact when a run repeats the previous one and harmless otherwise -- the first
   solve of an LM run starts with the coarse inverse the first solve of the PREVIOUS run on this structure had, same damping regime,
   while its own inversion runs on the second stream; and a run that has repeated the previous run solve for solve so far sizes its
   next batch of iterations from that run instead of extrapolating; 0 = neither, what bench.py's "heuristics_off" prices).
   "spmv_upper" (default -1 = automatic: on beyond 1536 free poses, where the PCG kernels are bound by bytes; 1 / 0 = on / off: the
   PCG iteration as three launches straight from the upper-triangular BSR storage -- SpMV with the transposed products parked per
   block, row updates + P^T r per aggregate, preconditioner -- instead of two launches on a row-ordered copy of both triangles).
   "reduction_chunks" (default 0 = automatic; landmark partitions only: number of block-row ranges the reduced matrix is produced -- and,
   by the multi-GPU driver, summed -- in, see cuba_hip_schur_part),
   "landmark_reorder" (default 1, takes effect with the next cuba_hip_set_graph: the free landmarks are renumbered internally by (first, last)
   observing pose of the internal pose order, so that landmark-major data inherit the trajectory's locality whatever the caller's ids are;
   every host-pointer entry point keeps the caller's landmark numbering; a landmark partition other than the whole range, and the host
   pipeline, use the caller's order),
   cuba_hip_optimize takes the decision of every trial -- gain ratio, acceptance, next damping -- on the device and enqueues the next
   trial without having seen it, a rejected trial being undone by a conditional restore launch, and runs back-substitution, update and
   evaluation of a trial as ONE pass over the edges (control flow of /root/reference/src/cuda_bundle_adjustment.cpp:816-851; a
   landmark partition and "profile" use the host loop over the stage kernels instead).
   Execution: "pcg_graph" (default 0, or 1 when the environment has CUBA_HIP_GRAPHS=1: PCG iterations are replayed as hipGraphs of
   4 ... 256 iterations and of the exact batch lengths that come back; opt-in since round 6 -- a lone handle gains ~2 %, a process
   that has instantiated a hipGraph no longer overlaps the kernel chains of several handles; see cuba_hip_create),
   "device_setup" (default 1: the edge sort of
   cuba_hip_set_graph and the whole symbolic analysis of cuba_hip_build_structure run on the GPU; 0 = the host pipeline, the
   independent cross-check of the tests), "pose_reorder" (default 1: when most blocks of the reduced matrix lie far off its diagonal
   in the caller's pose numbering -- arbitrary vertex ids --, the free poses are renumbered internally along the trajectory found by
   a strongest-neighbour walk over the co-visibility counts, which the aggregates of the preconditioner need; every host-pointer
   entry point keeps the caller's numbering, only cuba_hip_device_pointer / cuba_hip_reduction_buffer expose the internal one),
   "profile" (0/1: per-stage synchronising wall-clock like the reference's get_time_point(), src/cuda_bundle_adjustment.cpp:43-47).
   Variants that were built, measured slower and removed again (first-generation Schur kernel with fp64 atomics, single-launch
   PCG iteration, speculative trial tail, in-line coarse refresh policy, CU-masked side stream, forcing term on the PCG tolerance)
   are documented with their numbers in DESIGN.md section 4 and profiles/. */
int cuba_hip_set_option(cuba_hip_solver* s, const char* key, double value);

/* ---- graph upload ---------------------------------------------------------------------------- */

/* Replaces: the SoA gather + uploads of CudaBlockSolver::initialize / buildStructure
   (src/cuda_bundle_adjustment.cpp:115-261, :319-354).
   q[4*Pt], t[3*Pt], cam[5*Pt] = (fx,fy,cx,cy,bf) per pose, Xw[3*Lt];
   edge_pose[E], edge_landmark[E] solver indices; edge_dim[E] in {2,3};
   meas[3*E] (third component ignored for monocular edges); omega[E] scalar information.
   Edges whose two ends are both fixed must already be removed (as the reference does, :204-243). */
int cuba_hip_set_graph(cuba_hip_solver* s, int Pt, int Pf, int Lt, int Lf,
	const double* q, const double* t, const double* cam, const double* Xw,
	int E, const int32_t* edge_pose, const int32_t* edge_landmark, const uint8_t* edge_dim,
	const double* meas, const double* omega);

/* The same upload in two steps, for a caller whose meas / omega arrays stay valid (and unchanged) until cuba_hip_set_graph_end: _begin
   returns once every other argument has been consumed, while the 32 bytes per edge of measurements and information may still be
   crossing PCIe on a second stream; the caller then typically calls cuba_hip_build_structure -- the symbolic analysis needs the index
   arrays only, so the transfer hides under it (0.7 ms of a 1.2 ms analysis at KITTI-00 size) -- and cuba_hip_set_graph_end, which
   waits for the transfer.  Any entry point that needs the values finishes the transfer itself; page-locked arrays
   (cuba_hip_host_alloc) are what makes the transfer asynchronous.  A cuba_hip_hint_unchanged promise covers _begin like set_graph. */
int cuba_hip_set_graph_begin(cuba_hip_solver* s, int Pt, int Pf, int Lt, int Lf,
	const double* q, const double* t, const double* cam, const double* Xw,
	int E, const int32_t* edge_pose, const int32_t* edge_landmark, const uint8_t* edge_dim,
	const double* meas, const double* omega);
int cuba_hip_set_graph_end(cuba_hip_solver* s);

/* A promise about the NEXT cuba_hip_set_graph call only (it is consumed by that call, successful or not):
   same_edges  != 0: edge_pose / edge_landmark / edge_dim and all five counts are identical to those of the previous successful call
                     (the library then skips its own comparison of the index arrays);
   same_values != 0: in addition meas and omega are bit-identical to the previous call's, so their 32 bytes per edge need not cross
                     PCIe again (the estimates q / t / cam / Xw are uploaded in any case).
   A host layer that re-walks its own graph anyway (the reference re-reads every edge in initialize(), src/cuda_bundle_adjustment.cpp:
   204-243) learns both facts for free while it copies; a caller that cannot vouch for them simply does not call this.  A wrong
   promise is not detected: the solver then works on the previous call's edges / values. */
int cuba_hip_hint_unchanged(cuba_hip_solver* s, int same_edges, int same_values);

/* Replaces: CudaBundleAdjustment::setRobustKernels (src/cuda_bundle_adjustment.cpp:781-784). */
int cuba_hip_set_robust_kernel(cuba_hip_solver* s, int edge_type, int kind, double delta);

/* Replaces: CudaBlockSolver::buildStructure (:263-366): gpu::buildHplStructure,
   HschurSparseBlockMatrix::constructFromVertices, gpu::findHschureMulBlockIndices and
   SparseLinearSolver::initialize (symbolic analysis).  Called implicitly if omitted. */
int cuba_hip_build_structure(cuba_hip_solver* s);

/* ---- per-iteration stages (same granularity as CudaBlockSolver's public methods) ---------------- */

/* Replaces: CudaBlockSolver::computeErrors -> gpu::computeActiveErrors x2 (:368-382). Total robust chi2. */
int cuba_hip_compute_errors(cuba_hip_solver* s, double* chi2);

/* Replaces: CudaBlockSolver::buildSystem -> gpu::constructQuadraticForm x2 (:384-410).
   Marks the current estimate as the linearisation point; the Jacobian products themselves are
   (re)computed inside cuba_hip_max_diagonal / cuba_hip_schur, nothing is materialised per edge. */
int cuba_hip_build_system(cuba_hip_solver* s);

/* Replaces: CudaBlockSolver::maxDiagonal -> gpu::maxDiagonal x2 (:412-418). */
int cuba_hip_max_diagonal(cuba_hip_solver* s, double* max_diag);

/* Replaces: CudaBlockSolver::setLambda / restoreDiagonal -> gpu::addLambda / gpu::restoreDiagonal
   (:420-430).  lambda is applied on the fly, so restore only forgets it. */
int cuba_hip_set_lambda(cuba_hip_solver* s, double lambda);
int cuba_hip_restore_diagonal(cuba_hip_solver* s);

/* CudaBlockSolver::solve (:432-481) in three separately callable parts (so a landmark-partitioned
   multi-GPU driver can all-reduce between them), and as one call:
     schur           gpu::computeBschure + gpu::computeHschure                        (:443-444)
     solve_reduced   gpu::convertHschureBSRToCSR + SparseLinearSolver::solve          (:452-453),
                     here a block-Jacobi preconditioned block PCG on the BSR matrix
     back_substitute gpu::schurComplementPost                                          (:463)
   *ok = 0 reports a numerical failure (non-SPD pivot / PCG breakdown), the reference's
   "factorize failed" path (src/cuda_linear_solver.cpp:406-410). */
int cuba_hip_schur(cuba_hip_solver* s);
int cuba_hip_solve_reduced(cuba_hip_solver* s, int* ok);
int cuba_hip_back_substitute(cuba_hip_solver* s);
int cuba_hip_solve(cuba_hip_solver* s, int* ok);

/* Replaces: CudaBlockSolver::update -> gpu::updatePoses + gpu::updateLandmarks (:483-492). */
int cuba_hip_update(cuba_hip_solver* s);

/* Replaces: CudaBlockSolver::computeScale -> gpu::computeScale (:494-500). sum x (lambda x + b). */
int cuba_hip_compute_scale(cuba_hip_solver* s, double lambda, double* scale);

/* Replaces: CudaBlockSolver::push / pop (:502-510). */
int cuba_hip_push(cuba_hip_solver* s);
int cuba_hip_pop(cuba_hip_solver* s);

/* A second, caller-controlled copy of the estimates on the device (push / pop above are the LM loop's own and are overwritten by
   every trial of cuba_hip_optimize): snapshot keeps [q | t | Xw] as they are now, restore brings them back with one device-to-device
   copy on the handle's stream -- repeated runs from one starting point (benchmarks, parameter sweeps) need no host round trip.
   No counterpart in the reference, whose callers re-upload through initialize(). */
int cuba_hip_snapshot_state(cuba_hip_solver* s);      /* = slot 0 */
int cuba_hip_restore_state(cuba_hip_solver* s);
/* The same with several copies (slot in [0, CUBA_HIP_SNAPSHOT_SLOTS)): a caller that re-optimises from a SET of starting points --
   a benchmark whose timed runs must not replay one another, a sweep over perturbations -- prepares them once and switches between
   them with one device-to-device copy each.  Copies are dropped by cuba_hip_set_graph and whenever the library changes its
   internal pose order ("pose_reorder"): restoring an absent slot is CUBA_HIP_ERR_STATE, never a silent mix-up of pose rows. */
enum { CUBA_HIP_SNAPSHOT_SLOTS = 64 };
int cuba_hip_snapshot_state_slot(cuba_hip_solver* s, int slot);
int cuba_hip_restore_state_slot(cuba_hip_solver* s, int slot);

/* ---- whole Levenberg-Marquardt run -------------------------------------------------------------- */

/* Replaces: the loop of CudaBundleAdjustmentImpl::optimize (src/cuda_bundle_adjustment.cpp:793-857)
   with identical control flow (tau = 1e-5, at most 10 trials per iteration, g2o's rho / lambda rules),
   executed inside the library to avoid per-stage host round trips.
   chi2_per_iter[niterations] receives BatchInfo::chi2 for each executed iteration, *n_done their count. */
int cuba_hip_optimize(cuba_hip_solver* s, int niterations, double* chi2_per_iter, int* n_done);

/* The same for SEVERAL handles at once, in one launch chain.  Replaces: independent CudaBundleAdjustment objects optimised side by
   side (include/cuda_bundle_adjustment.h:34-125 -- ORB-SLAM's local BA windows, BASELINE configs[3]'s independent graphs when they share
   a GPU).  Every handle runs its own Levenberg-Marquardt loop exactly as cuba_hip_optimize would -- its own damping, acceptance
   decisions (taken on the device, per graph), iteration counts and stopping -- and everything of a trial but the PCG iterations on its
   own stream; the PCG iterations of all graphs are issued as ONE chain of batched launches (the graph number is the launch grid's second
   dimension, the kernels' arguments come from a device table), so the host pays 2 launches per iteration for the batch instead of 2
   per graph.  Results are bit-identical to cuba_hip_optimize on each handle alone.  Graphs of one size class batch together (two-launch
   iteration with a two-level preconditioner, the same coarse-dimension and occupancy class, same "pcg_tol", same device, distinct
   streams, no "profile", no landmark partition); anything else is run one handle after the other, with the same results.
   handles[n] (distinct, n <= 64), chi2_per_iter[n * niterations] (row i = handle i; may be NULL), n_done[n];
   *batched_solves (optional) = reduced solves that ran as a batch (0: the fallback ran).  The calling thread drives all handles: do not
   use them from other threads meanwhile.  Errors are recorded on handles[0]. */
int cuba_hip_optimize_batch(cuba_hip_solver** handles, int n, int niterations, double* chi2_per_iter, int* n_done, int* batched_solves);

/* ---- results ------------------------------------------------------------------------------------ */

/* Replaces: CudaBlockSolver::finalize downloads (:512-526). */
int cuba_hip_get_solution(cuba_hip_solver* s, double* q, double* t, double* Xw);
int cuba_hip_set_solution(cuba_hip_solver* s, const double* q, const double* t, const double* Xw);

/* Replaces: CudaBlockSolver::getChiSqs -> gpu::computeChiSquares x2 (:528-543).
   Non-robust omega*|r|^2 per edge, in the caller's edge order. */
int cuba_hip_chi_squares(cuba_hip_solver* s, double* chi2_per_edge);
/* The same in two steps, for a caller that has host work of its own to do meanwhile (the C++ layer copies the estimates back into the
   caller's vertices): _begin enqueues the evaluation and the copy into chi2_per_edge (page-locked memory from cuba_hip_host_alloc, or the
   copy is not asynchronous) and returns; cuba_hip_chi_squares_end waits for it.  No other call on the handle in between. */
int cuba_hip_chi_squares_begin(cuba_hip_solver* s, double* chi2_per_edge);
int cuba_hip_chi_squares_end(cuba_hip_solver* s);

/* Replaces: CudaBlockSolver::getTimeProfile (:545-562). Seconds per bucket. */
int cuba_hip_get_profile(cuba_hip_solver* s, double seconds[CUBA_HIP_PROFILE_ITEMS]);

/* Counters of the last optimize / solve: [0] PCG iterations (total), [1] LM trials (total),
   [2] number of 6x6 blocks in upper-triangular Hsc, [3] number of Schur block products (nmul),
   [4] coarse-inverse refreshes of the two-level preconditioner, [5] host looks at the device stop flag,
   [6] PCG iterations enqueued (>= [0]: launches after convergence return at once), [7] dimension of the coarse
   system of the two-level preconditioner. */
int cuba_hip_get_counters(cuba_hip_solver* s, int64_t counters[8]);
/* One counter by name (totals since cuba_hip_set_graph): the eight above as "pcg_iterations", "lm_trials", "coarse_refreshes",
   "pcg_host_looks", "pcg_iterations_enqueued", plus "coarse_inline_inversions" (coarse inversions that ran on the work stream in front
   of a solve -- the others ran on the second stream under an earlier trial's PCG), "pcg_unconverged_solves",
   "pcg_graph_instantiations" (hipGraphs of PCG iteration batches built), "precond_fp32_fallbacks" (solves repeated with the fp64
   coarse inverse after the fp32-stored one broke the PCG down), "exact_solve_fallbacks" (reduced solves that went to the exact
   solver, see "direct_fallback"; 0 on well-conditioned graphs), "exact_solve_failures" (of these, the ones that met a non-positive pivot), "host_looks" (times the host waited for a device report:
   PCG batches + the LM decisions it had to see), "pcg_iterations_plain_launches" (iterations enqueued as plain launches although graphs
   are on: graph not instantiated yet), "graph_uploads" (successful cuba_hip_set_graph[_begin] calls in the
   life of the handle, never reset: a caller that keeps state about "what the device holds" -- the promises of cuba_hip_hint_unchanged --
   stores this number with it and distrusts its state when the two differ), "value_bytes_uploaded" (bytes of measurements and
   information that crossed PCIe in the life of the handle: 32 per edge and full upload, 36 per OWNED edge for cuba_hip_set_graph_partition),
   "late_decision_records" (life of the handle: LM decision records that had not reached host memory when the report behind them had, and
   were fetched by a stream synchronisation instead; 0 in normal operation).
   Unknown name: CUBA_HIP_ERR_INVALID_ARGUMENT. */
int cuba_hip_get_counter(cuba_hip_solver* s, const char* name, int64_t* value);

/* PCG iteration count of every reduced solve since cuba_hip_set_graph, oldest first (at most `capacity` are written,
   *n_solves receives their number); a NEGATIVE entry is a solve that stopped at pcg_max_iter with the stop test
   unsatisfied, *n_unconverged counts those.  Such a solve is reported as a failure (*ok = 0 from
   cuba_hip_solve_reduced / cuba_hip_solve; cuba_hip_optimize rejects the trial and raises lambda) -- the role of
   the reference's "factorize failed" path, src/cuda_linear_solver.cpp:406-410 -- unless the option
   "pcg_accept_unconverged" = 1 asks for the best iterate to be used as an inexact step, or -- the default, "direct_fallback" = 1 --
   the exact solver takes the solve over (the entry then records the iterations the PCG had used; solves that went to the exact solver
   straight away are recorded as 0). */
int cuba_hip_get_pcg_history(cuba_hip_solver* s, int32_t* iterations, int capacity, int* n_solves, int64_t* n_unconverged);

/* ---- introspection (parity tests) and multi-GPU plumbing ------------------------------------------ */

/* Structure of the reduced system: upper-triangular BSR (replaces the accessors of
   HschurSparseBlockMatrix, src/sparse_block_matrix.h:81-101). row_ptr[Pf+1], col_ind[nblk]. */
int cuba_hip_get_hsc_structure(cuba_hip_solver* s, int32_t* row_ptr, int32_t* col_ind, int* nblk);

/* Copy an internal device array to the host; out may be NULL to query *count only. */
int cuba_hip_get_array(cuba_hip_solver* s, int which, double* out, size_t* count);

/* Measurement hook for bench.py: average device milliseconds per launch, taken with HIP events on the
   handle's stream over `reps` back-to-back launches, of
     [0] residual_chi2  [1] linearize+Schur  [2] pcg_spmv  [3] pcg_update (+restrict)  [4] back_substitute
     [5] reserved (0)
     [6] coarse_setup (assemble + invert the coarse matrix, once per solve; 0 if disabled).
   Clobbers the increments and the reduced system (not the estimates). */
enum { CUBA_HIP_TIMED_KERNELS = 7 };
int cuba_hip_time_kernels(cuba_hip_solver* s, int reps, double ms_per_launch[CUBA_HIP_TIMED_KERNELS]);

/* ---- landmark-partitioned multi-GPU operation (no counterpart in the single-GPU reference) ----------------
   Every rank uploads the WHOLE graph (so all ranks share one Hsc pattern and one pose ordering) and then
   restricts itself to the landmarks [landmark_begin, landmark_end) of the solver order: residuals, Schur
   contributions, back-substitution and landmark updates are evaluated for those landmarks only.
   Per trial the driver sums cuba_hip_reduction_buffer() over the ranks between cuba_hip_schur and
   cuba_hip_solve_reduced; chi2 and the landmark parts of max-diagonal / scale are reduced as scalars. */
/* (landmark_begin, landmark_end) = (0, -1) removes the restriction again. */
int cuba_hip_set_partition(cuba_hip_solver* s, int landmark_begin, int landmark_end);
/* cuba_hip_set_graph and cuba_hip_set_partition in one call, for a rank that knows its range when it uploads: the index arrays and the
   estimates go up whole (the block pattern and the pose order are global), but meas / omega are READ ONLY for the edges whose landmark lies
   in [landmark_begin, landmark_end) -- 36 bytes per owned edge cross PCIe instead of 32 per edge of the graph, i.e. about 1 / N of the
   value arrays on each of N ranks -- and need not hold anything meaningful elsewhere.  The other edges' values read as zeros on the
   device; no stage of a partitioned handle looks at them (cuba_hip_chi_squares reports 0 for them).  Until the next upload the handle
   therefore refuses a cuba_hip_set_partition beyond this range (CUBA_HIP_ERR_STATE; lifting the restriction included) and ignores a
   "same values" promise (cuba_hip_hint_unchanged).  With the option "device_setup" = 0 the whole arrays are read. */
int cuba_hip_set_graph_partition(cuba_hip_solver* s, int Pt, int Pf, int Lt, int Lf,
	const double* q, const double* t, const double* cam, const double* Xw,
	int E, const int32_t* edge_pose, const int32_t* edge_landmark, const uint8_t* edge_dim,
	const double* meas, const double* omega, int landmark_begin, int landmark_end);
/* First half of cuba_hip_max_diagonal: accumulate Hpp (diagonal blocks of the reduction buffer), bp, Hll. */
int cuba_hip_assemble(cuba_hip_solver* s);
/* Second half: max diag of the (externally summed) Hpp and of this rank's Hll. */
int cuba_hip_max_diagonal_parts(cuba_hip_solver* s, double* pose_part, double* landmark_part);
/* sum x (lambda x + b) split into the replicated pose part and this rank's landmark part. */
int cuba_hip_compute_scale_parts(cuba_hip_solver* s, double lambda, double* pose_part, double* landmark_part);
/* Device address of an internal array (ids as for cuba_hip_get_array) for zero-copy collectives. */
int cuba_hip_device_pointer(cuba_hip_solver* s, int which, void** device_ptr, size_t* count);

/* {Pt, Pf, Lt, Lf, E} of the uploaded graph. */
int cuba_hip_get_sizes(cuba_hip_solver* s, int sizes[5]);

/* Test hook for the one dense kernel of the path: the blocked Gauss-Jordan inversion (matrix-core tile products) that builds the
   coarse inverse of the two-level preconditioner.  A, Ainv: n x n, column-major, SPD input.  No solver handle involved. */
int cuba_hip_debug_dense_inverse(int device, int n, const double* A, double* Ainv);

/* Test hooks for the exact reduced solve (csrc/ba_direct.hip: sparse tile Cholesky on the matrix cores + triangular solves), the
   counterpart of SparseLinearSolver::initialize / solve (/root/reference/src/cuda_linear_solver.cpp:278-348, 386-415).  A: n x n
   column-major, symmetric (n a multiple of 6: it is cut into the 6 x 6 blocks of a reduced system; off-diagonal blocks that are
   identically zero are not part of the pattern); x = A^-1 b.  *not_positive_definite != 0 reports a non-positive pivot (ref :406-410).
   slack: multiple-elimination slack of the ordering (-1 = automatic); stats (optional) = {tile columns, tiles, levels, slack used}.
   No solver handle involved.  cuba_hip_debug_dense_solve = the same with slack -1 and no statistics. */
int cuba_hip_debug_dense_solve(int device, int n, const double* A, const double* b, double* x, int* not_positive_definite);
int cuba_hip_debug_sparse_solve(int device, int n, const double* A, const double* b, double* x, int* not_positive_definite, int slack, int32_t stats[4]);
/* The symbolic phase alone, on the host (no device needed): ordering, fill, elimination-tree levels and gather lists for the
   upper-triangular block pattern (row_ptr[n_poses + 1], col_ind; diagonal block first in every row).  which: 0 header {tile columns,
   tiles, levels, slack, gather entries, blocks}, 1 posOfSeg, 2 colPtr, 3 rowIdx, 4 gPtr, 5 gather (4 ints per entry), 6 lvlPtr,
   7 lvlTiles, 8 lvlColPtr, 9 lvlCols, 10 blkTile (see SparseCholPlan in csrc/ba_kernels.hpp).  *count = length of the array; out may be
   NULL to ask for it. */
int cuba_hip_debug_sparse_plan(int n_poses, const int32_t* row_ptr, const int32_t* col_ind, int slack, int which, int32_t* out, size_t capacity, size_t* count);

/* A driver that runs the Levenberg-Marquardt loop itself through the stage calls announces the start of a run (a new lambda_0):
   the coarse inverse of the two-level preconditioner and the iteration-count predictions of the previous run are dropped, as
   cuba_hip_optimize does at its start; a hand-over to the exact reduced solve that an earlier run made (option "direct_fallback")
   is forgotten as well, so the new run starts with PCG again.  Optional (they would be refreshed after one slow solve anyway). */
int cuba_hip_begin_run(cuba_hip_solver* s);

/* The HIP stream the handle enqueues on (a collective library must order its operations with the solver's kernels). */
int cuba_hip_get_stream(cuba_hip_solver* s, void** hip_stream);

/* Enqueue (no host synchronisation) the evaluation of the current estimate over this handle's landmark range and leave
   three device scalars (element size cuba_hip_scalar_size()) at *device_scalars3:
     [0] robust chi2 of the local edges, [1] landmark part of sum x (lambda x + b) from the last cuba_hip_back_substitute,
     [2] pose part (only if with_scale != 0).  A multi-GPU driver sums [0..1] over the ranks in-stream. */
int cuba_hip_evaluate_device(cuba_hip_solver* s, double lambda, int with_scale, void** device_scalars3);

/* cuba_hip_schur in parts, for a driver that overlaps the sum over the ranks with the pass that produces it.  A landmark-partitioned
   handle cuts its reduced matrix at block rows into ranges of about equal size (option "reduction_chunks": 0 = automatic, one range
   per 8 MiB of matrix values and at most 8 -- a single part below 16 MiB --, n = at most n ranges; the cuts follow from the block
   pattern alone, so all ranks make the same ones) and runs the off-diagonal block pass range by range.  cuba_hip_schur_parts reports
   the number of parts (1 without a partition).  cuba_hip_schur_part(part) enqueues part 0 = linearisation, landmark pass, pose pass
   (all diagonal blocks, bsc, bp) and the off-diagonal blocks of range 0, or part c > 0 = the off-diagonal blocks of range c; parts
   must be run in the order 0 .. n - 1, all of them.  Once the work of part c is complete in stream order, the elements
   [ranges[0], ranges[0] + ranges[1]) of cuba_hip_reduction_buffer -- the blocks of range c -- and [ranges[2], ranges[2] + ranges[3])
   -- [bsc | bp] for part 0, empty otherwise -- hold this rank's final contribution.  The results are those of cuba_hip_schur bit for
   bit (which, on such a handle, simply runs all parts). */
int cuba_hip_schur_parts(cuba_hip_solver* s, int* n_parts);
int cuba_hip_schur_part(cuba_hip_solver* s, int part, size_t ranges[4]);

/* Device address + length (in doubles) of the contiguous buffer [Hsc values | bsc | bp] that a
   landmark-partitioned multi-GPU driver must sum across ranks between cuba_hip_schur and
   cuba_hip_solve_reduced (RCCL all-reduce over xGMI; no equivalent in the single-GPU reference). */
int cuba_hip_reduction_buffer(cuba_hip_solver* s, void** device_ptr, size_t* count);

#ifdef __cplusplus
}
#endif

#endif /* CUBA_HIP_H_ */
