/*
 * cuba_hip_dist.h -- C ABI of the native multi-GPU driver (libcuba_hip_dist.so): ONE graph, landmark-partitioned over
 * the GPUs of a node, one process (or host thread) per GPU, RCCL over xGMI for the exchange.
 *
 * The single-GPU reference has no counterpart of this file; the loop it runs is the reference's Levenberg-Marquardt
 * (CudaBundleAdjustmentImpl::optimize, /root/reference/src/cuda_bundle_adjustment.cpp:793-857) with the stage calls of
 * CudaBlockSolver (:368-510) replaced by the partition-aware entry points of include/cuba_hip.h.
 *
 * Per LM trial every rank
 *   1. linearises and Schur-reduces ITS landmarks        cuba_hip_schur                      (local)
 *   2. sums the reduction buffer [Hsc | bsc | bp]        ONE large all-reduce, in-stream     (xGMI) -- or, for a reduced matrix of
 *                                                        16 MiB and more (option "reduction_chunks" of the solver handle), one
 *                                                        all-reduce per block-row range, issued on a second stream as soon as the
 *                                                        range is complete, under the block pass of the next range (cuba_hip_schur_part)
 *   3. solves the reduced system                         cuba_hip_solve_reduced              (replicated: identical inputs and
 *                                                        fixed summation orders give bit-identical increments on every rank,
 *                                                        so nothing is broadcast)
 *   4. back-substitutes / updates its landmarks + poses  cuba_hip_back_substitute, _update   (local)
 *   5. evaluates its edges                               cuba_hip_evaluate_device            (local), then a 2-scalar all-reduce
 *      {chi2, landmark part of the gain-ratio denominator} in-stream and ONE read-back of three scalars.
 * No scalar ever makes a per-rank host round trip of its own; the host synchronises twice per trial (PCG stop flag,
 * evaluation), exactly as the single-GPU loop does.
 *
 * Every rank must have uploaded the WHOLE graph (cuba_hip_set_graph) into its solver handle before the driver is created;
 * the driver restricts the handle to [landmark_begin, landmark_end) (cuba_hip_set_partition).  The ranges of all ranks must
 * tile [0, Lt).  Status codes are those of cuba_hip.h.
 */
#ifndef CUBA_HIP_DIST_H_
#define CUBA_HIP_DIST_H_

#include <stddef.h>

#include "cuba_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cuba_hip_dist cuba_hip_dist;

/* Collective operations the driver needs, for communicators other than RCCL (tests: ranks as host threads on one GPU).
   Both are IN PLACE on a DEVICE buffer of `count` elements of `scalar_size` bytes (4 or 8) and must be ordered after the
   work already enqueued on `hip_stream` and before any work enqueued on it later.  Return 0 on success. */
typedef struct cuba_hip_comm_ops
{
	void* ctx;
	int (*allreduce_sum)(void* ctx, void* device_buf, size_t count, int scalar_size, void* hip_stream);
	int (*allreduce_max)(void* ctx, void* device_buf, size_t count, int scalar_size, void* hip_stream);
} cuba_hip_comm_ops;

/* 128-byte RCCL unique id (ncclGetUniqueId): rank 0 creates it, the launcher hands it to every rank. */
enum { CUBA_HIP_DIST_UNIQUE_ID_BYTES = 128 };
int cuba_hip_dist_unique_id(void* id128);

/* Driver over a NEW RCCL communicator (ncclCommInitRank with the given id; collective: every rank must call it). */
int cuba_hip_dist_create_rccl(cuba_hip_solver* s, const void* id128, int rank, int world,
	int landmark_begin, int landmark_end, cuba_hip_dist** out);
/* Driver over the caller's communicator (an ncclComm_t of `world` ranks, this process being `rank`); not destroyed with the driver. */
int cuba_hip_dist_attach_rccl(cuba_hip_solver* s, void* nccl_comm, int rank, int world,
	int landmark_begin, int landmark_end, cuba_hip_dist** out);
/* Driver over caller-supplied collectives. */
int cuba_hip_dist_create_custom(cuba_hip_solver* s, const cuba_hip_comm_ops* ops, int rank, int world,
	int landmark_begin, int landmark_end, cuba_hip_dist** out);

/* Levenberg-Marquardt over the partitioned graph; same contract as cuba_hip_optimize, identical results on every rank. */
int cuba_hip_dist_optimize(cuba_hip_dist* d, int niterations, double* chi2_per_iter, int* n_done);

/* Make every rank's estimates complete: each rank contributes the landmarks it owns (one all-reduce of the landmark
   part of the state); afterwards cuba_hip_get_solution returns the full solution on every rank. */
int cuba_hip_dist_complete_solution(cuba_hip_dist* d);

/* [0] large all-reduces, [1] small all-reduces, [2] elements moved by the large ones, [3] LM trials */
int cuba_hip_dist_get_counters(cuba_hip_dist* d, long long counters[4]);
/* Parts the per-trial sum of the reduction buffer is issued in (1 = one all-reduce behind the whole Schur pass), and the number of
   all-reduces so far that were issued while a later part of the pass was still to be enqueued (0 with one part). */
int cuba_hip_dist_reduction_parts(cuba_hip_dist* d, int* n_parts, long long* overlapped);

const char* cuba_hip_dist_last_error(const cuba_hip_dist* d);
int cuba_hip_dist_destroy(cuba_hip_dist* d);

#ifdef __cplusplus
}
#endif

#endif /* CUBA_HIP_DIST_H_ */
