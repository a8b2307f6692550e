/*
 * ipc_amd.h -- C ABI of the MI355X-native IPC consistency engine (libipc_amd.so).
 *
 * Drop-in boundary for the hot path of EmilioOlivastri/IPC: everything the reference does
 * between "here is the odometry chain + the loop-closure candidates" and "this candidate set
 * is consistent".  Each entry point names the reference interface it replaces (file:line in
 * the reference tree).  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions
 *   - every function returns 0 on success and a negative ipc_status on failure, never throws,
 *     never aborts; ipc_last_error() returns a human-readable message for the calling thread.
 *   - measurements / information matrices use the g2o text-file layout the reference loads
 *     (src/utils.cpp:114): SE2 "x y theta" + 6 upper-triangular information values;
 *     SE3 "x y z qx qy qz qw" + 21 upper-triangular values in (x y z qx qy qz) order.
 *   - vertex ids are 0..V-1 and odometry edge j joins vertex j -> j+1 (the reference's own
 *     contract, src/consensus.cpp:13-23).
 *   - pointers named d_* are DEVICE pointers (HBM of the engine's GPU), everything else is
 *     host memory.  `stream` is a hipStream_t passed as void* (NULL = the engine's own
 *     NON-BLOCKING stream, which is not ordered with the legacy default stream: pass an explicit
 *     stream when other work, e.g. a collective, must be ordered with the call).
 *   - a handle is bound to one GPU and must not be used from two threads at once (the
 *     reference's IPC object is not re-entrant either).
 *   - environment: the faithful mode keeps up to 16 solves in flight, one per HIP stream; streams that share a hardware
 *     queue run one after the other and the runtime's default is 4 queues.  The HOST PROGRAM exports
 *     GPU_MAX_HW_QUEUES=24 before its first HIP call for full speed (the library does not touch the environment: round 5;
 *     ipc_tester_2D/3D and bench.py do it in their main, INTEGRATION.md shows the line for the reference's testers).
 *     Without it the window is 4 solves and a probe measures how many streams really run abreast.  IPC_SPEC_STATS=1 prints how many
 *     of the engine's streams were measured to run side by side.  The order in which the pipeline starts its solves
 *     follows a prediction of each verdict (the candidate's own chi2 at the poses it starts from; IPC_SPEC_PREDICT,
 *     INTEGRATION.md): predictions schedule, no result depends on them.
 */
#ifndef IPC_AMD_H
#define IPC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ipc_engine ipc_engine_t;

typedef enum {
    IPC_OK = 0,
    IPC_ERR_ARG = -1,        /* bad argument / out-of-contract graph        */
    IPC_ERR_HIP = -2,        /* HIP runtime failure (message has the detail) */
    IPC_ERR_STATE = -3,      /* call order (e.g. no candidates set)          */
    IPC_ERR_LIMIT = -4       /* a size the engine cannot hold (ipc_set_max: N beyond the LDS-resident mask);
                                chains of any length are solved -- see ipc_solve_rows */
} ipc_status;

/* The knobs the path reads: struct Config fields used by IPC::IPC
 * (reference include/ipc/utils.hpp:22-38, src/consensus.cpp:18-32). */
typedef struct {
    double fast_reject_th;
    int    fast_reject_iter_base;
    double slow_reject_th;
    int    slow_reject_iter_base;
    double s_factor;
} ipc_params_t;

/* Per-cell diagnostics of the last ipc_solve_rows() (parity tests, profiling). */
typedef struct {
    int    i, j;             /* candidate indices (i == j: diagonal cell)   */
    int    lo, hi;           /* vertex-id span of the sub-problem           */
    double max_chi2;         /* max over the cell's edges of chi2           */
    double chi2_total;       /* sum of chi2 at exit                          */
    int    iterations;       /* dog-leg outer iterations executed            */
    int    tries;            /* total trial steps                            */
    int    flags;            /* bit0 Terminate, bit1 Fail                    */
    int    evals;            /* residual evaluations actually executed        */
} ipc_cell_info_t;

const char* ipc_last_error(void);

/* Replaces IPC<EDGE,VERTEX>::IPC (reference src/consensus.cpp:9-33): takes the odometry chain
 * (getProblemOdom + sort, :13-15), scales its information by s_factor (robustifyVoters, :21,
 * src/consensus_utils.cpp:124-130) and propagates the open-loop guess from vertex 0
 * (propagateGuess, :23, src/consensus_utils.cpp:99-116) -- all on the GPU `device`.
 * dim = 2 (SE2) or 3 (SE3). odom_meas [V-1][3|7], odom_info [V-1][6|21]. */
int ipc_create(int dim, int n_vertices, const double* odom_meas, const double* odom_info,
               const ipc_params_t* params, int device, ipc_engine_t** out);

/* Replaces IPC<EDGE,VERTEX>::~IPC (src/consensus.cpp:35-40). */
int ipc_destroy(ipc_engine_t* h);

/* The candidate list the harness feeds to agreementCheck one by one
 * (reference src/simulation.cpp:24-26,34-47): ids [N][2] (from,to), meas [N][3|7],
 * info [N][6|21], in FILE order; the engine applies the cmpTime processing order
 * (src/utils.cpp:379-390) with the (max id, index) tie-break. */
int ipc_set_candidates(ipc_engine_t* h, int n, const int* ids, const double* meas,
                       const double* info);

/* One more candidate at the END of the list (index N), without touching the incremental state: the consensus set
 * (candidate indices), the current poses and the solves the engine has in flight stay what they are.  For a harness that
 * meets its loop closures one by one (reference src/simulation.cpp:34-47 hands IPC::agreementCheck edges it has never
 * announced).  Cost: one record written in place by a one-thread kernel that carries it in its arguments -- no
 * re-upload, no allocation (the arrays grow geometrically, log2 N times over a run), no hipFree, no host
 * synchronisation.  *index_out = the new candidate's index. */
int ipc_append_candidate(ipc_engine_t* h, const int* ids, const double* meas, const double* info, int* index_out);

/* cmpTime processing order (host copy, N ints). */
int ipc_candidate_order(ipc_engine_t* h, int* order_out);

/* Open-loop poses after propagateGuess: SE2 [V][3] (x y theta), SE3 [V][12] (R row-major, t). */
int ipc_initial_poses(ipc_engine_t* h, double* poses_out);

/* ---- consistency matrix (the batched form of
 *      isAgreeingWithCurrentState, reference src/consensus_utils.cpp:7-22) ------------------ */

/* rows per rank of a shard: ceil(N / world). */
int ipc_rows_per_rank(int n, int world);

/* Which rank solves which row of the matrix and where the row sits in that rank's shard: slot_out[i] = owner * rpr +
 * index, rpr = ipc_rows_per_rank(n, world).  ids [n][2] as in ipc_set_candidates.  policy 0: row-cyclic (i % world);
 * policy 1 (the engine's default, IPC_ROW_BALANCE=cost): balanced by cost -- a row's cost is the number of poses its
 * cells sweep (sum over the overlapping pairs (i, j > i) of the union chain length + its own chain), rows go,
 * costliest first, to the least loaded rank with a free slot.  Pure host code (no GPU): every rank of a distributed
 * run computes the same map. */
int ipc_row_assignment(int n, const int* ids, int world, int policy, int* slot_out);

/* Solve every cell (i, j >= i) of the rows this rank owns (ipc_row_assignment with the engine's policy).  d_upper is a
 * device buffer of ipc_rows_per_rank(N, world) * ceil(N/64) uint64 words; the row of candidate i sits at index
 * slot[i] % rpr: bit j (j > i) = pair cell solved and consistent, bit i = diagonal cell consistent.  Non-overlapping
 * pairs are not solved (their bit stays 0; see ipc_assemble_matrix). */
int ipc_solve_rows(ipc_engine_t* h, int rank, int world, uint64_t* d_upper, void* stream);
/* Stream contract of ipc_solve_rows: the call enqueues on `stream` and on streams of the engine that
 * fork from / join back into it, and it blocks the HOST ONCE: behind the cell kernels, for the counts of the
 * cells to solve again (failed factorisations, IPC_LM_RETRY, and cells within IPC_BORDERLINE_BAND of their
 * threshold; skipped when both are off).  The borderline cells are then solved again on the device by the cell
 * kernels themselves (g2o's literal trial loop, compact per-bin lists built on the device, no copies through the
 * host); only cells whose linear solve failed -- degenerate information matrices -- go one by one through the
 * host-driven Levenberg solver.  The FIRST call after the candidates, the rank or the world changed blocks once
 * more, before the cell kernels: the planning pass's cell counts come back and buffers grow with hipMalloc; the
 * cell lists are then kept (round 5).  Whatever the faithful mode has in flight is given up first (it restarts with
 * the next ipc_agreement_check).  Cells whose chain
 * is longer than the largest cell kernel (SE3: 4096 poses, SE2: 16384 with the default policies) are
 * solved one at a time by the cluster solver of ipc_agreement_check -- correct for any length, host
 * driven and slow (reference cfg/3D/GRID_params.yaml: 8000 poses); ipc_solve_report() counts them. */

/* Counts over the cells of the last ipc_solve_rows() (blocks until the device is idle). */
typedef struct {
    int cells;               /* solved cells                                                   */
    int long_cells;          /* of those, solved by the cluster-solver fallback                */
    int failed_cells;        /* the optimisation ended in g2o's Fail state (flags & 2): the linear solve kept meeting
                                non-positive pivots up to the largest Levenberg damping (or the sub-problem has more than
                                24 000 unknowns AND no band structure, ipc_incremental_counters); the cell's chi2 is that
                                of the last good state                                                              */
    int capped_cells;        /* ran to the iteration cap without the dog-leg terminating       */
    int nan_cells;           /* max chi2 is NaN (counts as "agrees", like chi2 > th does in
                                the reference, src/consensus_utils.cpp:18)                     */
    int damped_cells;        /* cells whose linear solve met a non-positive pivot and were solved again with g2o's
                                Levenberg retry (lambda 1e-7 x 10 per failure up to 1e3, sticky for the rest of the
                                optimisation) on the literal normal equations                                      */
    int literal_cells;       /* cells whose max chi2 ended within IPC_BORDERLINE_BAND (default 4 sqrt(IPC_TERMINATE_EPS),
                                relative) of their threshold and were solved again by g2o's literal trial loop, so the
                                convergence test cannot have changed their decision                                 */
} ipc_solve_report_t;
int ipc_solve_report(ipc_engine_t* h, ipc_solve_report_t* out);

/* Build the full symmetric N x N bit matrix (row-major, ceil(N/64) words per row) from the
 * all-gathered shards d_gathered[world][rows_per_rank][words] (row i at gathered row slot[i]):  C[i][j] = solved bit when the
 * id intervals overlap with positive length (reference src/consensus.cpp:157-159), else
 * C[i][i] & C[j][j]. */
int ipc_assemble_matrix(ipc_engine_t* h, const uint64_t* d_gathered, int world,
                        uint64_t* d_bits, void* stream);

/* Greedy consistent-set maximisation over the matrix in cmpTime order (SURVEY.md 8a row P2;
 * plays the role of computeIndependentSubgraph + accept/reject, reference
 * src/consensus.cpp:43-75,124-171).  d_accepted: N bytes (1 = in the consensus set). */
int ipc_set_max(ipc_engine_t* h, const uint64_t* d_bits, uint8_t* d_accepted, void* stream);

/* Single-GPU convenience: solve + assemble + set-max, host outputs (any may be NULL):
 * bits_out [N][ceil(N/64)], accepted_out [N]. */
int ipc_run(ipc_engine_t* h, uint64_t* bits_out, uint8_t* accepted_out);

/* Set-only mode (one GPU): the accepted set of ipc_run without the cells the set-max never reads -- the diagonal cells
 * first, then the pair cells among the candidates whose own cell passed (SURVEY.md 8a row P2 only ever tests those).
 * Same accepted set as ipc_run by construction; no consistency matrix comes out of it (reported separately from the
 * candidate-pairs/s metric, which is defined over the full matrix).  *solved_cells_out: cells actually solved. */
int ipc_run_set_only(ipc_engine_t* h, uint8_t* accepted_out, int* solved_cells_out);

/* Matrix mode over several GPUs of one node from ONE process (what the reference's single-process testers need to use
 * more than one GPU; a multi-process run gathers with RCCL instead, ipc_amd/dist.py).  engines[r], r < n_engines, were
 * created on different devices with the same chain and given the same candidate list; engines[r] acts as rank r of
 * world n_engines: all solve their rows concurrently, the shards are gathered onto engines[0]'s device by peer copies
 * over xGMI, which assembles the matrix and runs the set-max.  Outputs as ipc_run.
 * Peer access: the shards move with hipMemcpyPeerAsync, which works whether or not the devices have peer access to each
 * other -- with access (hipDeviceEnablePeerAccess, the CALLER's decision: it is a process-wide setting the library does not
 * touch) the copy goes GPU to GPU over xGMI; without it, or where the platform denies it (IOMMU / container restrictions:
 * hipDeviceCanAccessPeer = 0), the runtime stages the copy through host memory: the same bytes arrive, at PCIe speed
 * (78 MB for C5: ~5 ms instead of < 1 ms).  Engines that share a device (tests) copy device to device. */
int ipc_run_sharded(ipc_engine_t** engines, int n_engines, uint64_t* bits_out, uint8_t* accepted_out);

/* Diagnostics of the last ipc_solve_rows(): number of solved cells, and their records. */
int ipc_cell_count(ipc_engine_t* h, int* n_cells);
int ipc_cell_info(ipc_engine_t* h, ipc_cell_info_t* out, int capacity);

/* HIP-event time (ms) spent in the cell-solver kernels of the last ipc_solve_rows(), the
 * number of solver launches and the pose-iterations... (bench.py roofline).  Blocks until the
 * stream has drained. */
int ipc_solver_time_ms(ipc_engine_t* h, double* ms, int* launches);

int ipc_synchronize(ipc_engine_t* h);

/* ---- faithful incremental mode and final map -------------------
 * The reference's own sequential algorithm on the GPU: state = current pose estimates + the
 * consensus set, one cluster solve per candidate.  Poses are SE2 [V][3] (x y theta) or SE3
 * [V][12] (R row-major, t), as in ipc_initial_poses(). */

/* Outcome of one cluster solve. */
typedef struct {
    int    lo, hi;           /* vertex-id span of the cluster                           */
    int    n_cluster_loops;  /* accepted edges absorbed into the cluster (0: fast path) */
    int    iterations, tries, flags;   /* as in ipc_cell_info_t                         */
    double max_chi2;         /* max edge chi2 after the optimisation                    */
    double chi2_total;       /* sum of chi2 after the optimisation                      */
    double chi2_initial;     /* sum of chi2 before it                                   */
} ipc_check_info_t;

/* Back to the state right after IPC::IPC (src/consensus.cpp:23-27): current poses = open-loop
 * propagation, empty consensus set.  Implicit in ipc_set_candidates(). */
int ipc_incremental_reset(ipc_engine_t* h);

/* Everything the first ipc_agreement_check would otherwise set up inside the caller's timed loop (src/simulation.cpp:36-38
 * times every call): the pose buffers, the streams and workspaces of the solves in flight, the stream-concurrency probe.
 * Belongs to construction (IPC::IPC, src/consensus.cpp:9-33, is not timed by the harness either); optional. */
int ipc_incremental_prepare(ipc_engine_t* h);

/* Replaces IPC<EDGE,VERTEX>::agreementCheck (src/consensus.cpp:43-75) for candidate k (FILE
 * index): computeIndependentSubgraph (:124-171), fast/slow threshold and iteration base
 * (:50-52), isAgreeingWithCurrentState on chain [lo,hi] + cluster loops + candidate from the
 * current poses (consensus_utils.cpp:7-22); on agreement the poses are kept, k joins the
 * consensus set and the tail is re-propagated (propagateCurrentGuess, consensus_utils.cpp:61-71);
 * otherwise the state is untouched.  *agrees = 1 / 0.  info may be NULL.
 * Called in the processing order (ipc_candidate_order) the call finds most results waiting: the library solves ahead of
 * the caller, several candidates at a time (IPC_SPEC_WINDOW, IPC_SPEC_AHEAD), from the current state and from the states
 * finished accepts leave behind, and uses a result only if the state it started from is the committed one when its turn
 * comes -- decisions, iteration counts and chi2 are those of the one-at-a-time loop, bit for bit.  Any other order is
 * served correctly as well: the candidate asked for moves to the head of the engine's prediction and the solves behind
 * it stay valid (they assumed rejects in front of them); an edit of the set throws the work done ahead away. */
int ipc_agreement_check(ipc_engine_t* h, int k, int* agrees, ipc_check_info_t* info);

/* What the faithful mode had to do besides the plain device-resident solve since the engine was created (diagnostics; the
 * decisions are the same either way):
 *   lost_launches          persistent launches whose grid barrier gave up (a workgroup never became resident: foreign work on
 *                          the GPU).  In the look-ahead pipeline the check is redone alone when its turn comes; a check that runs
 *                          alone is launched again, twice (relaunches), before the host-driven kernels take over
 *   host_solver_fallbacks  checks redone by the host-driven solver: lost three times, or the capacitance factorisation met a
 *                          non-positive pivot -- g2o's Levenberg retry (src/utils.cpp:104-105, "dl_var") on the literal normal
 *                          equations follows
 *   literal_band_solves    damped solves that factored the literal normal equations in the banded + bordered layout (round 6:
 *                          any cluster size; dense store below IPC_LITERAL_BAND_MIN_N = 3 072 unknowns or without a band,
 *                          and then only up to 24 000 unknowns -- beyond that the optimisation ends in Fail and the candidate
 *                          is rejected) */
typedef struct {
    long host_solver_fallbacks, lost_launches, relaunches, literal_band_solves;
} ipc_incremental_counters_t;
int ipc_incremental_counters(ipc_engine_t* h, ipc_incremental_counters_t* out);

/* IPC::getMaxConsensusSet (include/ipc/consensus.hpp:16): candidate FILE indices in set order. */
int ipc_consensus_size(ipc_engine_t* h, int* n);
int ipc_consensus_set(ipc_engine_t* h, int* out);

/* IPC::removeEdgeFromCnS (src/consensus.cpp:77-96): drops the first member joining the same
 * vertex pair as candidate k; *removed = 1 if one was found. */
int ipc_remove_from_consensus(ipc_engine_t* h, int k, int* removed);
/* IPC::addEdgeToCnS (src/consensus.cpp:98-119): appends k unless a member joins the same vertex
 * pair, then re-sorts the set by cmpEdgesTime (stable). */
int ipc_add_to_consensus(ipc_engine_t* h, int k);

/* Current pose estimates (the g2o vertex estimates the reference mutates). */
int ipc_current_poses(ipc_engine_t* h, double* poses_out);

/* Resume the loop of src/simulation.cpp:34-47 from a saved state.  Everything IPC::agreementCheck reads and writes is the
 * vertex estimates of the borrowed optimizer and _max_consensus_set (include/ipc/consensus.hpp:23-32), so a pair
 * (ipc_current_poses, ipc_consensus_set) taken at any point of a run, handed back here, continues that run -- on this engine
 * or on another engine of the same graph and candidate list (round 6: how the tests put the engine into late states of
 * BASELINE configs[3] / [4] whose checks the CPU oracle was given hours for).  poses as ipc_current_poses() returns them
 * (SE3: bit for bit the state; SE2: cos / sin of theta are recomputed, a rounding-level difference); cns = candidate FILE
 * indices in set order; resume_position = how many candidates of the processing order (ipc_candidate_order) count as
 * already handed out -- it only tells the look-ahead pipeline where the caller will continue, any value 0..N is correct. */
int ipc_incremental_set_state(ipc_engine_t* h, const double* poses, const int* cns, int n_cns, int resume_position);

/* Final map (src/simulation.cpp:50-65): open-loop guess, odometry information back to
 * (info * s) / s, every candidate with accepted[k] != 0, optimize(iterations) with vertex 0
 * fixed (the harness uses 1000).  poses_out and info may be NULL. */
int ipc_final_optimize(ipc_engine_t* h, const uint8_t* accepted, int iterations, double* poses_out,
                       ipc_check_info_t* info);

/* ---- diagnostics of the incremental mode's dense solver (tests only) ---------------------------------
 * Solves the SPD system the cluster solve factors per dog-leg iteration (the capacitance matrix of the accepted
 * loops).  system: (n+1) x n column major, lower triangle of S in rows 0..n-1, right-hand side in row n.
 * mode 0: one launch per block column (host-driven solver); mode 1: the orchestration inside the persistent
 * cluster kernel, on `workgroups` workgroups.  Both must return the same bits.  info: 0, or 1 + the first block
 * column with a non-positive pivot. */
int ipc_debug_dense_solve(int n, const double* system, int mode, int workgroups, double* x_out, int* info_out);

/* The same for the BANDED capacitance system of large clusters (round 5: the faithful mode on BASELINE configs[3] / [4];
 * the reference factors whatever computeIndependentSubgraph, src/consensus.cpp:124-171, grows the cluster to with g2o's
 * sparse solver, src/utils.cpp:104-105).  system: nb + m - 1 columns of W + m doubles -- column j holds rows j .. j+W-1
 * (band; W >= 64) and then the m dense rows (the wide loops' unknowns, right-hand side last); nb band columns. */
int ipc_debug_band_solve(int nb, int m, int W, const double* system, int workgroups, double* x_out, int* info_out);
/* computeIndependentSubgraph (src/consensus.cpp:124-171) on bare intervals (host code, no GPU): the accepted edges -- positions
 * into lo / hi -- a candidate [klo, khi] absorbs, and the hull.  sweep 0: the reference's fixed-point re-scan; sweep 1: one pass
 * over the intervals sorted by first vertex (what the engine uses for sets of 512 and more).  Same set, same hull. */
int ipc_debug_absorbed_edges(int klo, int khi, int n, const int* lo, const int* hi, int sweep, int* members_out, int* n_members_out,
                             int* lo_out, int* hi_out);
/* The band structure found for a set of loops (host code, no GPU): a / b = first / last vertex per loop, d = 3 (SE2) or
 * 6 (SE3) unknowns per loop, min_n = smallest system that is banded at all (0: always).  *use_out = 0: the dense solver
 * takes the cluster and nlb_out / bwb_out / order_out are NOT written; *use_out = 1: nlb_out = loops in the band,
 * bwb_out = half-bandwidth in blocks, order_out[nl] = the loops in band order, the border's wide loops last. */
int ipc_debug_band_plan(int d, int nl, const int* a, const int* b, int min_n, int* use_out, int* nlb_out, int* bwb_out,
                        int* order_out);

#ifdef __cplusplus
}
#endif
#endif /* IPC_AMD_H */
