/*
 * sfb.h -- C-ABI of the MI355X-native batched QP / MPC / EKF engine.
 *
 * This is the drop-in boundary for the hot path of pettni/smooth_feedback (reference @ v1).
 * The reference has no FFI layer: its boundary is a set of C++ templates instantiated in the
 * caller's translation unit.  Each entry point below names the reference interface it replaces
 * (file:line relative to the reference tree).  The C++ front in include/smooth_feedback_amd/
 * keeps the reference's class/function names and forwards to these functions.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ / torch types;
 *   - every function returns an sfb_status (0 = ok); per-problem results use the reference's
 *     QPSolutionStatus values (qp.hpp:82-92) in `code[]`; the library never aborts or throws;
 *   - `*_batch` functions take DEVICE pointers and are asynchronous on `stream` (a hipStream_t,
 *     NULL = default stream); results are valid after the stream is synchronised;
 *   - `*_batch_host` functions take HOST pointers, stage through device memory and are synchronous;
 *   - all floating point data is IEEE fp64; batch items are contiguous, batch-major;
 *   - there is NO CPU fallback: without a usable HIP device every compute call fails with
 *     SFB_ERR_NO_DEVICE;
 *   - DEVICES: every call works on the calling thread's CURRENT HIP device (hipSetDevice); the library never
 *     switches devices.  Several GPUs = one process (or one thread with its own current device) per GPU, each
 *     with its own plans' device copies, workspaces and streams -- items are independent, so a batch is sharded
 *     by slicing the arrays (bench.py --gpus N: one rank per GPU, one gather of the small outputs).  Objects
 *     that own device memory (sfb_mpc_swarm) remember their device and refuse calls from another one.
 */
#ifndef SFB_H
#define SFB_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum sfb_status {
  SFB_OK              = 0,
  SFB_ERR_INVALID_ARG = 1,
  SFB_ERR_UNSUPPORTED = 2, /* size / option outside what the HIP kernels implement */
  SFB_ERR_HIP         = 3, /* a HIP runtime call failed; see sfb_last_error() */
  SFB_ERR_NO_DEVICE   = 4
} sfb_status;

/* smooth::feedback::QPSolutionStatus, qp.hpp:82-92 (declaration order == value) */
typedef enum sfb_qp_status {
  SFB_QP_OPTIMAL           = 0,
  SFB_QP_POLISH_FAILED     = 1,
  SFB_QP_PRIMAL_INFEASIBLE = 2,
  SFB_QP_DUAL_INFEASIBLE   = 3,
  SFB_QP_MAX_ITERATIONS    = 4,
  SFB_QP_MAX_TIME          = 5,
  SFB_QP_UNKNOWN           = 6
} sfb_qp_status;

/*
 * smooth::feedback::QPSolverParams, qp_solver.hpp:29-68.  Numeric options are `float` exactly as
 * in the reference and are widened to double at the same places (:353-356, :587, :593, ...).
 * std::optional members are encoded with a negative value meaning "unset".
 */
typedef struct sfb_qp_params {
  float alpha;              /* :35  relaxation parameter                (1.6f)  */
  float rho;                /* :37  first dual step size                (0.1f)  */
  float sigma;              /* :39  second dual step length             (1e-6f) */
  int32_t scaling;          /* :42  scale problem                       (1)     */
  float eps_abs;            /* :45                                      (1e-3f) */
  float eps_rel;            /* :47                                      (1e-3f) */
  float eps_primal_inf;     /* :49                                      (1e-4f) */
  float eps_dual_inf;       /* :51                                      (1e-4f) */
  int64_t max_iter;         /* :54  optional<uint32_t>; <0 = unset      (-1)    */
  int64_t max_time_ns;      /* :57  optional<nanoseconds>; <0 = unset   (-1).  As in the reference (:504-507) the
                                    limit is tested at stopping checks that leave the status open, against
                                    the time since THIS problem's solve started -- here on the device clock
                                    (10 ns ticks), per batch item, time spent suspended in a time-sliced
                                    launch included.  Wall-clock limits are nondeterministic by nature:
                                    prefer max_iter for reproducible runs.                       */
  uint32_t stop_check_iter; /* :60                                      (25)    */
  int32_t polish;           /* :63                                      (1)     */
  uint32_t polish_iter;     /* :65                                      (5)     */
  float delta;              /* :67                                      (1e-6f) */
  int32_t verbose;          /* :32  host-pointer entry points print a summary of the call (phase times,
                                    status histogram, iteration statistics) and, when the call is ONE problem,
                                    the reference's per-iteration table before it (:409-420, :490-501; as data
                                    for any batch: sfb_sparse_qp_solve_batch_trace); ignored by the asynchronous
                                    device-pointer entry points                 (0)     */
  int32_t reuse_factor;     /* NOT in the reference (which re-scales and re-factorises on every solve(), reusing
                                    only the symbolic analysis, :424-426).  Shared-pattern sparse entry points only;
                                    the dense kernels ignore it.  Non-zero = the caller vouches that P and A of
                                    EVERY item are bit-identical to what the previous call with the same plan,
                                    workspace, batch, sigma and scaling solved for that item (a time-invariant
                                    MPC: only q, l, u move between ticks).  The kernel then keeps, per item,
                                    whatever that call left in the workspace and this call would recompute to the
                                    same bits: the compacted A, the scaling when c (which depends on |q|) comes
                                    out the same, and the LDL' factor when additionally the rho vector (which
                                    rows are equalities) is unchanged -- otherwise it recomputes.  Results are
                                    bit-identical to a call without the flag by construction.  The first call
                                    on a workspace may set it too (nothing to keep yet).  Ignored for items a
                                    pruned plan's guard sends to the fallback pool and in launches with an
                                    explicit order (their workspace slots are not the items').   (0)     */
} sfb_qp_params;

/* With max_iter unset the reference loops until a stopping test fires (possibly forever, e.g.
 * stop_check_iter==1, qp_solver.hpp:465).  The device path bounds every solve by this many
 * iterations and reports SFB_QP_MAX_ITERATIONS (iter == cap) when it is hit. */
#define SFB_QP_DEVICE_ITER_CAP 20000000

/* Largest n+m the register/LDS-resident dense kernels handle (lane i owns KKT row i).  Larger dense problems are
 * accepted by the same entry points:
 *   n+m <= SFB_QP_DENSE_BIG_MAX_K: one QP per wavefront with the KKT matrix and its PIVOTED dense LDL'
 *     (Eigen::LDLT<.,Upper>, qp_solver.hpp:259,:428,:462) in a per-QP HBM workspace -- same arithmetic as the
 *     small kernels, bit-identical to the dense CPU restatement (the reference's ASIF example, n = 3, m = 203, and
 *     test, n = 4, m = 301, are of this size);
 *   beyond: the shared-pattern sparse kernel with a full pattern -- same ADMM and stopping tests on the same
 *     matrix entries, but a fill-reducing elimination order without pivoting: agreement to rounding only. */
#define SFB_QP_DENSE_MAX_K 64
#define SFB_QP_DENSE_BIG_MAX_K 1024

const char *sfb_version(void);
/* Thread-local description of the last non-OK status returned on this thread. */
const char *sfb_last_error(void);

/* Debug knobs (tests, A/B measurements, diagnostics): launch shapes and engine choices, listed in
 * smooth_feedback_amd/csrc/knobs.h -- e.g. "SFB_SP_GRID" = "4" forces the time-sliced launch of the sparse kernel on a tiny
 * grid.  None changes a result.  value == NULL clears a knob; an unknown name is SFB_ERR_INVALID_ARG.  This call is the ONLY
 * way to set them: the library reads no environment variable. */
sfb_status sfb_debug_set(const char *name, const char *value);
/* Number of visible HIP devices (0 if none / runtime unusable). */
sfb_status sfb_device_count(int *count);

/*
 * Several devices from ONE process.  The reference has no parallelism (its benchmark "batch"
 * is a sequential for loop, benchmarks/bench_types.hpp:93); the items of a batch are independent, so a batch shards
 * trivially.  Two ways to use more than one GPU:
 *   - one process per GPU (torchrun / MPI): every entry point of this header works on the process's current device;
 *     the launcher shards the batch (bench.py, smooth_feedback_amd/sharding.py);
 *   - one process, the *_multi host-pointer entry points below: the batch is cut into one contiguous shard per entry
 *     of the device list, every shard runs on its own host thread with its device current (own plan upload, own
 *     workspace, own stream) and writes its slice of the caller's output arrays -- host memory is the gathering
 *     point, there is no device-to-device exchange on this path.  Results are identical to the single-device call.
 * sfb_set_devices: the device list of the *_multi entry points (default / count == 0: every visible device).  An
 * ordinal may appear more than once (its shards are then serialised on that device; used by the tests on 1-GPU boxes).
 */
sfb_status sfb_set_devices(const int *devices, int count);
sfb_status sfb_get_devices(int *devices, int capacity, int *count);

/* Defaults of QPSolverParams (qp_solver.hpp:29-68). */
void sfb_qp_params_default(sfb_qp_params *prm);

/*
 * Batched dense QP solve:   min 1/2 x'Px + q'x   s.t.  l <= Ax <= u      for `batch` problems.
 *
 * Replaces, per batch item, smooth::feedback::solve_qp(pbm, prm, warmstart)
 * (qp_solver.hpp:779-787) == QPSolver<QuadraticProgram<M,N,double>>(pbm, prm).solve(pbm, warmstart)
 * (:276, :343-568): scaling (:673-730), rho selection (:361-374), KKT + pivoted LDL' (:399-433),
 * ADMM loop with stopping tests every stop_check_iter iterations (:447-510, :574-644), polish
 * (:92-204, :515-539) and un-scaling (:544-548).
 *
 * Layout (qp.hpp:31-45, Eigen default column-major), item b at offset b*size:
 *   P  [batch][n*n]  col-major n x n (upper triangle feeds the KKT matrix, the full matrix is
 *                    used for residuals/objective exactly as the reference does)
 *   q  [batch][n]    A [batch][m*n] col-major m x n      l,u [batch][m]  (+-inf allowed)
 *   warm_x [batch][n], warm_y [batch][m]: previous primal/dual (both NULL = cold start)
 * Outputs (QPSolution, qp.hpp:95-108):
 *   x [batch][n] primal, y [batch][m] dual, obj [batch] (nullable), iter [batch] (nullable),
 *   code [batch] (sfb_qp_status values).
 * Non-finite data: like the reference, no input is validated -- NaN / inf in P, q, A propagate through the IEEE arithmetic
 * (l = +inf or u = -inf is the pre-check's PrimalInfeasible), and the kernels make of them exactly what the CPU restatement
 * does (tests/test_qp_dense_gpu.py, tests/test_qp_sparse_gpu.py::test_non_finite_*: codes, iteration counts and values up to
 * NaN payloads).  Shortcuts that rest on "0 times an entry is 0" (the first refinement round of polish) are covered by them.
 * Requires 1 <= n, 1 <= m (n+m > SFB_QP_DENSE_MAX_K: see there; n+m <= 19 198).
 * Working memory: sizes with n+m <= 32 (the four-per-wave kernel's records and queue) and n+m > SFB_QP_DENSE_MAX_K
 * (the factor in HBM) need device memory beyond the arguments.  This entry point takes it stream-ordered
 * (hipMallocAsync / hipFreeAsync) per call; sfb_qp_dense_solve_batch_ws below takes it from the caller instead.
 */
sfb_status sfb_qp_dense_solve_batch(const sfb_qp_params *prm, int64_t batch, int n, int m, const double *P,
                                    const double *q, const double *A, const double *l, const double *u,
                                    const double *warm_x, const double *warm_y, double *x, double *y,
                                    double *obj, uint32_t *iter, int32_t *code, void *stream);

/* Same (n+m <= 128) with the reference's verbose table (qp_solver.hpp:409-420, :490-501) as DATA: trace [batch][trace_rows][5]
 * (device) receives one row (ITER, OBJ, PRI_RES, DUA_RES, TIME in microseconds of the device clock since the solve began) per
 * stopping check of every problem, computed on the iterates of the solve itself; checks beyond trace_rows are dropped, the caller
 * presets ITER = -1 to tell used rows from unused ones.  One wave per problem whatever the batch size (a diagnostic path);
 * results are those of sfb_qp_dense_solve_batch, bit for bit.  SFB_ERR_UNSUPPORTED for n+m > 128. */
sfb_status sfb_qp_dense_solve_batch_trace(const sfb_qp_params *prm, int64_t batch, int n, int m, const double *P,
                                          const double *q, const double *A, const double *l, const double *u,
                                          const double *warm_x, const double *warm_y, double *x, double *y, double *obj,
                                          uint32_t *iter, int32_t *code, double *trace, int32_t trace_rows, void *stream);
/* Same, and the reference's closing summary (qp_solver.hpp:550-565) as DATA: phase_us (device, nullable) = batch x 16 doubles, of
 * which the first batch x 6 receive per problem the microseconds of the device's wall clock spent in
 *   [0] scaling and pre-check (before the reference's t0, :376)   [1] matrix filling (pivot order, zero, fill)
 *   [2] factorization   [3] iteration   [4] polish   [5] un-scale and report
 * (the remaining batch x 10 doubles are scratch for the kernel's stamps).  trace is nullable when trace_rows == 0. */
sfb_status sfb_qp_dense_solve_batch_phases(const sfb_qp_params *prm, int64_t batch, int n, int m, const double *P,
                                           const double *q, const double *A, const double *l, const double *u,
                                           const double *warm_x, const double *warm_y, double *x, double *y, double *obj,
                                           uint32_t *iter, int32_t *code, double *trace, int32_t trace_rows, double *phase_us,
                                           void *stream);

/*
 * Explicit workspaces.  The reference's QPSolver owns its work memory, allocated once by analyze()
 * (qp_solver.hpp:297-338) and re-used by every solve(); the equivalent here is an opaque device buffer the caller
 * creates once and hands to every call, so that no call allocates:
 *   sfb_qp_dense_workspace_bytes   device bytes a dense call of this shape needs (0 for 32 < n+m <= 64)
 *   sfb_workspace_create/destroy   device memory on the CURRENT device (bytes = 0 is allowed)
 *   sfb_workspace_info             its device pointer and size -- the pointer is also what the shared-pattern sparse
 *                                  entry points take as `workspace` (sfb_sparse_qp_plan_workspace_bytes)
 *   sfb_qp_dense_solve_batch_ws    sfb_qp_dense_solve_batch on that memory; asynchronous on `stream`.  One call at
 *                                  a time per workspace (calls on the same stream are ordered; use one workspace
 *                                  per stream otherwise).  SFB_ERR_INVALID_ARG if the workspace is too small or of
 *                                  another device.
 */
typedef struct sfb_workspace sfb_workspace;
sfb_status sfb_qp_dense_workspace_bytes(const sfb_qp_params *prm, int64_t batch, int n, int m, int64_t *bytes);
sfb_status sfb_workspace_create(int64_t bytes, sfb_workspace **out);
void sfb_workspace_destroy(sfb_workspace *workspace);
sfb_status sfb_workspace_info(const sfb_workspace *workspace, void **device_ptr, int64_t *bytes);
sfb_status sfb_qp_dense_solve_batch_ws(const sfb_qp_params *prm, int64_t batch, int n, int m, const double *P,
                                       const double *q, const double *A, const double *l, const double *u,
                                       const double *warm_x, const double *warm_y, double *x, double *y,
                                       double *obj, uint32_t *iter, int32_t *code, sfb_workspace *workspace,
                                       void *stream);

/* Same with host pointers (H2D copy, solve, D2H copy, synchronous) on the current device.  One device staging buffer
 * per device is kept between calls (the analogue of the working memory a reference QPSolver object keeps,
 * qp_solver.hpp:297-338); concurrent callers are not serialised.  sfb_host_staging_trim() frees what is kept. */
sfb_status sfb_qp_dense_solve_batch_host(const sfb_qp_params *prm, int64_t batch, int n, int m,
                                         const double *P, const double *q, const double *A, const double *l,
                                         const double *u, const double *warm_x, const double *warm_y,
                                         double *x, double *y, double *obj, uint32_t *iter, int32_t *code);
void sfb_host_staging_trim(void);
/* ... with the verbose table as data (trace [batch][trace_rows][5], HOST memory; see sfb_qp_dense_solve_batch_trace; n+m <= 128).
 * With prm->verbose on ONE problem of that size class the host entry point above prints this table itself. */
sfb_status sfb_qp_dense_solve_batch_host_trace(const sfb_qp_params *prm, int64_t batch, int n, int m, const double *P,
                                               const double *q, const double *A, const double *l, const double *u,
                                               const double *warm_x, const double *warm_y, double *x, double *y, double *obj,
                                               uint32_t *iter, int32_t *code, double *trace, int32_t trace_rows);
/* ... and the per-phase times (phase_us [batch][6], HOST memory, nullable; see sfb_qp_dense_solve_batch_phases).
 * sfb_qp_dense_solve_batch_host with prm->verbose set and batch == 1 prints table and summary from the same data. */
sfb_status sfb_qp_dense_solve_batch_host_phases(const sfb_qp_params *prm, int64_t batch, int n, int m, const double *P,
                                                const double *q, const double *A, const double *l, const double *u,
                                                const double *warm_x, const double *warm_y, double *x, double *y, double *obj,
                                                uint32_t *iter, int32_t *code, double *trace, int32_t trace_rows,
                                                double *phase_us);
/* ... and sharded over the device list (sfb_set_devices); same arguments, same results. */
sfb_status sfb_qp_dense_solve_batch_host_multi(const sfb_qp_params *prm, int64_t batch, int n, int m,
                                               const double *P, const double *q, const double *A, const double *l,
                                               const double *u, const double *warm_x, const double *warm_y,
                                               double *x, double *y, double *obj, uint32_t *iter, int32_t *code);

/* ------------------------------------------------------------------------------------------
 * Sparse QPs sharing ONE sparsity pattern (a swarm of MPC problems from the same transcription).
 *
 * Layout follows QuadraticProgramSparse (qp.hpp:60-79): P in CSC (Eigen::SparseMatrix default)
 * exactly as the caller stores it -- only entries with col >= row enter the KKT matrix
 * (qp_solver.hpp:384) while residuals/objective multiply by P as stored (:589,:547) --, A in CSR
 * (Eigen::RowMajor, qp.hpp:74).  Index arrays are shared by the batch; values are per item:
 *   Px [batch][nnzP], q [batch][n], Ax [batch][nnzA], l,u [batch][m].
 * Indices must be strictly ascending inside each column of P / row of A (Eigen compressed form).
 * ---------------------------------------------------------------------------------------- */
typedef struct sfb_sparse_qp_plan sfb_sparse_qp_plan; /* opaque */

/*
 * Symbolic analysis, once per pattern.  Replaces QPSolver<QuadraticProgramSparse>::analyze
 * (qp_solver.hpp:297-338) and SimplicialLDLT::analyzePattern (:424): KKT pattern, fill-reducing
 * elimination order, pattern of L.  Host only (no device needed).
 *   ordering: 0 natural, 1 minimum degree (Eigen's AMD is not reproducible without Eigen; any
 *   fill-reducing order yields the same algorithm up to rounding);  user_perm (n+m entries,
 *   new -> old, nullable) overrides `ordering`.
 */
sfb_status sfb_sparse_qp_plan_create(int n, int m, const int32_t *P_colptr, const int32_t *P_rowind,
                                     const int32_t *A_rowptr, const int32_t *A_colind, int ordering,
                                     const int32_t *user_perm, sfb_sparse_qp_plan **plan);
/* Same with a constrained minimum-degree order: stage[n+m] (nullable) -- unknowns (variables 0..n-1,
 * constraint duals n..n+m-1) of a lower stage are eliminated before any of a higher stage.  An MPC
 * transcription marks the states shared by neighbouring mesh intervals as stage 1: shorter
 * elimination tree and less fill than unconstrained minimum degree. */
sfb_status sfb_sparse_qp_plan_create_staged(int n, int m, const int32_t *P_colptr, const int32_t *P_rowind,
                                            const int32_t *A_rowptr, const int32_t *A_colind, int ordering,
                                            const int32_t *user_perm, const int32_t *stage,
                                            sfb_sparse_qp_plan **plan);
/*
 * Same for a transcription that stores entries of A which are zero for EVERY item (the reference's ocp_to_qp
 * writes dense Jacobian blocks: block_add, utils/sparse.hpp:33-50, ocp_to_qp.hpp:258-264 -- two thirds of the
 * stored entries of the SE2 x R^3 MPC problem are explicit zeros).  A_keep[nnzA] (nullable): 0 = the caller
 * declares this stored entry zero in every item solved with the plan.  The KKT pattern, the elimination order
 * and the pattern of L are built from the kept entries only (the headline MPC pattern: nnz(L) 41 030 -> 13 710);
 * the value arrays keep the caller's layout.  Exactness: a zero entry contributes exact zeros to every sum it
 * takes part in, so the results equal those of the whole pattern under the same elimination order up to the
 * sign of zeros.  The declaration is CHECKED on the device for every item: an item with a non-zero (or NaN)
 * masked entry is solved on the whole pattern instead (same elimination order), by fallback launches inside
 * the same call -- a wrong mask costs time, never correctness.
 */
sfb_status sfb_sparse_qp_plan_create_pruned(int n, int m, const int32_t *P_colptr, const int32_t *P_rowind,
                                            const int32_t *A_rowptr, const int32_t *A_colind, int ordering,
                                            const int32_t *user_perm, const int32_t *stage, const uint8_t *A_keep,
                                            sfb_sparse_qp_plan **plan);
void sfb_sparse_qp_plan_destroy(sfb_sparse_qp_plan *plan);
/* nnz of the KKT upper triangle, nnz of L (strictly lower), bytes of device workspace PER ITEM (batch times this
 * always suffices; sfb_sparse_qp_plan_workspace_bytes is the exact requirement of a call). */
sfb_status sfb_sparse_qp_plan_info(const sfb_sparse_qp_plan *plan, int64_t *nnzK, int64_t *nnzL,
                                   int64_t *workspace_bytes_per_item);
/* Exact device workspace of one call with `batch` items. */
sfb_status sfb_sparse_qp_plan_workspace_bytes(const sfb_sparse_qp_plan *plan, int64_t batch, int64_t *bytes);
/* Pruned plans: kept entries of A and nnz(L) of the fallback (whole-pattern) analysis; otherwise nnzA and nnz(L). */
sfb_status sfb_sparse_qp_plan_pruned_info(const sfb_sparse_qp_plan *plan, int64_t *nnzA_kept, int64_t *nnzL_fallback);
/* The elimination order in use (n+m entries, new -> old). */
sfb_status sfb_sparse_qp_plan_get_perm(const sfb_sparse_qp_plan *plan, int32_t *perm);
/* The order in which the numeric factorisation sums the contributions to an entry of L: rank[n+m] of every
 * column of the permuted matrix (a postorder of the plan's elimination tree; sources are summed in ascending
 * rank).  A floating-point detail, exposed so that a CPU restatement can reproduce the factor bit for bit.
 * fallback != 0: the order of a pruned plan's whole-pattern fallback analysis. */
sfb_status sfb_sparse_qp_plan_get_factor_order(const sfb_sparse_qp_plan *plan, int fallback, int32_t *rank);

/*
 * Batched sparse solve (device pointers, asynchronous on `stream`).  Replaces, per item,
 * QPSolver<QuadraticProgramSparse<double>>::solve(pbm, warmstart) (qp_solver.hpp:343-568 with the
 * sparse branches :379-397, :423-426, :452-460) as called by MPC::operator() (mpc.hpp:491).
 * `workspace`: device buffer of sfb_sparse_qp_plan_workspace_bytes(plan, batch) bytes (batch *
 * workspace_bytes_per_item always suffices), 16-byte aligned, caller-owned, reusable.
 * Outputs as sfb_qp_dense_solve_batch.
 */
sfb_status sfb_sparse_qp_solve_batch(sfb_sparse_qp_plan *plan, const sfb_qp_params *prm, int64_t batch,
                                     const double *Px, const double *q, const double *Ax, const double *l,
                                     const double *u, const double *warm_x, const double *warm_y, double *x,
                                     double *y, double *obj, uint32_t *iter, int32_t *code, void *workspace,
                                     void *stream);
/* Same with a launch order: order[batch] (device, nullable) is a permutation of 0..batch-1, launch position -> item.
 * The results do not depend on it.  A caller that can predict the iteration counts (an MPC swarm: those of the
 * previous tick) puts the longest-running items first, so that they run alongside the bulk of the batch instead of
 * finishing alone at the end of the launch. */
sfb_status sfb_sparse_qp_solve_batch_ordered(sfb_sparse_qp_plan *plan, const sfb_qp_params *prm, int64_t batch,
                                             const double *Px, const double *q, const double *Ax, const double *l,
                                             const double *u, const double *warm_x, const double *warm_y, double *x,
                                             double *y, double *obj, uint32_t *iter, int32_t *code, void *workspace,
                                             const int32_t *order, void *stream);
/* Same with the reference's verbose table (qp_solver.hpp:409-420, :490-501) as DATA: trace [batch][trace_rows][5] (device)
 * receives, per item and stopping check, one row (ITER, OBJ = (0.5 P x + q).x, PRI_RES = |A x - z|_inf,
 * DUA_RES = |P x + q + A'y|_inf on the unscaled iterate -- the reference's expressions -- and TIME in microseconds of
 * the device clock since the item's solve began); checks beyond trace_rows are dropped, the caller presets ITER = -1 to
 * recognise unused rows.  One wave per item without time slicing, same arithmetic and results as the plain call
 * (a launch of a separate instance of the kernel: diagnostics, not the fast path). */
sfb_status sfb_sparse_qp_solve_batch_trace(sfb_sparse_qp_plan *plan, const sfb_qp_params *prm, int64_t batch,
                                           const double *Px, const double *q, const double *Ax, const double *l,
                                           const double *u, const double *warm_x, const double *warm_y, double *x,
                                           double *y, double *obj, uint32_t *iter, int32_t *code, void *workspace,
                                           double *trace, int32_t trace_rows, void *stream);
/* Same, and the reference's closing summary (qp_solver.hpp:550-565: Matrix filling / Factorization / Iteration / Polish) as
 * DATA: phase_us [batch][6] (device, required) receives per item the microseconds of the device's wall clock spent in
 *   [0] scaling and pre-check (before the reference's t0, :376)   [1] matrix filling   [2] factorization   [3] iteration
 *   [4] polish   [5] un-scale and report
 * -- their sum is the time the item held its wavefront, [1] + .. + [4] is the reference's "Total time".  trace is nullable here.
 * Like the table: the TRACE instance of the kernel, one wave per item, same arithmetic and results as the plain call. */
sfb_status sfb_sparse_qp_solve_batch_phases(sfb_sparse_qp_plan *plan, const sfb_qp_params *prm, int64_t batch,
                                            const double *Px, const double *q, const double *Ax, const double *l,
                                            const double *u, const double *warm_x, const double *warm_y, double *x,
                                            double *y, double *obj, uint32_t *iter, int32_t *code, void *workspace,
                                            double *trace, int32_t trace_rows, double *phase_us, void *stream);
/* Same with host pointers (synchronous).  The device buffers are owned by the plan and kept between calls
 * (grow-only, freed by sfb_sparse_qp_plan_destroy) -- the analogue of the working memory a QPSolver object
 * keeps between solves (qp_solver.hpp:242-338); host-pointer calls on ONE plan and ONE device are serialised. */
sfb_status sfb_sparse_qp_solve_batch_host(sfb_sparse_qp_plan *plan, const sfb_qp_params *prm, int64_t batch,
                                          const double *Px, const double *q, const double *Ax, const double *l,
                                          const double *u, const double *warm_x, const double *warm_y,
                                          double *x, double *y, double *obj, uint32_t *iter, int32_t *code);
/* ... with the verbose table as data (trace [batch][trace_rows][5], HOST memory, nullable; see
 * sfb_sparse_qp_solve_batch_trace).  sfb_sparse_qp_solve_batch_host with prm->verbose set and batch == 1 -- the
 * reference's use of the flag, one QPSolver object -- collects the table this way and prints it in the reference's
 * format (header, one line per stopping check, summary). */
sfb_status sfb_sparse_qp_solve_batch_host_trace(sfb_sparse_qp_plan *plan, const sfb_qp_params *prm, int64_t batch,
                                                const double *Px, const double *q, const double *Ax, const double *l,
                                                const double *u, const double *warm_x, const double *warm_y, double *x,
                                                double *y, double *obj, uint32_t *iter, int32_t *code, double *trace,
                                                int32_t trace_rows);
/* ... and the per-phase times (phase_us [batch][6], HOST memory, nullable; see sfb_sparse_qp_solve_batch_phases).  With
 * prm->verbose set and batch == 1 the summary of qp_solver.hpp:550-565 is printed from them. */
sfb_status sfb_sparse_qp_solve_batch_host_phases(sfb_sparse_qp_plan *plan, const sfb_qp_params *prm, int64_t batch,
                                                 const double *Px, const double *q, const double *Ax, const double *l,
                                                 const double *u, const double *warm_x, const double *warm_y,
                                                 double *x, double *y, double *obj, uint32_t *iter, int32_t *code,
                                                 double *trace, int32_t trace_rows, double *phase_us);
/* ... and sharded over the device list (sfb_set_devices): one contiguous shard, host thread, plan upload and workspace
 * per device; same arguments, same results (calls on different devices are not serialised). */
sfb_status sfb_sparse_qp_solve_batch_host_multi(sfb_sparse_qp_plan *plan, const sfb_qp_params *prm, int64_t batch,
                                                const double *Px, const double *q, const double *Ax, const double *l,
                                                const double *u, const double *warm_x, const double *warm_y,
                                                double *x, double *y, double *obj, uint32_t *iter, int32_t *code);

/* ------------------------------------------------------------------------------------------
 * Device-side MPC assembly and the device-resident swarm.
 *
 * sfb_mpc_assemble_batch replaces, per agent, the numeric fill of the QP that MPC::operator() does before
 * the solve (mpc.hpp:473-486): ocp_to_qp_update_dyn (ocp_to_qp.hpp:240-275), ocp_to_qp_update_cr (:279-323)
 * and ocp_to_qp_update_ce (:326-373) -- given the per-node linearisation that needs the user's callbacks
 * (dynamics f, its right-Jacobians, the desired trajectory), which stays on the host.  The kernel performs
 * the arithmetic of those functions in their order, including ad(f + dxdes) of the state group (:262-264),
 * so A, l, u equal the host transcription bit for bit.
 *
 * Constraint rows [dyn (N*nx) | cr (N*ncr) | ce (nx)], variables [x_0..x_N (nx each) | u_0..u_{N-1} (nu each)],
 * N = kmesh * nivals collocation nodes; A in CSR with the pattern of ocp_to_qp_allocate (:56-69):
 *   dyn row (node M+i of interval starting at node M, component d):
 *        for j = 0..kmesh: block j == i -> nx entries (columns of x_{M+i}), else 1 entry (component d of x_{M+j});
 *        then nu entries (u_{M+i})                                   => kmesh + nx + nu entries
 *   cr row: nx entries (x_node) then nu entries (u_node);   ce row: nx entries (x_0).
 *
 * Per-agent record of the linearisation ("stage blocks"), contiguous doubles, matrices ROW-major (d, c):
 *   [ f (N*nx) | dxdes (N*nx) | dfdx (N*nx*nx) | dfdu (N*nx*nu) | c (N*ncr) | dcdx (N*ncr*nx) | dcdu (N*ncr*nu)
 *     | e (nx) | J (nx*nx) ]
 *   f, dfdx, dfdu: dynamics and right-Jacobians at (xdes(t+t_i), udes(t+t_i)); dxdes: body velocity of the
 *   desired trajectory; c, dcdx, dcdu: running constraint and Jacobians; e = xdes(t) (-) x and
 *   J = d^r exp^{-1}(e) (MPCCE, mpc.hpp:288-301).
 * With layout->jac_keep the Jacobian blocks are packed: [ f | dxdes | dfdx kept (N * n_fx) | dfdu kept | c | dcdx kept |
 * dcdu kept | e | J kept ].
 * If shared_jac is given (time-invariant linearisation: group-linear model on a fixed trajectory), the
 * Jacobians [dfdx | dfdu | dcdx | dcdu] are read from that ONE record and the per-agent record shrinks to
 *   [ f | dxdes | c | e | J ].
 * ---------------------------------------------------------------------------------------- */
typedef enum { SFB_LIE_RN = 0, SFB_LIE_SE2 = 1, SFB_LIE_SO3 = 2 } sfb_lie_kind;

typedef struct sfb_mpc_layout {
  int32_t nx, nu, ncr;      /* dof of state and input, running-constraint rows per node */
  int32_t kmesh, nivals;    /* collocation nodes per interval, intervals */
  double tf;                /* horizon (MPCParams::tf, mpc.hpp:322) */
  const double *alpha;      /* [nivals] interval scale alpha of interval_diffmat_unscaled (ocp_to_qp.hpp:243)  */
  const double *D;          /* [(kmesh+1)*kmesh] differentiation matrix, D[j*kmesh + i] = Dus(j, i) (:243,:268) */
  int32_t nparts;           /* components of the state bundle, in order; 0 = commutative state (no ad term) */
  const int32_t *part_kind; /* [nparts] sfb_lie_kind */
  const int32_t *part_dof;  /* [nparts] sums to nx (SE2 and SO3: 3) */
  const double *crl, *cru;  /* [ncr] bounds of the running constraint (OCP::crl / cru as set by the MPC constructor) */
  const uint8_t *jac_keep;  /* nullable.  Packed per-agent Jacobians: one flag per entry of the blocks
                               [dfdx nx*nx | dfdu nx*nu | dcdx ncr*nx | dcdu ncr*nu | J nx*nx], row-major (d, c);
                               the record then holds, per node (J: once), only the entries whose flag is set, in
                               that order -- the others are 0.0 by the CALLER's guarantee (the dense Jacobian blocks
                               of a bundle state are mostly structural zeros: two thirds of the headline model's
                               record).  Ignored for records that share their Jacobians.  nu <= 32. */
} sfb_mpc_layout;

/* doubles in one per-agent record (shared_jac == 0: with the Jacobians) and in the shared Jacobian record */
int64_t sfb_mpc_record_doubles(const sfb_mpc_layout *layout, int shared_jac);
int64_t sfb_mpc_shared_jac_doubles(const sfb_mpc_layout *layout);
/* nnz of A for the layout */
int64_t sfb_mpc_nnzA(const sfb_mpc_layout *layout);

/* Device pointers, asynchronous on `stream`.  records [batch][record_doubles], shared_jac nullable,
 * Ax [batch][nnzA], l, u [batch][m = N*nx + N*ncr + nx].  layout and its arrays are host memory. */
sfb_status sfb_mpc_assemble_batch(const sfb_mpc_layout *layout, int64_t batch, const double *records,
                                  const double *shared_jac, double *Ax, double *l, double *u, void *stream);

/*
 * A swarm of `agents` MPC controllers with one transcription, resident on the device: P, q (constant over
 * ticks: set by the constructor, mpc.hpp:423), the warm start each agent keeps between calls (mpc.hpp:509-516) and all solver
 * memory stay in HBM.  One tick = upload the records, assemble, solve, store the warm starts, download the
 * small outputs.  Replaces the loop `for each agent: u = mpc(t, x)` (MPC::operator(), mpc.hpp:458-519).
 * Warm-started ticks launch the agents in descending order of their previous tick's iteration count
 * (sfb_sparse_qp_solve_batch_ordered).
 *   plan: pattern of the transcription's QP (its A pattern is checked against the layout); must outlive the swarm.
 *   Px [nnzP], q [n]: host, shared by all agents.
 */
typedef struct sfb_mpc_swarm sfb_mpc_swarm; /* opaque */
sfb_status sfb_mpc_swarm_create(sfb_sparse_qp_plan *plan, const sfb_mpc_layout *layout, const double *Px,
                                const double *q, int64_t agents, sfb_mpc_swarm **swarm);
void sfb_mpc_swarm_destroy(sfb_mpc_swarm *swarm);
/* forget the warm starts (MPC::reset_warmstart, mpc.hpp:603) */
sfb_status sfb_mpc_swarm_reset_warmstart(sfb_mpc_swarm *swarm);
/*
 * One tick, host pointers, synchronous.  records / shared_jac as above (host).  warmstart != 0: every agent
 * starts from the last solution it stored; a solution is stored when its code is Optimal, MaxTime or
 * MaxIterations (mpc.hpp:510-516), otherwise the agent keeps the older one.
 * Outputs (host): du0 [agents][nu] = primal entries of u_0 (the caller applies udes(t) (+) du0, mpc.hpp:518),
 * iter [agents] (nullable), code [agents], primal [agents][n] and dual [agents][m] (both nullable: full
 * solution for x_traj / u_traj, mpc.hpp:494-507).
 */
sfb_status sfb_mpc_swarm_step_host(sfb_mpc_swarm *swarm, const sfb_qp_params *prm, const double *records,
                                   const double *shared_jac, int warmstart, double *du0, uint32_t *iter,
                                   int32_t *code, double *primal, double *dual);

/* Pipelined upload (optional).  sfb_mpc_swarm_host_records returns a PINNED host buffer [agents][record_doubles
 * (shared_jac = 0)] owned by the swarm; the host threads that linearise write their agents' records straight into it
 * and announce finished ranges with sfb_mpc_swarm_upload(first, count), which starts an asynchronous copy on the
 * swarm's copy stream and returns at once -- the DMA of one chunk overlaps with the linearisation of the next.
 * sfb_mpc_swarm_step_host called with records == that buffer (and shared_jac == NULL) waits for those copies, uploads
 * whatever was not announced, and proceeds as above.  A range must not be rewritten between its upload and the step. */
sfb_status sfb_mpc_swarm_host_records(sfb_mpc_swarm *swarm, double **records);
sfb_status sfb_mpc_swarm_upload(sfb_mpc_swarm *swarm, int64_t first, int64_t count);

/* Records produced ON the device.  sfb_mpc_swarm_device_records returns the swarm's device record buffer
 * ([agents][record_doubles], the current packing); a caller whose model is device-callable fills it with its own kernel
 * (include/smooth_feedback_amd/mpc_device.hpp does, one thread per agent and node) on the null stream, and
 * sfb_mpc_swarm_step_resident runs the tick from there: assemble, solve, warm starts, small outputs -- nothing but the
 * agents' times and states goes up. */
sfb_status sfb_mpc_swarm_device_records(sfb_mpc_swarm *swarm, double **records, int64_t *record_doubles);
sfb_status sfb_mpc_swarm_step_resident(sfb_mpc_swarm *swarm, const sfb_qp_params *prm, int warmstart, double *du0,
                                       uint32_t *iter, int32_t *code, double *primal, double *dual);

/* Switch the packing of the per-agent records (layout->jac_keep semantics; NULL = unpacked) of an existing swarm --
 * e.g. back to full records when a linearisation turns out to have a non-zero where the flags said zero.  The swarm's
 * buffers are sized for unpacked records, warm starts and solver memory are untouched.  record_doubles (nullable)
 * receives the new record length.  Pending uploads are waited for and forgotten. */
sfb_status sfb_mpc_swarm_set_jac_keep(sfb_mpc_swarm *swarm, const uint8_t *jac_keep, int64_t *record_doubles);
/* Device buffers of the last tick (Ax [agents][nnzA], l, u [agents][m]) for inspection; valid until the next call. */
sfb_status sfb_mpc_swarm_debug_buffers(sfb_mpc_swarm *swarm, const double **Ax, const double **l, const double **u);

/* ------------------------------------------------------------------------------------------
 * Batched Lie-group EKF: covariance propagation and Kalman update for `batch` independent filters.
 *
 * Replaces the matrix part of smooth::feedback::EKF<G>::predict / ::update (ekf.hpp:79-103,
 * :116-139).  The caller (host) keeps what needs the user's callbacks on the group:
 *   predict: A = -ad(f(t,g)) + d^r f/dx at the estimate (ekf.hpp:86-87) and the state step
 *            g <- g (+) dt f (:97);     update: H = d^r h/dx (:119), r = y (-) h(g), g <- g (+) delta (:137).
 * One predict call is ONE explicit-Euler substep of the covariance ODE (the default stepper,
 * ekf.hpp:30,:96); substepping and re-linearisation are the caller's loop exactly as in :93-102.
 * Layout: item-major contiguous, matrices column-major (Eigen default): P, A, Q [batch][dof*dof],
 * H [batch][ny*dof], R [batch][ny*ny], r [batch][ny], delta [batch][dof].  Only the upper triangles
 * of Q and R are read (ekf.hpp:77,:114).  q_shared / r_shared / dt_shared != 0: Q / R / dt point to
 * ONE matrix / scalar used by every item.  info[batch] (nullable): 0 ok, 1 = LDLT of S failed.
 * Supported sizes: dof, ny <= 16.  dof in {2,3,4,6,7} with ny in {1,2,3}, and (4,4), (6,6), run one filter per lane,
 * register-resident (the streaming kernels of the benchmark configuration); dof in {8,9,10} with ny <= 3 use a
 * per-lane update behind a separate predict launch; every other size (the reference's own tests also use (3,10))
 * runs one filter per wavefront with its matrices in LDS.  Same results in every case.
 * Device pointers, asynchronous on `stream`; P is updated in place.
 * ---------------------------------------------------------------------------------------- */
sfb_status sfb_ekf_predict_batch(int64_t batch, int dof, const double *A, const double *Q, int q_shared,
                                 const double *dt, int dt_shared, double *P, void *stream);
sfb_status sfb_ekf_update_batch(int64_t batch, int dof, int ny, const double *H, const double *R, int r_shared,
                                const double *r, double *P, double *delta, int32_t *info, void *stream);
/* predict with an explicit choice of the stepper (ekf.hpp:27-31: the Stp template argument): ONE step of the
 * covariance ODE dP/dt = symU(A P + P A' + Q) by explicit Euler (SFB_EKF_EULER == sfb_ekf_predict_batch) or by
 * boost::numeric::odeint::runge_kutta4 (SFB_EKF_RK4, the stepper tests/test_ekf.cpp:113-115 instantiates) with ONE
 * A for all four stages -- exact for dynamics f(t, x) that do not depend on t explicitly; see
 * sfb_ekf_predict_rk4_batch otherwise.  The state step g <- stepper(g) stays on the host. */
typedef enum { SFB_EKF_EULER = 0, SFB_EKF_RK4 = 1 } sfb_ekf_stepper;
sfb_status sfb_ekf_predict_stepper_batch(int stepper, int64_t batch, int dof, const double *A, const double *Q,
                                         int q_shared, const double *dt, int dt_shared, double *P, void *stream);
/* runge_kutta4 step of the covariance ODE with the linearisation at the stage times: the reference evaluates
 * A = -ad(f(t_s, g)) + d^r f/dx|_(t_s, g) inside cov_ode (ekf.hpp:84-89) at t_s = t (A), t + dt/2 (A_mid, stages 2
 * and 3) and t + dt (A_end); g is frozen during the covariance step (:96).  A_mid == A_end == NULL: A everywhere. */
sfb_status sfb_ekf_predict_rk4_batch(int64_t batch, int dof, const double *A, const double *A_mid, const double *A_end,
                                     const double *Q, int q_shared, const double *dt, int dt_shared, double *P,
                                     void *stream);
sfb_status sfb_ekf_predict_rk4_batch_host(int64_t batch, int dof, const double *A, const double *A_mid,
                                          const double *A_end, const double *Q, int q_shared, const double *dt,
                                          int dt_shared, double *P);
/* predict immediately followed by update in one launch (one pass over P). */
sfb_status sfb_ekf_predict_update_batch(int64_t batch, int dof, int ny, const double *A, const double *Q,
                                        int q_shared, const double *dt, int dt_shared, const double *H,
                                        const double *R, int r_shared, const double *r, double *P,
                                        double *delta, int32_t *info, void *stream);
/* Host-pointer variants (stage through device memory, synchronous).  Pass NULL A to skip predict,
 * NULL H to skip update. */
sfb_status sfb_ekf_predict_stepper_batch_host(int stepper, int64_t batch, int dof, const double *A, const double *Q,
                                              int q_shared, const double *dt, int dt_shared, double *P);
sfb_status sfb_ekf_step_batch_host(int64_t batch, int dof, int ny, const double *A, const double *Q, int q_shared,
                                   const double *dt, int dt_shared, const double *H, const double *R,
                                   int r_shared, const double *r, double *P, double *delta, int32_t *info);

/*
 * Synthetic workload of the reference benchmark: random_qp(m, n, density, rng)
 * (benchmarks/bench_types.hpp:19-41) drawn `batch` times from ONE std::default_random_engine
 * seeded with `seed` (benchmarks/bench.cpp:146,170).  Host buffers, layout as above. CPU only.
 */
sfb_status sfb_random_qp_batch(uint32_t seed, int64_t batch, int m, int n, double density, double *P,
                               double *q, double *A, double *l, double *u);

#ifdef __cplusplus
}
#endif
#endif /* SFB_H */
