/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * CPU restatement, in plain C, of the dense QP path of pettni/smooth_feedback @ v1:
 *   include/smooth/feedback/qp_solver.hpp  (QPSolver::scale/solve/check_stopping, detail::polish_qp)
 *   include/smooth/feedback/qp.hpp         (QuadraticProgram, QPSolution, QPSolutionStatus)
 * and of the one third-party routine on that path that is absent from /root/reference:
 *   Eigen 3.4.0  Eigen/src/Cholesky/LDLT.h  (LDLT<.,Upper>::compute / solveInPlace; pinned only
 *   by the reference CI, .github/workflows/build_and_test.yml:29-36).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * PARITY STATUS: pinned against every known-answer test of tests/test_qp.cpp (status codes
 * exactly, primal/objective to the tests' tolerances; see tests/test_oracle_qp_golden.py).
 * The reference itself cannot be built here (no Eigen / Boost / smooth), so bit-level parity
 * with a real Eigen build (summation order inside Eigen's products, pivot tie-breaking) is
 * UNPINNED; the summation orders below are this oracle's own fixed choice and the HIP kernel
 * follows the same choice so that the two agree bit-for-bit.
 */
#ifndef SFB_QP_ORACLE_H
#define SFB_QP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same field order / types as sfb_qp_params in include/sfb.h (checked by tests). */
typedef struct oracle_qp_params {
  float alpha;            /* qp_solver.hpp:35 */
  float rho;              /* :37 */
  float sigma;            /* :39 */
  int32_t scaling;        /* :42 */
  float eps_abs;          /* :45 */
  float eps_rel;          /* :47 */
  float eps_primal_inf;   /* :49 */
  float eps_dual_inf;     /* :51 */
  int64_t max_iter;       /* :54  (<0 : unset / unlimited) */
  int64_t max_time_ns;    /* :57  (<0 : unset). Wall-clock, nondeterministic; oracle honours it. */
  uint32_t stop_check_iter; /* :60 */
  int32_t polish;         /* :63 */
  uint32_t polish_iter;   /* :65 */
  float delta;            /* :67 */
  int32_t verbose;        /* :32 (ignored by the oracle) */
  int32_t reuse_factor;   /* product extension (include/sfb.h): results never depend on it; ignored by the oracle,
                             which like the reference scales and factorises on every solve */
} oracle_qp_params;

void oracle_qp_params_default(oracle_qp_params *p);

/* QPSolutionStatus values, qp.hpp:82-92 */
enum {
  ORACLE_QP_OPTIMAL = 0,
  ORACLE_QP_POLISH_FAILED = 1,
  ORACLE_QP_PRIMAL_INFEASIBLE = 2,
  ORACLE_QP_DUAL_INFEASIBLE = 3,
  ORACLE_QP_MAX_ITERATIONS = 4,
  ORACLE_QP_MAX_TIME = 5,
  ORACLE_QP_UNKNOWN = 6
};

/*
 * One dense solve == solve_qp(pbm, prm, warmstart)  (qp_solver.hpp:779-787).
 * P: n*n col-major, q: n, A: m*n col-major, l,u: m. warm_x / warm_y may be NULL (cold start).
 * Outputs: x[n], y[m], *obj, *iter, *code.  Returns 0, or -1 on bad arguments / alloc failure.
 */
int oracle_qp_dense_solve(const oracle_qp_params *prm, int n, int m, const double *P, const double *q,
                          const double *A, const double *l, const double *u, const double *warm_x,
                          const double *warm_y, double *x, double *y, double *obj, uint32_t *iter,
                          int32_t *code);

/*
 * Sequential loop over a batch exactly like benchmarks/bench_types.hpp:93, optionally split
 * statically over nthreads POSIX threads (contiguous chunks). Batch-major contiguous arrays.
 */
int oracle_qp_dense_solve_batch(const oracle_qp_params *prm, int64_t batch, int n, int m, const double *P,
                                const double *q, const double *A, const double *l, const double *u,
                                const double *warm_x, const double *warm_y, double *x, double *y,
                                double *obj, uint32_t *iter, int32_t *code, int nthreads);

/*
 * Sparse branch (oracle/qp_sparse_oracle.c): QPSolver<QuadraticProgramSparse<double>>::solve for a
 * batch sharing one sparsity pattern.  P: CSC (Pp[n+1], Pi) as stored; A: CSR (Ap[m+1], Aj)
 * (qp.hpp:60-79); Px [batch][nnzP], Ax [batch][nnzA].  perm[n+m]: elimination order of the KKT
 * matrix (new -> old), NULL = natural.  nnzL_out (nullable): fill of the factor.
 */
int oracle_qp_sparse_solve_batch(const oracle_qp_params *prm, int64_t batch, int n, int m, const int32_t *Pp,
                                 const int32_t *Pi, const double *Px, const double *q, const int32_t *Ap,
                                 const int32_t *Aj, const double *Ax, const double *l, const double *u,
                                 const int32_t *perm, const double *warm_x, const double *warm_y, double *x,
                                 double *y, double *obj, uint32_t *iter, int32_t *code, int nthreads,
                                 int64_t *nnzL_out);
/*
 * Same with an explicit ACCUMULATION ORDER of the numeric factorisation: forder[n+m] (nullable) = rank of every
 * column of the permuted matrix; the sources of an entry of L are summed in ascending rank.  NULL = a postorder of
 * the elimination tree (children ascending).  Any order is a valid summation order; it is an input so that a
 * problem analysed WITHOUT its explicit zeros (the product's pruned plans) and the same problem with them can be
 * made to sum in the same order -- their elimination trees differ, hence their postorders.
 */
int oracle_qp_sparse_solve_batch_ordered(const oracle_qp_params *prm, int64_t batch, int n, int m, const int32_t *Pp,
                                         const int32_t *Pi, const double *Px, const double *q, const int32_t *Ap,
                                         const int32_t *Aj, const double *Ax, const double *l, const double *u,
                                         const int32_t *perm, const int32_t *forder, const double *warm_x,
                                         const double *warm_y, double *x, double *y, double *obj, uint32_t *iter,
                                         int32_t *code, int nthreads, int64_t *nnzL_out);

/*
 * The reference's verbose table (qp_solver.hpp:409-420, :490-501) as data instead of text: while a trace buffer is
 * set, the sparse batch calls record one row (ITER, OBJ, PRI_RES, DUA_RES, tolerance of PRI_RES, tolerance of DUA_RES
 * -- qp_solver.hpp:580-590) per stopping check of every item into trace[batch][cap][6]; unused rows have ITER = -1.  Process-global; clear with (NULL, 0).
 */
void oracle_qp_sparse_set_trace(double *trace, int cap);
/* the same for oracle_qp_dense_solve_batch: trace[batch][cap][6] */
void oracle_qp_dense_set_trace(double *trace, int cap);

/*
 * Restatement of Eigen 3.4 LDLT (unblocked, diagonal pivoting) exposed for unit tests.
 * W: k*k row-major work matrix, lower triangle (incl. diagonal) holds the symmetric input on
 * entry and L (unit, strictly lower) + D (diagonal) on exit.  tr[k]: transpositions.
 * Returns 1 on success (info()==Success), 0 on failure (NumericalIssue).
 */
int oracle_ldlt_factor(int k, double *W, int ld, int *tr);
void oracle_ldlt_solve(int k, const double *W, int ld, const int *tr, double *b);

/* EKF matrix part (oracle/ekf_oracle.c), ekf.hpp:84-102 (Euler substep) and :119-138. */
void oracle_ekf_predict(int dof, const double *A, const double *Q, double dt, double *P);
void oracle_ekf_predict_rk4(int dof, const double *A, const double *Q, double dt, double *P);
/* ... with the linearisation of every stage time (A0 at t, Am at t + dt/2, Ae at t + dt), as cov_ode re-evaluates it
 * (ekf.hpp:84-89) for dynamics that depend on t explicitly */
void oracle_ekf_predict_rk4_tv(int dof, const double *A0, const double *Am, const double *Ae, const double *Q, double dt,
                               double *P);
void oracle_ekf_predict_rk4_tv_batch(int64_t batch, int dof, const double *A0, const double *Am, const double *Ae,
                                     const double *Q, int q_shared, const double *dt, int dt_shared, double *P);
void oracle_ekf_predict_rk4_batch(int64_t batch, int dof, const double *A, const double *Q, int q_shared,
                                  const double *dt, int dt_shared, double *P);
int oracle_ekf_update(int dof, int ny, const double *H, const double *R, const double *r, double *P, double *delta);
void oracle_ekf_predict_batch(int64_t batch, int dof, const double *A, const double *Q, int q_shared, const double *dt,
                              int dt_shared, double *P);
void oracle_ekf_update_batch(int64_t batch, int dof, int ny, const double *H, const double *R, int r_shared,
                             const double *r, double *P, double *delta, int32_t *info);

#ifdef __cplusplus
}
#endif
#endif
