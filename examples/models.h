/* Example / test harness around the C++ host front (the headers under include/smooth_feedback_amd): concrete MPC
 * models with a C interface so that Python tests and bench.py can drive the host-side assembly.
 * NOT part of the product C-ABI (that is include/sfb.h); built into libsfb_models.so. */
#ifndef SFBX_MODELS_H
#define SFBX_MODELS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* variant 6 : X = Bundle<SE2,R3>, U = R2   -- the vehicle of examples/mpc_asif_vehicle.cpp:42-79
 * variant 12: X = Bundle<SE2,R3,SE2,R3>, U = R2 -- two such vehicles driven by one input pair
 *             (synthetic: matches BASELINE.json's "nx=12, nu=2" problem size n = m = 740 at K = 50) */
int sfbx_mpc_dims(int variant, int K, int *n, int *m, int *nnzP, int *nnzA, int *Nx, int *Nu, int *N);
/* pattern (+ P values, identical for all agents).  Arrays sized by sfbx_mpc_dims. */
int sfbx_mpc_pattern(int variant, int K, double tf, int32_t *Pp, int32_t *Pi, double *Pval, int32_t *Ap, int32_t *Aj);
/* elimination stages (n+m entries) suggested by the MPC front for the solver's ordering */
int sfbx_mpc_stage(int variant, int K, int32_t *stage);
/* Assemble `batch` agents: agent b runs at time t_b = 0.025*(b % 400) from x_b = xdes(t_b) (+) xi_b,
 * xi_b ~ U(-0.5,0.5)^Nx from std::mt19937_64(seed + b).  Aval [batch][nnzA], l,u [batch][m]. */
int sfbx_mpc_assemble_batch(int variant, int K, double tf, int64_t batch, uint64_t seed, double *Aval, double *l,
                            double *u, int threads);
/* Closed loop of tests/test_mpc.cpp:34-117 (SE2 state, R2 input, f = (u0, 0, u1), -1 <= u <= 1):
 * three consecutive MPC calls with warm start, then three without. u_out[6][2], codes[6]. Needs a GPU. */
int sfbx_test_mpc_se2(double *u_out, int32_t *codes, int32_t *traj_sizes);
/* tests/test_mpc.cpp:60-155 (StaticProperties as static_asserts, Api, Constructors) written against
 * <smooth/feedback/mpc.hpp> with only the Lie types renamed, plus the declaration of examples/mpc_asif_vehicle.cpp:64.
 * codes[12]: Api code0..3, Constructors code1..5, original-after-the-copy's-setter / fresh controller, the vehicle.
 * out[14]: [0] rel |u1 - u2|, [1] rel |u3 - u1|, [2] us.size() + 1 == xs.size(), [3] f.t_, [4] cr.t_, [5] pointer overload,
 * [6..9] rel |u1 - u2..5| of the copies / moves, [10] max |u| of the vehicle, [11] copies own their analyses,
 * [12] rel |original after set_udes on a COPY - fresh controller with that udes| (copies share the desired
 * trajectories, mpc.hpp:407, 607-608), [13] ... and that input differs from before.  Needs a GPU. */
int sfbx_test_mpc_api(double *out, int32_t *codes);
/* The host-only half of that front (no GPU): out[0] > 0 a setter on a COPY moves the original's assembly, out[1] == 0 every copy
 * sees the same desired trajectories, out[2] copies own their QP, out[3] no analysis travels with a copy, out[4] == 0 the const
 * assembly path leaves a by-reference functor's set_time alone, out[5] == 232 (Ncr 2, Nx 3, Nu 2), out[6] type properties. */
int sfbx_test_mpc_front_host(double *out);
/* MPC API beyond operator(): out[0] set_xdes_rel / set_udes_rel (mpc.hpp:539-586) vs the absolute-time setters (max abs
 * difference of A, l, u), out[1] the same controller with Time = std::chrono::steady_clock::time_point (time.hpp:25-89),
 * out[2..4] set_weights (mpc.hpp:593-598: stored, not transcribed; the constructor transcribes), out[5..8] lazy structure
 * refresh after set_xdes / set_udes and plan pinning (1.0 = as specified).  Host only (no GPU). */
int sfbx_test_mpc_time_and_setters(double *out);
/* examples/mpc_doubleintegrator.cpp:31-101 in closed loop for `ticks` ticks of 50 ms (time-invariant QP matrices: the
 * solver front flags every tick after the first as reuse_factor), then the same loop with the reuse switched off:
 * u_out / iters / codes [ticks] and u_ref / iters_ref [ticks]; reuse_count = flagged solves of the first loop;
 * seconds[2] = wall time of the two loops.  Needs a GPU. */
int sfbx_test_mpc_doubleintegrator(int ticks, double *u_out, uint32_t *iters, int32_t *codes, double *u_ref, uint32_t *iters_ref,
                                   int64_t *reuse_count, double *seconds);
/* tests/test_ocp_to_qp.cpp:41-107 with the GENERIC front ocp_to_qp() (include/smooth_feedback_amd/ocp_to_qp.hpp):
 * out[0..6] sizes (n, m, |q|, |l|, |u|, cols of P, rows of A), out[7..8] = min(A var - l), min(u - A var) for the exact
 * trajectory (:105-106), out[9..11] cost entries; solve != 0 (needs a GPU): out[12] status of solve_qp, out[13..17]
 * values of the qpsol_to_ocpsol() trajectory. */
int sfbx_test_ocp_to_qp_basic(double *out, int solve);
/* tests/test_ocp_to_qp.cpp:41-107 through the MPC transcription (double integrator, two intervals of 5 LGR nodes, tf = 2):
 * out = {min(A var - l), min(u - A var), N, n, m, intervals} for the exact parabola trajectory.  Host only (no GPU). */
int sfbx_test_ocp_to_qp_parabola(double *out);
/* tests/test_qp.cpp StaticProperties (:37-52, static_asserts in models.cpp), SolverAPI (:338-372), SparseSolverAPI
 * (:374-415) and PartialDynamic (:124-147) through the reference's include path <smooth/feedback/qp_solver.hpp> and
 * namespace: primal_dense / primal_sparse [5][2] = the five solvers' primal (original, copy, copy-assigned, moved,
 * move-assigned); primal_partial = {x0, x1, objective, hot-start x0, x1}.  Returns 0 on success.  Needs a GPU. */
int sfbx_test_qp_solver_api(double *primal_dense, double *primal_sparse, double *primal_partial);
/* QPSolver<sparse>: solve(A1), solve_batch(A2), solve(A1), solve(A1) on one solver (shared device workspace): out[0] third
 * == first bit for bit, out[1] / out[2] solves flagged reuse_factor after the third / fourth call (0 / 1), out[3] fourth
 * == first, out[4..7] primal of first and of the batch call.  Needs a GPU. */
int sfbx_test_solve_after_solve_batch(double *out);
/* Swarm tick through MPCSwarm (host assembly + one batched GPU solve): returns u0 [batch][2], codes. */
int sfbx_mpc_swarm_step(int variant, int K, double tf, int64_t batch, uint64_t seed, int ticks, double *u0,
                        int32_t *codes, uint32_t *iters);
/* The same swarm with every batched solve sharded over `devices` (sfb_set_devices; an ordinal may repeat) from this
 * one process: one host thread, plan upload and workspace per device.  Same outputs as sfbx_mpc_swarm_step. */
int sfbx_mpc_swarm_step_multi(int variant, int K, double tf, int64_t batch, uint64_t seed, int ticks, const int *devices,
                              int ndev, double *u0, int32_t *codes, uint32_t *iters);
/* Device-side assembly (sfb_mpc_assemble_batch / sfb_mpc_swarm, include/sfb.h): the layout of the variant's
 * transcription -- dims = {nx, nu, ncr, kmesh, nivals, nparts}, alpha [nivals], D [(kmesh+1)*kmesh], kind / dof
 * [nparts], crl / cru [ncr] --, the linearisation records of the agents of sfbx_mpc_assemble_batch
 * (rec [batch][record doubles]) and the swarm tick of sfbx_mpc_swarm_step through MPCSwarmDevice. */
int sfbx_mpc_layout(int variant, int K, double tf, int32_t *dims, double *alpha, double *D, int32_t *kind,
                    int32_t *dof, double *crl, double *cru);
int sfbx_mpc_records(int variant, int K, double tf, int64_t batch, uint64_t seed, double *rec, int threads);
int sfbx_mpc_swarm_device_step(int variant, int K, double tf, int64_t batch, uint64_t seed, int ticks, double *u0,
                               int32_t *codes, uint32_t *iters);
/* wall seconds of every swarm.step() of the last sfbx_mpc_swarm[_device]_step call; returns their number */
int sfbx_last_tick_seconds(double *out, int n);
/* group identities for the tests: returns max abs error over a set of checks */
double sfbx_lie_selftest(void);
/* EKF<G> front (include/smooth_feedback_amd/ekf.hpp) against the reference's own checks: PredictTimeCut
 * (tests/test_ekf.cpp:155-180), UpdateLinear (:50-103, R^3 / Ny 3) and an SE2 predict+update smoke.
 * Returns 0 and writes max errors: err[0] time-cut, err[1] linear update state, err[2] linear update cov. Needs a GPU. */
int sfbx_test_ekf(double *err);
/* PredictLinear (tests/test_ekf.cpp:104-153) with EKF<R^Nx, RK4>, dt = 1e-3, tau = 0.7, Q = 0, Nx in {3, 6}:
 * err[0] = max relative error of the estimate vs expm(A tau) xhat, err[1] = of the covariance vs F P F'.
 * The exact F (Nx*Nx, column-major, for Nx = 3 then 6) is supplied by the caller.  Needs a GPU. */
int sfbx_test_ekf_predict_linear(const double *A3, const double *F3, const double *A6, const double *F6, double *err);
/* the third size of the reference's test, Nx = 9 (tests/test_ekf.cpp:152): the generic one-filter-per-wave kernel */
int sfbx_test_ekf_predict_linear9(const double *A9, const double *F9, double *err);
/* asif_to_qp() (include/smooth_feedback_amd/asif.hpp) for the case of tests/test_asif.cpp:37-95: X = SE2, f = (u0, 0, u1),
 * h = position (nh = 2), bu = (-0.1, 1), K = 3, input box [-1,1]^2 around c = 0, T = 1, alpha = 1, dt = 0.1.
 * x0 = (angle, px, py).  Out, column-major: P[9] q[3] A[9*3] l[9] u[9].  Host only (no GPU). */
int sfbx_asif_basic_qp(const double *x0, const double *udes, double *P, double *q, double *A, double *l, double *u);
/* ASIFilter on the GPU.  which = 0: the SO3 filter of tests/test_asif.cpp:103-131 (K = 100, nh = 3: n = 4, m = 301,
 * sparse-kernel path); which = 1: the vehicle filter of examples/mpc_asif_vehicle.cpp:95-129 (K = 200: n = 3, m = 203,
 * polish off); which = 2: the same vehicle with the default K = 10 (n = 3, m = 13: dense-kernel path).
 * Out: u[3], code, iter, and the QP that was solved (sized by dims[0] = n, dims[1] = m) with its primal/dual. */
int sfbx_test_asif(int which, double *u_out, int32_t *code, uint32_t *iter, int32_t *dims, double *P, double *q,
                   double *A, double *l, double *u, double *x, double *y);
/* ASIFSwarm: `batch` vehicles (state = xdes(t_b) (+) xi_b as in sfbx_mpc_assemble_batch, u_des ~ U(-0.5,0.5)^2),
 * K constraint instances, `ticks` consecutive calls (warm start from tick 2).  Out for the LAST tick: filtered
 * u [batch][2], codes, iters, the QPs [batch][...] (n = 3, m = K + 3) and their primal/dual solutions. */
int sfbx_asif_swarm_step(int64_t batch, uint64_t seed, int K, int ticks, double *u_out, int32_t *codes, uint32_t *iters,
                         double *P, double *q, double *A, double *l, double *u, double *x, double *y, double *wx,
                         double *wy);
/* the states (x, y, cos, sin, v0, v1, v2) [batch][7] and desired inputs [batch][2] sfbx_asif_swarm_step starts from */
int sfbx_asif_swarm_states(int64_t batch, uint64_t seed, double *states, double *udes);
/* vehicle EKFs, one host EKF<> object per filter: `steps` x (predict(Q, tau, dt) + update(y[step], R)); the host twin of
 * sfbx_ekf_swarm_device (models_device.hip) */
int sfbx_ekf_swarm_host(int64_t batch, int steps, int rk4, double tau, double dt, const double *states, const double *P0,
                        const double *y, double *states_out, double *P_out);
/* mesh: nodes (N+1), weights (N+1), Dus ((K+1)*K col-major) for `n` intervals of K points */
int sfbx_mesh(int n_ivals, int K, double *nodes, double *weights, double *Dus);

#ifdef __cplusplus
}
#endif
#endif
