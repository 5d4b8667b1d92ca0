/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * CPU restatement of the matrix part of smooth::feedback::EKF (pettni/smooth_feedback @ v1,
 * include/smooth/feedback/ekf.hpp): the covariance ODE right-hand side + one explicit Euler step
 * (predict, :84-89 with the default boost::numeric::odeint::euler stepper :30,:96) and the Kalman
 * update (:119-138, Eigen LDLT of the innovation covariance).  The Lie-group linearisation
 * (A = -ad(f) + d^r f/dx, H = d^r h/dx, innovation y (-) h) stays on the host, exactly as in the
 * device path.  Matrices are column-major (Eigen default), item-major contiguous.
 *
 * PARITY STATUS: pinned by the reference's own analytic checks (tests/test_ekf.cpp:50-103 linear
 * Kalman update identities, :105-153 propagation against expm, with many small Euler steps and with the
 * test's own runge_kutta4 stepper) in tests/test_oracle_ekf.py; summation orders inside Eigen's small products are unpinned, the fixed
 * order used here (k ascending, fma) is shared with the HIP kernel.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "qp_oracle.h"

/* P <- P + dt * symU(A P + P A' + Q)      ekf.hpp:88 and euler::do_step (x += dt * dxdt) */
void oracle_ekf_predict(int dof, const double *A, const double *Q, double dt, double *P)
{
  const int n = dof;
  double *dP  = (double *)malloc(sizeof(double) * (size_t)n * n);
  for (int j = 0; j < n; ++j)
    for (int i = 0; i <= j; ++i) {
      double m1 = 0.0, m2 = 0.0;
      for (int k = 0; k < n; ++k) m1 = fma(A[i + k * n], P[k + j * n], m1); /* (A P)_ij  */
      for (int k = 0; k < n; ++k) m2 = fma(P[i + k * n], A[j + k * n], m2); /* (P A')_ij */
      const double s = (m1 + m2) + Q[i + j * n];
      dP[i + j * n]  = s;
      dP[j + i * n]  = s; /* selfadjointView<Upper> */
    }
  for (int e = 0; e < n * n; ++e) P[e] = P[e] + dt * dP[e];
  free(dP);
}

/* Covariance right-hand side  dP = symU(A P + P A' + Q)  (ekf.hpp:84-89), all n*n entries written */
static void ekf_cov_rhs(int n, const double *A, const double *Q, const double *P, double *dP)
{
  for (int j = 0; j < n; ++j)
    for (int i = 0; i <= j; ++i) {
      double m1 = 0.0, m2 = 0.0;
      for (int k = 0; k < n; ++k) m1 = fma(A[i + k * n], P[k + j * n], m1);
      for (int k = 0; k < n; ++k) m2 = fma(P[i + k * n], A[j + k * n], m2);
      const double s = (m1 + m2) + Q[i + j * n];
      dP[i + j * n]  = s;
      dP[j + i * n]  = s;
    }
}

/* One step of boost::numeric::odeint::runge_kutta4 (the stepper tests/test_ekf.cpp:113-115 instantiates;
 * explicit_generic_rk with a = {1/2; 0,1/2; 0,0,1}, b = {1/6,1/3,1/3,1/6}) on the covariance ODE with ONE A
 * for all stages (time-invariant dynamics; oracle_ekf_predict_rk4_tv below takes the A of every stage time).  Stage states  P + (dt a_ij) k_j ; result  P + (dt b1) k1 + (dt b2) k2 + (dt b3) k3
 * + (dt b4) k4  accumulated left to right (odeint's scale_sum5).  Zero tableau entries add exact zeros. */
void oracle_ekf_predict_rk4(int dof, const double *A, const double *Q, double dt, double *P)
{
  oracle_ekf_predict_rk4_tv(dof, A, A, A, Q, dt, P);
}

/* The same with the linearisation re-evaluated at the stage times as cov_ode does (ekf.hpp:84-89): A0 at t,
 * Am at t + dt/2 (stages 2 and 3), Ae at t + dt (stage 4).  The state g is frozen during the covariance step. */
void oracle_ekf_predict_rk4_tv(int dof, const double *A0, const double *Am, const double *Ae, const double *Q, double dt,
                               double *P)
{
  const int n = dof, nn = dof * dof;
  double *k  = (double *)malloc(sizeof(double) * (size_t)nn);
  double *Pi = (double *)malloc(sizeof(double) * (size_t)nn);
  double *S  = (double *)malloc(sizeof(double) * (size_t)nn);
  const double b1 = dt * (1.0 / 6.0), b2 = dt * (1.0 / 3.0), c2 = dt * 0.5, c4 = dt * 1.0;
  ekf_cov_rhs(n, A0, Q, P, k);                                  /* k1 */
  for (int e = 0; e < nn; ++e) { S[e] = P[e] + b1 * k[e]; Pi[e] = P[e] + c2 * k[e]; }
  ekf_cov_rhs(n, Am, Q, Pi, k);                                 /* k2 */
  for (int e = 0; e < nn; ++e) { S[e] = S[e] + b2 * k[e]; Pi[e] = P[e] + c2 * k[e]; }
  ekf_cov_rhs(n, Am, Q, Pi, k);                                 /* k3 */
  for (int e = 0; e < nn; ++e) { S[e] = S[e] + b2 * k[e]; Pi[e] = P[e] + c4 * k[e]; }
  ekf_cov_rhs(n, Ae, Q, Pi, k);                                 /* k4 */
  for (int e = 0; e < nn; ++e) P[e] = S[e] + b1 * k[e];
  free(k); free(Pi); free(S);
}

/* ekf.hpp:119-138.  H: ny x dof, R: ny x ny (upper used), r = y (-) h(g): ny.  Outputs: delta (dof)
 * = K r (to be applied as g (+) delta on the host), P updated in place. Returns LDLT info (1 ok). */
int oracle_ekf_update(int dof, int ny, const double *H, const double *R, const double *r, double *P, double *delta)
{
  const int n = dof, m = ny;
  double *T  = (double *)malloc(sizeof(double) * (size_t)m * n); /* H * Psym   */
  double *HP = (double *)malloc(sizeof(double) * (size_t)m * n); /* H * P      */
  double *W  = (double *)calloc((size_t)m * m, sizeof(double));  /* S, row-major lower == upper of Eigen */
  double *X  = (double *)malloc(sizeof(double) * (size_t)m * n); /* S^-1 (H P) */
  double *IK = (double *)malloc(sizeof(double) * (size_t)n * n);
  double *Pn = (double *)malloc(sizeof(double) * (size_t)n * n);
  int *tr    = (int *)malloc(sizeof(int) * (size_t)m);
  double *col = (double *)malloc(sizeof(double) * (size_t)m);
#define PSYM(i, j) ((i) <= (j) ? P[(i) + (j) * n] : P[(j) + (i) * n])
  for (int j = 0; j < n; ++j)
    for (int a = 0; a < m; ++a) {
      double s1 = 0.0, s2 = 0.0;
      for (int k = 0; k < n; ++k) s1 = fma(H[a + k * m], PSYM(k, j), s1);
      for (int k = 0; k < n; ++k) s2 = fma(H[a + k * m], P[k + j * n], s2);
      T[a + j * m]  = s1;
      HP[a + j * m] = s2;
    }
  for (int b = 0; b < m; ++b) /* :129-130  S = triU(H Psym H' + R) */
    for (int a = 0; a <= b; ++a) {
      double s = 0.0;
      for (int k = 0; k < n; ++k) s = fma(T[a + k * m], H[b + k * m], s);
      W[b * m + a] = s + R[a + b * m]; /* lower(b,a) == upper(a,b) */
    }
  const int ok = oracle_ldlt_factor(m, W, m, tr); /* :134  S.selfadjointView<Upper>().ldlt() */
  for (int j = 0; j < n; ++j) {                     /* .solve(H * P_) column by column */
    for (int a = 0; a < m; ++a) col[a] = HP[a + j * m];
    oracle_ldlt_solve(m, W, m, tr, col);
    for (int a = 0; a < m; ++a) X[a + j * m] = col[a];
  }
  /* K = X' (dof x ny);  delta = K r  (:137) */
  for (int i = 0; i < n; ++i) {
    double s = 0.0;
    for (int a = 0; a < m; ++a) s = fma(X[a + i * m], r[a], s);
    delta[i] = s;
  }
  /* P = symU((I - K H) P)  (:138) */
  for (int k = 0; k < n; ++k)
    for (int i = 0; i < n; ++i) {
      double s = 0.0;
      for (int a = 0; a < m; ++a) s = fma(X[a + i * m], H[a + k * m], s);
      IK[i + k * n] = ((i == k) ? 1.0 : 0.0) - s;
    }
  for (int j = 0; j < n; ++j)
    for (int i = 0; i <= j; ++i) {
      double s = 0.0;
      for (int k = 0; k < n; ++k) s = fma(IK[i + k * n], P[k + j * n], s);
      Pn[i + j * n] = s;
      Pn[j + i * n] = s;
    }
  memcpy(P, Pn, sizeof(double) * (size_t)n * n);
#undef PSYM
  free(T); free(HP); free(W); free(X); free(IK); free(Pn); free(tr); free(col);
  return ok;
}

/* batch drivers: item-major contiguous arrays; q_shared / r_shared / dt_shared: one value for all */
void oracle_ekf_predict_batch(int64_t batch, int dof, const double *A, const double *Q, int q_shared, const double *dt,
                              int dt_shared, double *P)
{
  const size_t nn = (size_t)dof * dof;
  for (int64_t b = 0; b < batch; ++b)
    oracle_ekf_predict(dof, A + (size_t)b * nn, q_shared ? Q : Q + (size_t)b * nn, dt_shared ? dt[0] : dt[b],
                       P + (size_t)b * nn);
}
void oracle_ekf_predict_rk4_batch(int64_t batch, int dof, const double *A, const double *Q, int q_shared,
                                  const double *dt, int dt_shared, double *P)
{
  const size_t nn = (size_t)dof * dof;
  for (int64_t b = 0; b < batch; ++b)
    oracle_ekf_predict_rk4(dof, A + (size_t)b * nn, q_shared ? Q : Q + (size_t)b * nn, dt_shared ? dt[0] : dt[b],
                           P + (size_t)b * nn);
}
void oracle_ekf_predict_rk4_tv_batch(int64_t batch, int dof, const double *A0, const double *Am, const double *Ae,
                                     const double *Q, int q_shared, const double *dt, int dt_shared, double *P)
{
  const size_t nn = (size_t)dof * dof;
  for (int64_t b = 0; b < batch; ++b)
    oracle_ekf_predict_rk4_tv(dof, A0 + (size_t)b * nn, Am + (size_t)b * nn, Ae + (size_t)b * nn,
                              q_shared ? Q : Q + (size_t)b * nn, dt_shared ? dt[0] : dt[b], P + (size_t)b * nn);
}
void oracle_ekf_update_batch(int64_t batch, int dof, int ny, const double *H, const double *R, int r_shared,
                             const double *r, double *P, double *delta, int32_t *info)
{
  const size_t nn = (size_t)dof * dof, mn = (size_t)ny * dof, mm = (size_t)ny * ny;
  for (int64_t b = 0; b < batch; ++b) {
    const int ok = oracle_ekf_update(dof, ny, H + (size_t)b * mn, r_shared ? R : R + (size_t)b * mm, r + (size_t)b * ny,
                                     P + (size_t)b * nn, delta + (size_t)b * dof);
    if (info) info[b] = ok ? 0 : 1;
  }
}
