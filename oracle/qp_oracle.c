/*
 * ORACLE (test infrastructure, NOT product code) -- see qp_oracle.h for scope and parity status.
 *
 * Plain-C restatement of smooth::feedback::QPSolver<QuadraticProgram<M,N,double>> (dense path).
 * All citations are file:line in /root/reference (pettni/smooth_feedback @ v1).
 *
 * Floating-point discipline (shared with the HIP kernel so both agree bit-for-bit):
 *   - compiled with -ffp-contract=off: every a*b+c written below rounds twice unless it is
 *     spelled fma();
 *   - dot products / triangular-solve accumulations are fma() chains in ONE fixed order,
 *     stated at each site (Eigen's internal orders are not reproducible without Eigen);
 *   - element-wise expressions keep the association of the Eigen expression templates
 *     (SURVEY.md section 8 "Rounding-order notes").
 */
#include "qp_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* std::max / std::min semantics (first argument wins ties / NaN) */
static inline double dmax(double a, double b) { return (a < b) ? b : a; }
static inline double dmin(double a, double b) { return (b < a) ? b : a; }

void oracle_qp_params_default(oracle_qp_params *p)
{
  /* qp_solver.hpp:29-68 */
  p->alpha           = 1.6f;
  p->rho             = 0.1f;
  p->sigma           = 1e-6f;
  p->scaling         = 1;
  p->eps_abs         = 1e-3f;
  p->eps_rel         = 1e-3f;
  p->eps_primal_inf  = 1e-4f;
  p->eps_dual_inf    = 1e-4f;
  p->max_iter        = -1;
  p->max_time_ns     = -1;
  p->stop_check_iter = 25;
  p->polish          = 1;
  p->polish_iter     = 5;
  p->delta           = 1e-6f;
  p->verbose         = 0;
  p->reuse_factor    = 0;
}

/* ------------------------------------------------------------------------------------------
 * Eigen 3.4.0 LDLT<MatrixXd, Upper> restated (third-party, absent from /root/reference; call
 * sites qp_solver.hpp:259,428,462 and :187-188).  Eigen factorises the transposed view as a
 * lower-triangular problem (ldlt_inplace<Lower>::unblocked), which is what is written here:
 * W is row-major with leading dimension ld, W(i,j), j<=i, holds the symmetric matrix.
 *
 *   for k = 0..size-1:
 *     pivot = first index of the largest |W(i,i)|, i >= k   (the trailing diagonal is NOT
 *             updated by earlier steps: the algorithm is left-looking)
 *     symmetric swap k <-> pivot restricted to the lower triangle
 *     temp(j)  = D(j) * L(k,j)                       j < k
 *     D(k)     = W(k,k) - sum_j L(k,j) * temp(j)     (dot product first, then subtract)
 *     L(i,k)   = (W(i,k) - sum_j L(i,j)*temp(j)) / D(k)   i > k
 *     zero pivot: column below must be zero, and no non-zero pivot may follow, else failure.
 *
 * Summation order (this oracle's choice): s = 0; for j ascending: s = fma(L(.,j), temp(j), s).
 * ---------------------------------------------------------------------------------------- */
#define WM(i, j) W[(size_t)(i) * (size_t)ld + (size_t)(j)]

int oracle_ldlt_factor(int k, double *W, int ld, int *tr)
{
  if (k <= 1) {
    if (k == 1) tr[0] = 0;
    return 1;
  }
  int found_zero_pivot = 0, ret = 1;
  double *temp = (double *)malloc(sizeof(double) * (size_t)k);
  if (!temp) return 0;

  for (int kk = 0; kk < k; ++kk) {
    /* mat.diagonal().tail(size-k).cwiseAbs().maxCoeff(&idx): strict '>' => first maximum */
    int p       = kk;
    double best = fabs(WM(kk, kk));
    for (int i = kk + 1; i < k; ++i) {
      const double a = fabs(WM(i, i));
      if (a > best) {
        best = a;
        p    = i;
      }
    }
    tr[kk] = p;
    if (p != kk) {
      for (int t = 0; t < kk; ++t) {
        const double tmp = WM(kk, t);
        WM(kk, t)        = WM(p, t);
        WM(p, t)         = tmp;
      }
      for (int i = p + 1; i < k; ++i) {
        const double tmp = WM(i, kk);
        WM(i, kk)        = WM(i, p);
        WM(i, p)         = tmp;
      }
      {
        const double tmp = WM(kk, kk);
        WM(kk, kk)       = WM(p, p);
        WM(p, p)         = tmp;
      }
      for (int i = kk + 1; i < p; ++i) {
        const double tmp = WM(i, kk);
        WM(i, kk)        = WM(p, i);
        WM(p, i)         = tmp;
      }
    }

    const int rs = k - kk - 1;
    if (kk > 0) {
      for (int j = 0; j < kk; ++j) temp[j] = WM(j, j) * WM(kk, j);
      double s = 0.0;
      for (int j = 0; j < kk; ++j) s = fma(WM(kk, j), temp[j], s);
      WM(kk, kk) -= s;
      for (int i = kk + 1; i < k; ++i) {
        double t = 0.0;
        for (int j = 0; j < kk; ++j) t = fma(WM(i, j), temp[j], t);
        WM(i, kk) -= t;
      }
    }

    const double akk = WM(kk, kk);
    const int valid  = fabs(akk) > 0.0;

    if (kk == 0 && !valid) {
      for (int j = 0; j < k; ++j) {
        tr[j] = j;
        for (int i = j + 1; i < k; ++i) ret = ret && (WM(i, j) == 0.0);
      }
      free(temp);
      return ret;
    }

    if (rs > 0 && valid) {
      for (int i = kk + 1; i < k; ++i) WM(i, kk) /= akk;
    } else if (rs > 0) {
      for (int i = kk + 1; i < k; ++i) ret = ret && (WM(i, kk) == 0.0);
    }

    if (found_zero_pivot && valid) {
      ret = 0;
    } else if (!valid) {
      found_zero_pivot = 1;
    }
  }
  free(temp);
  return ret;
}

/* LDLT::_solve_impl: dst = P b; L^-1; D^-1 (|d| <= DBL_MIN -> 0, true division); L^-T; P^T.
 * Order: forward row i: s=b_i; j ascending: s = fma(-L(i,j), x_j, s).
 *        backward row i: s=b_i; j DEscending from k-1 to i+1: s = fma(-L(j,i), x_j, s). */
void oracle_ldlt_solve(int k, const double *W, int ld, const int *tr, double *b)
{
  for (int i = 0; i < k; ++i) {
    const int p = tr[i];
    if (p != i) {
      const double t = b[i];
      b[i]           = b[p];
      b[p]           = t;
    }
  }
  for (int i = 0; i < k; ++i) {
    double s = b[i];
    for (int j = 0; j < i; ++j) s = fma(-WM(i, j), b[j], s);
    b[i] = s;
  }
  for (int i = 0; i < k; ++i) {
    const double d = WM(i, i);
    if (fabs(d) > DBL_MIN) {
      b[i] /= d;
    } else {
      b[i] = 0.0;
    }
  }
  for (int i = k - 1; i >= 0; --i) {
    double s = b[i];
    for (int j = k - 1; j > i; --j) s = fma(-WM(j, i), b[j], s);
    b[i] = s;
  }
  for (int i = k - 1; i >= 0; --i) {
    const int p = tr[i];
    if (p != i) {
      const double t = b[i];
      b[i]           = b[p];
      b[p]           = t;
    }
  }
}
#undef WM

/* ------------------------------------------------------------------------------------------ */

typedef struct {
  int n, m, k;
  const double *P, *q, *A, *l, *u; /* col-major P(n x n), A(m x n)  -- qp.hpp:31-45 */
  oracle_qp_params prm;
  /* QPSolver members, qp_solver.hpp:733-756 */
  double c;
  double *sx, *sx_inc, *sy, *sy_inc;
  double *z, *z_next, *rho, *p;
  double *x_us, *dx_us, *Px, *Aty, *Ax, *y_us, *z_us, *dy_us;
  double *H; /* k x k row-major lower (== transposed view of Eigen's Upper) */
  int *tr;
  double *primal, *dual; /* sol_.primal, sol_.dual */
} qp_work;

#define PM(i, j) w->P[(size_t)(i) + (size_t)(j) * (size_t)w->n]
#define AM(i, j) w->A[(size_t)(i) + (size_t)(j) * (size_t)w->m]

/* QPSolver::scale, qp_solver.hpp:673-730 */
static void qp_scale(qp_work *w)
{
  const int n = w->n, m = w->m;
  for (int j = 0; j < n; ++j) w->sx[j] = 1.0;     /* :675 */
  for (int i = 0; i < m; ++i) w->sy[i] = 1.0;     /* :676 */
  for (int j = 0; j < n; ++j) w->sx_inc[j] = 0.0; /* :678 */

  /* :681-685 column inf-norms of P (InnerIterator over a dense col-major matrix visits all
   * entries of column i) */
  for (int col = 0; col < n; ++col)
    for (int row = 0; row < n; ++row) w->sx_inc[col] = dmax(w->sx_inc[col], fabs(PM(row, col)));
  for (int i = 0; i < n; ++i)
    if (w->sx_inc[i] == 0.0) w->sx_inc[i] = 1.0; /* :688-690 */

  /* :693  c = 1 / max({1e-6, mean(sx_inc), ||q||inf}); mean = (sequential sum)/n */
  double sum = w->sx_inc[0];
  for (int j = 1; j < n; ++j) sum += w->sx_inc[j];
  const double mean = sum / (double)n;
  double qn         = 0.0;
  for (int j = 0; j < n; ++j) qn = dmax(qn, fabs(w->q[j]));
  w->c = 1.0 / dmax(dmax(1e-6, mean), qn);

  int iter = 0;
  double crit;
  do { /* :698-729 */
    for (int j = 0; j < n; ++j) w->sx_inc[j] = 0.0;
    for (int i = 0; i < m; ++i) w->sy_inc[i] = 0.0;
    for (int col = 0; col < n; ++col)
      for (int row = 0; row < n; ++row) /* :704-707  ((c*sx_r)*sx_c)*P_rc */
        w->sx_inc[col] = dmax(w->sx_inc[col], fabs(w->c * w->sx[row] * w->sx[col] * PM(row, col)));
    for (int col = 0; col < n; ++col)
      for (int row = 0; row < m; ++row) { /* :712-714  (sy_r*sx_c)*A_rc */
        const double Aij = fabs(w->sy[row] * w->sx[col] * AM(row, col));
        w->sx_inc[col]   = dmax(w->sx_inc[col], Aij);
        w->sy_inc[row]   = dmax(w->sy_inc[row], Aij);
      }
    for (int j = 0; j < n; ++j)
      if (w->sx_inc[j] == 0.0) w->sx_inc[j] = 1.0; /* :719-721 */
    for (int i = 0; i < m; ++i)
      if (w->sy_inc[i] == 0.0) w->sy_inc[i] = 1.0; /* :722-724 */
    /* :726-727  sx <- sqrt(1/max(inc,1e-8)) * sx  (cwiseInverse then cwiseSqrt) */
    for (int j = 0; j < n; ++j) w->sx[j] = sqrt(1.0 / dmax(w->sx_inc[j], 1e-8)) * w->sx[j];
    for (int i = 0; i < m; ++i) w->sy[i] = sqrt(1.0 / dmax(w->sy_inc[i], 1e-8)) * w->sy[i];
    double a = 0.0, b = 0.0; /* maxCoeff of |inc - 1| */
    for (int j = 0; j < n; ++j) a = dmax(a, fabs(w->sx_inc[j] - 1.0));
    for (int i = 0; i < m; ++i) b = dmax(b, fabs(w->sy_inc[i] - 1.0));
    crit = dmax(a, b);
  } while (iter++ < 10 && crit > 0.1); /* :728-729 */
}

static double norm_inf(const double *v, int len)
{
  double r = 0.0;
  for (int i = 0; i < len; ++i) r = dmax(r, fabs(v[i]));
  return r;
}

/* mat-vec orders (oracle's fixed choice): s = 0; inner index ascending; s = fma(M_ij, v_j, s) */
static void mv_A(const qp_work *w, const double *v, double *out) /* out = A v */
{
  for (int i = 0; i < w->m; ++i) {
    double s = 0.0;
    for (int j = 0; j < w->n; ++j) s = fma(AM(i, j), v[j], s);
    out[i] = s;
  }
}
static void mv_At(const qp_work *w, const double *v, double *out) /* out = A' v */
{
  for (int j = 0; j < w->n; ++j) {
    double s = 0.0;
    for (int i = 0; i < w->m; ++i) s = fma(AM(i, j), v[i], s);
    out[j] = s;
  }
}
static void mv_P(const qp_work *w, const double *v, double *out) /* out = P v (P as stored, full) */
{
  for (int i = 0; i < w->n; ++i) {
    double s = 0.0;
    for (int j = 0; j < w->n; ++j) s = fma(PM(i, j), v[j], s);
    out[i] = s;
  }
}

/* QPSolver::check_stopping, qp_solver.hpp:574-644.  Returns -1 for std::nullopt. */
/* The reference's verbose table (qp_solver.hpp:490-501) as data, like oracle_qp_sparse_set_trace: while a trace buffer is set,
 * every stopping check of every item appends (ITER, OBJ, PRI_RES, DUA_RES, tolerance of PRI_RES, tolerance of DUA_RES) to
 * trace[item][cap][6]; unused rows keep ITER = -1.  Process-global; clear with (NULL, 0). */
static double *g_dense_trace            = NULL;
static int g_dense_trace_cap            = 0;
static __thread double *tl_dense_trace  = NULL; /* the rows of the item the calling thread is solving */
static __thread int tl_dense_trace_rows = 0;
void oracle_qp_dense_set_trace(double *trace, int cap) { g_dense_trace = trace; g_dense_trace_cap = cap; }

static int qp_check_stopping(qp_work *w)
{
  const int n = w->n, m = w->m;
  const double inf          = INFINITY;
  const oracle_qp_params *p = &w->prm;

  /* OPTIMALITY :584-594 */
  mv_A(w, w->x_us, w->Ax);
  const double Ax_norm = norm_inf(w->Ax, m);
  for (int i = 0; i < m; ++i) w->Ax[i] -= w->z_us[i];
  if (norm_inf(w->Ax, m) <= (double)p->eps_abs + (double)p->eps_rel * dmax(Ax_norm, norm_inf(w->z_us, m))) {
    mv_P(w, w->x_us, w->Px);
    mv_At(w, w->y_us, w->Aty);
    const double dual_scale = dmax(dmax(norm_inf(w->Px, n), norm_inf(w->q, n)), norm_inf(w->Aty, n));
    for (int j = 0; j < n; ++j) w->Px[j] += w->q[j] + w->Aty[j]; /* :592  Px += (q + Aty) */
    if (norm_inf(w->Px, n) <= (double)p->eps_abs + (double)p->eps_rel * dual_scale) return ORACLE_QP_OPTIMAL;
  }

  /* PRIMAL INFEASIBILITY :598-621 */
  mv_At(w, w->dy_us, w->Aty);
  const double Edy_norm = norm_inf(w->dy_us, m);
  double s              = 0.0;
  for (int i = 0; i < m; ++i) {
    if (w->u[i] != inf) {
      s += w->u[i] * dmax(0.0, w->dy_us[i]);
    } else if (w->dy_us[i] > (double)p->eps_primal_inf * Edy_norm) {
      s = inf;
      break;
    }
    if (w->l[i] != -inf) {
      s += w->l[i] * dmin(0.0, w->dy_us[i]);
    } else if (w->dy_us[i] < (double)(-p->eps_primal_inf) * Edy_norm) {
      s = inf;
      break;
    }
  }
  if (dmax(norm_inf(w->Aty, n), s) < (double)p->eps_primal_inf * Edy_norm) return ORACLE_QP_PRIMAL_INFEASIBLE;

  /* DUAL INFEASIBILITY :625-641 */
  mv_A(w, w->dx_us, w->Ax);
  const double dx_norm = norm_inf(w->dx_us, n);
  mv_P(w, w->dx_us, w->Px);
  double qdx = 0.0; /* q.dot(dx): ascending fma chain */
  for (int j = 0; j < n; ++j) qdx = fma(w->q[j], w->dx_us[j], qdx);
  const double thr = (double)p->eps_dual_inf * dx_norm;
  int dual_inf     = (norm_inf(w->Px, n) <= thr) && (qdx <= thr);
  for (int i = 0; i < m && dual_inf; ++i) {
    if (w->u[i] == inf) {
      dual_inf &= (w->Ax[i] >= (double)(-p->eps_dual_inf) * dx_norm);
    } else if (w->l[i] == -inf) {
      dual_inf &= (w->Ax[i] <= thr);
    } else {
      dual_inf &= (fabs(w->Ax[i]) < thr);
    }
  }
  if (dual_inf) return ORACLE_QP_DUAL_INFEASIBLE;
  return -1;
}

/* detail::polish_qp, qp_solver.hpp:92-204 (dense branch).  Operates on the SCALED primal/dual.
 * Returns 1 on success, 0 if the LDLT of the perturbed matrix failed (:190). */
static int qp_polish(qp_work *w)
{
  const int n = w->n, m = w->m;
  const double inf = INFINITY, eps = DBL_EPSILON;

  /* :113-123 active sets (lower first, then upper) */
  int nl = 0, nu = 0;
  for (int i = 0; i < m; ++i) {
    if (w->dual[i] < -100 * eps && w->l[i] != -inf) nl++;
    if (w->dual[i] > 100 * eps && w->u[i] != inf) nu++;
  }
  const int na = nl + nu, K = n + na;
  int *LU   = (int *)malloc(sizeof(int) * (size_t)(na > 0 ? na : 1));
  for (int i = 0, lc = 0, uc = 0; i < m; ++i) {
    if (w->dual[i] < -100 * eps && w->l[i] != -inf) LU[lc++] = i;
    if (w->dual[i] > 100 * eps && w->u[i] != inf) LU[nl + uc++] = i;
  }

  /* H upper triangle (:159-165) kept as a full symmetric K x K array Hs for the residual,
   * Hp = H + diag(delta,..,-delta,..) (:174-177) in row-major lower form for the LDLT. */
  double *Hs = (double *)calloc((size_t)K * (size_t)K, sizeof(double));
  double *Hp = (double *)calloc((size_t)K * (size_t)K, sizeof(double));
  double *h  = (double *)malloc(sizeof(double) * (size_t)K);
  double *t  = (double *)calloc((size_t)K, sizeof(double));
  double *r  = (double *)malloc(sizeof(double) * (size_t)K);
  int *tr    = (int *)malloc(sizeof(int) * (size_t)K);
  const double delta = (double)w->prm.delta;

  for (int i = 0; i < n; ++i)
    for (int j = i; j < n; ++j) { /* upper entry (i,j): ((c*sx_i)*P_ij)*sx_j  (:161) */
      const double v           = w->c * w->sx[i] * PM(i, j) * w->sx[j];
      Hs[(size_t)i * K + j]    = v;
      Hs[(size_t)j * K + i]    = v;
      Hp[(size_t)j * K + i]    = v; /* lower(j,i) == upper(i,j) */
    }
  for (int a = 0; a < na; ++a) {
    const int row = LU[a];
    for (int j = 0; j < n; ++j) { /* H(j, n+a) = (sy_r*A_rj)*sx_j  (:163) */
      const double v              = w->sy[row] * AM(row, j) * w->sx[j];
      Hs[(size_t)j * K + (n + a)] = v;
      Hs[(size_t)(n + a) * K + j] = v;
      Hp[(size_t)(n + a) * K + j] = v;
    }
  }
  for (int i = 0; i < n; ++i) Hp[(size_t)i * K + i] += delta;
  for (int a = 0; a < na; ++a) Hp[(size_t)(n + a) * K + (n + a)] -= delta;

  /* :179-182 */
  for (int j = 0; j < n; ++j) h[j] = -w->c * (w->sx[j] * w->q[j]);
  for (int a = 0; a < nl; ++a) h[n + a] = w->sy[LU[a]] * w->l[LU[a]];
  for (int a = 0; a < nu; ++a) h[n + nl + a] = w->sy[LU[nl + a]] * w->u[LU[nl + a]];

  int ok = oracle_ldlt_factor(K, Hp, K, tr); /* :187-190 */
  if (ok) {
    for (uint32_t it = 0; it != w->prm.polish_iter; ++it) { /* :193-195 */
      for (int i = 0; i < K; ++i) {                          /* r = h - Hsym * t */
        double s = 0.0;
        for (int j = 0; j < K; ++j) s = fma(Hs[(size_t)i * K + j], t[j], s);
        r[i] = h[i] - s;
      }
      oracle_ldlt_solve(K, Hp, K, tr, r);
      for (int i = 0; i < K; ++i) t[i] += r[i];
    }
    for (int j = 0; j < n; ++j) w->primal[j] = t[j];                       /* :199 */
    for (int a = 0; a < nl; ++a) w->dual[LU[a]] = t[n + a];                 /* :200 */
    for (int a = 0; a < nu; ++a) w->dual[LU[nl + a]] = t[n + nl + a];       /* :201 */
  }
  free(LU); free(Hs); free(Hp); free(h); free(t); free(r); free(tr);
  return ok;
}

int oracle_qp_dense_solve(const oracle_qp_params *prm, int n, int m, const double *P, const double *q,
                          const double *A, const double *l, const double *u, const double *warm_x,
                          const double *warm_y, double *x, double *y, double *obj, uint32_t *iter_out,
                          int32_t *code_out)
{
  if (!prm || n < 1 || m < 1 || !P || !q || !A || !l || !u || !x || !y) return -1;
  if ((warm_x == NULL) != (warm_y == NULL)) return -1;

  qp_work W;
  qp_work *w = &W;
  memset(w, 0, sizeof(*w));
  const int k = n + m;
  w->n = n; w->m = m; w->k = k;
  w->P = P; w->q = q; w->A = A; w->l = l; w->u = u;
  w->prm = *prm;

  /* analyze(): qp_solver.hpp:297-338 -- one allocation for all work vectors */
  const size_t nd = (size_t)(6 * n + 10 * m + k) + (size_t)k * (size_t)k;
  double *mem     = (double *)calloc(nd, sizeof(double));
  int *tr         = (int *)malloc(sizeof(int) * (size_t)k);
  if (!mem || !tr) { free(mem); free(tr); return -1; }
  double *ptr = mem;
  w->sx = ptr; ptr += n;  w->sx_inc = ptr; ptr += n;
  w->x_us = ptr; ptr += n; w->dx_us = ptr; ptr += n; w->Px = ptr; ptr += n; w->Aty = ptr; ptr += n;
  w->sy = ptr; ptr += m;  w->sy_inc = ptr; ptr += m;
  w->z = ptr; ptr += m;   w->z_next = ptr; ptr += m; w->rho = ptr; ptr += m;
  w->Ax = ptr; ptr += m;  w->y_us = ptr; ptr += m;  w->z_us = ptr; ptr += m; w->dy_us = ptr; ptr += m;
  ptr += m; /* spare */
  w->p = ptr; ptr += k;
  w->H = ptr;
  w->tr = tr;
  w->primal = x;
  w->dual   = y;
  w->c = 1.0;                                       /* :306 */
  for (int j = 0; j < n; ++j) w->sx[j] = 1.0;       /* :307 */
  for (int i = 0; i < m; ++i) w->sy[i] = 1.0;       /* :308 */

  const double inf = INFINITY;

  if (prm->scaling) qp_scale(w); /* :347 */

  /* :353-356 float parameters widened to double */
  const double rho_bar    = (double)prm->rho;
  const double alpha      = (double)prm->alpha;
  const double alpha_comp = 1.0 - alpha;
  const double sigma      = (double)prm->sigma;

  int ret_code = -1; /* std::nullopt */

  for (int i = 0; i < m; ++i) { /* :361-374 */
    if (l[i] == inf || u[i] == -inf || u[i] - l[i] < 0.0) ret_code = ORACLE_QP_PRIMAL_INFEASIBLE;
    if (l[i] == -inf && u[i] == inf) {
      w->rho[i] = 1e-6;
    } else if (w->sy[i] * fabs(l[i] - u[i]) < 1e-5) {
      w->rho[i] = 1e3 * rho_bar;
    } else {
      w->rho[i] = rho_bar;
    }
  }

  struct timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0); /* :376 */

  /* :399-404 dense KKT fill; only the upper triangle is read by LDLT<Upper>.  Stored here as the
   * lower triangle of the row-major k x k array H: H[r][c] (r>=c) == Eigen's H_(c, r). */
  double *H = w->H;
  for (int r = 0; r < n; ++r)
    for (int cidx = 0; cidx <= r; ++cidx) { /* upper entry (cidx, r): ((c*sx_c)*P_cr)*sx_r */
      double v = w->c * w->sx[cidx] * PM(cidx, r) * w->sx[r];
      if (cidx == r) v += sigma; /* :402 */
      H[(size_t)r * k + cidx] = v;
    }
  for (int i = 0; i < m; ++i) {
    for (int j = 0; j < n; ++j) H[(size_t)(n + i) * k + j] = w->sy[i] * AM(i, j) * w->sx[j]; /* :403 */
    for (int i2 = 0; i2 < i; ++i2) H[(size_t)(n + i) * k + (n + i2)] = 0.0;
    H[(size_t)(n + i) * k + (n + i)] = 1.0 / (-w->rho[i]); /* :404 (-rho).cwiseInverse() */
  }

  if (!oracle_ldlt_factor(k, H, k, tr)) ret_code = ORACLE_QP_UNKNOWN; /* :428-433 */

  /* :436-445 */
  if (warm_x) {
    for (int j = 0; j < n; ++j) w->primal[j] = (1.0 / w->sx[j]) * warm_x[j];
    for (int i = 0; i < m; ++i) w->dual[i] = w->c * ((1.0 / w->sy[i]) * warm_y[i]);
    for (int i = 0; i < m; ++i) { /* z = (Sy*A) * x_ws : s = fma((sy_i*A_ij), x_j, s), j ascending */
      double s = 0.0;
      for (int j = 0; j < n; ++j) s = fma(w->sy[i] * AM(i, j), warm_x[j], s);
      w->z[i] = s;
    }
  } else {
    for (int j = 0; j < n; ++j) w->primal[j] = 0.0;
    for (int i = 0; i < m; ++i) w->dual[i] = 0.0;
    for (int i = 0; i < m; ++i) w->z[i] = 0.0;
  }

  /* :447-510 main loop */
  uint32_t iter = 0;
  const uint32_t sci = prm->stop_check_iter;
  for (; (prm->max_iter < 0 || (int64_t)iter != prm->max_iter) && ret_code < 0; ++iter) {
    for (int j = 0; j < n; ++j) w->p[j] = sigma * w->primal[j] - w->c * w->sx[j] * q[j]; /* :450 */
    for (int i = 0; i < m; ++i) w->p[n + i] = w->z[i] - (1.0 / w->rho[i]) * w->dual[i];  /* :451 */
    oracle_ldlt_solve(k, H, k, tr, w->p);                                               /* :462 */

    const int chk = (sci != 0) && (iter % sci == 1); /* :465 (sci==0 would be UB in the reference) */
    if (chk) {
      memcpy(w->dx_us, w->primal, sizeof(double) * (size_t)n);
      memcpy(w->dy_us, w->dual, sizeof(double) * (size_t)m);
    }

    for (int j = 0; j < n; ++j) w->primal[j] = alpha * w->p[j] + alpha_comp * w->primal[j]; /* :470 */
    for (int i = 0; i < m; ++i) { /* :471-476 */
      const double rinv = 1.0 / w->rho[i];
      const double nu   = w->p[n + i];
      double zn         = alpha * (rinv * nu) + alpha_comp * (rinv * w->dual[i]) + w->z[i];
      zn                = dmax(zn, w->sy[i] * l[i]); /* cwiseMax */
      zn                = dmin(zn, w->sy[i] * u[i]); /* cwiseMin */
      w->z_next[i]      = zn;
      w->dual[i]        = alpha_comp * w->dual[i] + alpha * nu + w->rho[i] * w->z[i] - w->rho[i] * zn;
    }
    { double *tmp = w->z; w->z = w->z_next; w->z_next = tmp; } /* :477 */

    if (chk) { /* :479-509 */
      for (int j = 0; j < n; ++j) w->x_us[j] = w->sx[j] * w->primal[j];
      for (int i = 0; i < m; ++i) w->y_us[i] = w->sy[i] * w->dual[i] / w->c;
      for (int i = 0; i < m; ++i) w->z_us[i] = (1.0 / w->sy[i]) * w->z[i];
      for (int j = 0; j < n; ++j) w->dx_us[j] = w->sx[j] * (w->primal[j] - w->dx_us[j]);
      for (int i = 0; i < m; ++i) w->dy_us[i] = w->sy[i] * (w->dual[i] - w->dy_us[i]) / w->c;
      ret_code = qp_check_stopping(w);
      if (tl_dense_trace && tl_dense_trace_rows < g_dense_trace_cap) { /* :490-501, the three columns in the reference's expressions */
        double *row = tl_dense_trace + 6 * (size_t)tl_dense_trace_rows++;
        double o = 0.0, pri = 0.0, dua = 0.0;
        mv_P(w, w->x_us, w->Px);
        for (int j = 0; j < n; ++j) o += (0.5 * w->Px[j] + q[j]) * w->x_us[j];
        mv_A(w, w->x_us, w->Ax);
        for (int i = 0; i < m; ++i) pri = dmax(pri, fabs(w->Ax[i] - w->z_us[i]));
        mv_At(w, w->y_us, w->Aty);
        for (int j = 0; j < n; ++j) dua = dmax(dua, fabs(w->Px[j] + q[j] + w->Aty[j]));
        row[0] = (double)iter; row[1] = o; row[2] = pri; row[3] = dua;
        row[4] = (double)prm->eps_abs + (double)prm->eps_rel * dmax(norm_inf(w->Ax, m), norm_inf(w->z_us, m));
        row[5] = (double)prm->eps_abs + (double)prm->eps_rel * dmax(dmax(norm_inf(w->Px, n), norm_inf(q, n)), norm_inf(w->Aty, n));
      }
      if (ret_code < 0 && prm->max_time_ns >= 0) { /* :504-507 */
        struct timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        const int64_t el = (int64_t)(t1.tv_sec - t0.tv_sec) * 1000000000LL + (t1.tv_nsec - t0.tv_nsec);
        if (el > prm->max_time_ns) ret_code = ORACLE_QP_MAX_TIME;
      }
    }
  }

  /* :515-539 polish; a failed polish leaves code == Optimal (it is overwritten at :544) */
  if (ret_code == ORACLE_QP_OPTIMAL && prm->polish) (void)qp_polish(w);

  /* :544-548 */
  *code_out = (ret_code >= 0) ? ret_code : ORACLE_QP_MAX_ITERATIONS;
  for (int j = 0; j < n; ++j) w->primal[j] = w->sx[j] * w->primal[j];
  for (int i = 0; i < m; ++i) w->dual[i] = w->sy[i] * w->dual[i] / w->c;
  { /* objective = primal.dot(0.5*P*primal + q): t_i = (sum_j fma(0.5*P_ij, x_j)) + q_i ; then
     * ascending fma chain for the outer dot product */
    double o = 0.0;
    for (int i = 0; i < n; ++i) {
      double s = 0.0;
      for (int j = 0; j < n; ++j) s = fma(0.5 * PM(i, j), w->primal[j], s);
      o = fma(w->primal[i], s + q[i], o);
    }
    if (obj) *obj = o;
  }
  if (iter_out) *iter_out = iter;

  free(mem);
  free(tr);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */

typedef struct {
  const oracle_qp_params *prm;
  int64_t batch, *next; /* items are handed out in chunks (iteration counts are heavy-tailed) */
  int n, m;
  const double *P, *q, *A, *l, *u, *wx, *wy;
  double *x, *y, *obj;
  uint32_t *iter;
  int32_t *code;
  int rc;
} batch_job;

static void *batch_worker(void *arg)
{
  batch_job *j = (batch_job *)arg;
  const size_t n = (size_t)j->n, m = (size_t)j->m;
  for (int64_t b0; (b0 = __atomic_fetch_add(j->next, 16, __ATOMIC_RELAXED)) < j->batch;)
  for (int64_t b = b0; b < b0 + 16 && b < j->batch; ++b) {
    const size_t sb = (size_t)b;
    tl_dense_trace      = g_dense_trace ? g_dense_trace + 6 * sb * (size_t)g_dense_trace_cap : NULL;
    tl_dense_trace_rows = 0;
    for (int r = 0; tl_dense_trace && r < g_dense_trace_cap; ++r) tl_dense_trace[6 * r] = -1.0;
    int rc = oracle_qp_dense_solve(j->prm, j->n, j->m, j->P + sb * n * n, j->q + sb * n, j->A + sb * m * n,
                                   j->l + sb * m, j->u + sb * m, j->wx ? j->wx + sb * n : NULL,
                                   j->wy ? j->wy + sb * m : NULL, j->x + sb * n, j->y + sb * m,
                                   j->obj ? j->obj + sb : NULL, j->iter ? j->iter + sb : NULL, j->code + sb);
    if (rc) j->rc = rc;
  }
  return NULL;
}

int oracle_qp_dense_solve_batch(const oracle_qp_params *prm, int64_t batch, int n, int m, const double *P,
                                const double *q, const double *A, const double *l, const double *u,
                                const double *warm_x, const double *warm_y, double *x, double *y,
                                double *obj, uint32_t *iter, int32_t *code, int nthreads)
{
  if (batch < 0 || !code) return -1;
  if (nthreads < 1) nthreads = 1;
  if ((int64_t)nthreads > batch) nthreads = (int)(batch > 0 ? batch : 1);
  batch_job *jobs = (batch_job *)calloc((size_t)nthreads, sizeof(batch_job));
  pthread_t *th   = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
  if (!jobs || !th) { free(jobs); free(th); return -1; }
  int64_t next_item = 0;
  for (int t = 0; t < nthreads; ++t) {
    batch_job *j = &jobs[t];
    j->prm = prm; j->n = n; j->m = m;
    j->batch = batch; j->next = &next_item;
    j->P = P; j->q = q; j->A = A; j->l = l; j->u = u; j->wx = warm_x; j->wy = warm_y;
    j->x = x; j->y = y; j->obj = obj; j->iter = iter; j->code = code;
  }
  if (nthreads == 1) {
    batch_worker(&jobs[0]);
  } else {
    for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
  }
  int rc = 0;
  for (int t = 0; t < nthreads; ++t)
    if (jobs[t].rc) rc = jobs[t].rc;
  free(jobs);
  free(th);
  return rc;
}
