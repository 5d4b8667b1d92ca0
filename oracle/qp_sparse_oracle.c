/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * CPU restatement of the SPARSE branch of smooth::feedback::QPSolver (pettni/smooth_feedback @ v1):
 *   include/smooth/feedback/qp_solver.hpp  sparse branches :132-158,:169-173 (polish), :332-337
 *   (analyze), :379-397 (KKT fill), :423-426 (factorise), :452-460 (manual solve), scale :673-730,
 *   check_stopping :574-644;  qp.hpp:60-79 (QuadraticProgramSparse: P CSC, A CSR).
 * Third party on the path, absent from /root/reference: Eigen 3.4.0 SimplicialLDLT<.,Upper>
 * (AMD ordering + up-looking LDL' without numerical pivoting).
 *
 * PARITY STATUS.  Pinned: the sparse known answers of tests/test_qp.cpp (BasicSparse :103-122,
 * PortfolioOptimizationSparse :277-312, TwoDimensional dense==sparse :314-336) and agreement with
 * the dense oracle on the same problems (tests/test_oracle_qp_sparse.py).  UNPINNED: Eigen's AMD
 * ordering cannot be reproduced without Eigen, so the elimination order is an INPUT here (`perm`,
 * any fill-reducing permutation of the KKT matrix); the numeric factorisation computes the same
 * L, D as SimplicialLDLT for that order but accumulates each entry over its source columns in the
 * order of a POSTORDER of the elimination tree (children in ascending order; left-looking) instead
 * of Eigen's etree-reach order -- both are topological orders of the same dependencies, Eigen's
 * depends on the storage order of each row of the permuted matrix; the backward sweep pushes
 * row by row (descending) instead of Eigen's per-row dot product.  The polish step factorises the
 * reduced KKT system EMBEDDED in the full pattern (inactive rows zeroed, diagonal -delta), which is
 * algebraically the reference's reduced system with a different (fixed) elimination order.
 * The HIP kernel follows exactly these choices, so oracle and kernel agree bit for bit.
 */
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "qp_oracle.h"

static inline double dmax(double a, double b) { return (a < b) ? b : a; }
static inline double dmin(double a, double b) { return (b < a) ? b : a; }

/* ---------------- symbolic structure of the permuted KKT matrix ---------------- */
typedef struct {
  int n, m, k, nnzK, nnzL;
  int *perm, *pinv;       /* perm[new] = old, pinv[old] = new                                   */
  int *Kp, *Ki, *Kkind, *Kidx; /* lower CSC of the permuted KKT: column j, rows i >= j           */
  int *Lp, *Li;           /* strictly lower pattern of L, column-major, rows ascending           */
  int *Rp, *Rk, *Rpos;    /* row structure of L: row j -> (source column kk < j ascending, pos)   */
  int *RFk, *RFpos;       /* the same lists sorted by the POSTORDER rank of kk: the order in which
                             the numeric factorisation accumulates the sources of row j            */
} ksym;

enum { K_P = 0, K_A = 1, K_SIGMA = 2, K_RHO = 3 };

static void ksym_free(ksym *s)
{
  free(s->perm); free(s->pinv); free(s->Kp); free(s->Ki); free(s->Kkind); free(s->Kidx);
  free(s->Lp); free(s->Li); free(s->Rp); free(s->Rk); free(s->Rpos); free(s->RFk); free(s->RFpos);
}

typedef struct { int i, j, kind, idx; } kent;
static int kent_cmp(const void *a, const void *b)
{
  const kent *x = (const kent *)a, *y = (const kent *)b;
  if (x->j != y->j) return x->j - y->j;
  return x->i - y->i;
}

static int ksym_build(ksym *s, int n, int m, const int32_t *Pp, const int32_t *Pi, const int32_t *Ap,
                      const int32_t *Aj, const int32_t *perm_in, const int32_t *forder)
{
  memset(s, 0, sizeof(*s));
  const int k = n + m;
  s->n = n; s->m = m; s->k = k;
  s->perm = (int *)malloc(sizeof(int) * (size_t)k);
  s->pinv = (int *)malloc(sizeof(int) * (size_t)k);
  for (int i = 0; i < k; ++i) s->perm[i] = perm_in ? perm_in[i] : i;
  for (int i = 0; i < k; ++i) s->pinv[s->perm[i]] = i;

  /* entries of the KKT upper triangle (qp_solver.hpp:382-395), mapped to the permuted lower form */
  const int nnzP = Pp[n], nnzA = Ap[m];
  kent *e = (kent *)malloc(sizeof(kent) * (size_t)(nnzP + nnzA + k));
  char *hasdiag = (char *)calloc((size_t)n, 1);
  int ne = 0;
  for (int c = 0; c < n; ++c)
    for (int p = Pp[c]; p < Pp[c + 1]; ++p) {
      const int r = Pi[p];
      if (c >= r) { /* :384 */
        const int a = s->pinv[r], b = s->pinv[c];
        e[ne].i = a > b ? a : b; e[ne].j = a > b ? b : a; e[ne].kind = K_P; e[ne].idx = p;
        ++ne;
        if (r == c) hasdiag[r] = 1;
      }
    }
  for (int v = 0; v < n; ++v)
    if (!hasdiag[v]) { e[ne].i = e[ne].j = s->pinv[v]; e[ne].kind = K_SIGMA; e[ne].idx = v; ++ne; } /* :389 */
  for (int r = 0; r < m; ++r) {
    for (int p = Ap[r]; p < Ap[r + 1]; ++p) { /* :392  H(col, n+row) */
      const int a = s->pinv[Aj[p]], b = s->pinv[n + r];
      e[ne].i = a > b ? a : b; e[ne].j = a > b ? b : a; e[ne].kind = K_A; e[ne].idx = p;
      ++ne;
    }
    e[ne].i = e[ne].j = s->pinv[n + r]; e[ne].kind = K_RHO; e[ne].idx = r; ++ne; /* :395 */
  }
  free(hasdiag);
  qsort(e, (size_t)ne, sizeof(kent), kent_cmp);
  s->nnzK  = ne;
  s->Kp    = (int *)calloc((size_t)k + 1, sizeof(int));
  s->Ki    = (int *)malloc(sizeof(int) * (size_t)ne);
  s->Kkind = (int *)malloc(sizeof(int) * (size_t)ne);
  s->Kidx  = (int *)malloc(sizeof(int) * (size_t)ne);
  for (int t = 0; t < ne; ++t) {
    s->Kp[e[t].j + 1]++;
    s->Ki[t] = e[t].i; s->Kkind[t] = e[t].kind; s->Kidx[t] = e[t].idx;
  }
  for (int j = 0; j < k; ++j) s->Kp[j + 1] += s->Kp[j];

  /* rows of the lower form: row r -> columns j < r  (== column r of the upper form) */
  int *rp = (int *)calloc((size_t)k + 1, sizeof(int));
  for (int t = 0; t < ne; ++t)
    if (e[t].i != e[t].j) rp[e[t].i + 1]++;
  for (int r = 0; r < k; ++r) rp[r + 1] += rp[r];
  int *rj = (int *)malloc(sizeof(int) * (size_t)(rp[k] > 0 ? rp[k] : 1));
  int *fill = (int *)calloc((size_t)k, sizeof(int));
  for (int t = 0; t < ne; ++t)
    if (e[t].i != e[t].j) rj[rp[e[t].i] + fill[e[t].i]++] = e[t].j;
  free(e);

  /* elimination tree and pattern of L (up-looking reach, T. Davis' LDL symbolic) */
  int *parent = (int *)malloc(sizeof(int) * (size_t)k);
  int *flag   = (int *)malloc(sizeof(int) * (size_t)k);
  int *lnz    = (int *)calloc((size_t)k, sizeof(int));
  for (int r = 0; r < k; ++r) {
    parent[r] = -1;
    flag[r]   = r;
    for (int p = rp[r]; p < rp[r + 1]; ++p)
      for (int i = rj[p]; flag[i] != r; i = parent[i]) {
        if (parent[i] == -1) parent[i] = r;
        lnz[i]++;
        flag[i] = r;
      }
  }
  s->Lp = (int *)calloc((size_t)k + 1, sizeof(int));
  for (int j = 0; j < k; ++j) s->Lp[j + 1] = s->Lp[j] + lnz[j];
  s->nnzL = s->Lp[k];
  s->Li   = (int *)malloc(sizeof(int) * (size_t)(s->nnzL > 0 ? s->nnzL : 1));
  memset(fill, 0, sizeof(int) * (size_t)k);
  for (int r = 0; r < k; ++r) { /* second pass: row r is appended to every column on its reach */
    flag[r] = r;
    for (int p = rp[r]; p < rp[r + 1]; ++p)
      for (int i = rj[p]; flag[i] != r; i = parent[i]) {
        s->Li[s->Lp[i] + fill[i]++] = r;
        flag[i] = r;
      }
  }
  /* row structure */
  s->Rp = (int *)calloc((size_t)k + 1, sizeof(int));
  for (int p = 0; p < s->nnzL; ++p) s->Rp[s->Li[p] + 1]++;
  for (int r = 0; r < k; ++r) s->Rp[r + 1] += s->Rp[r];
  s->Rk   = (int *)malloc(sizeof(int) * (size_t)(s->nnzL > 0 ? s->nnzL : 1));
  s->Rpos = (int *)malloc(sizeof(int) * (size_t)(s->nnzL > 0 ? s->nnzL : 1));
  memset(fill, 0, sizeof(int) * (size_t)k);
  for (int j = 0; j < k; ++j)
    for (int p = s->Lp[j]; p < s->Lp[j + 1]; ++p) {
      const int r = s->Li[p];
      s->Rk[s->Rp[r] + fill[r]]   = j;
      s->Rpos[s->Rp[r] + fill[r]] = p;
      fill[r]++;
    }
  /* postorder of the elimination tree: depth first from every root in ascending order, children in
   * ascending order; rank[j] = position of column j in it */
  {
    int *head = (int *)malloc(sizeof(int) * (size_t)k), *next = (int *)malloc(sizeof(int) * (size_t)k);
    int *stack = (int *)malloc(sizeof(int) * (size_t)k), *rank = (int *)malloc(sizeof(int) * (size_t)k);
    for (int j = 0; j < k; ++j) head[j] = next[j] = -1;
    for (int j = k - 1; j >= 0; --j)
      if (parent[j] >= 0) { next[j] = head[parent[j]]; head[parent[j]] = j; }
    int cnt = 0;
    for (int r = 0; r < k; ++r) {
      if (parent[r] >= 0) continue;
      int sp = 0;
      stack[sp++] = r;
      while (sp > 0) {
        const int v = stack[sp - 1], c = head[v];
        if (c >= 0) { head[v] = next[c]; stack[sp++] = c; }
        else { rank[v] = cnt++; --sp; }
      }
    }
    /* an explicit accumulation order (rank of every column of the permuted matrix) overrides the postorder:
     * ANY order of a row's sources is a valid summation order for the left-looking loop below */
    if (forder)
      for (int j = 0; j < k; ++j) rank[j] = forder[j];
    s->RFk   = (int *)malloc(sizeof(int) * (size_t)(s->nnzL > 0 ? s->nnzL : 1));
    s->RFpos = (int *)malloc(sizeof(int) * (size_t)(s->nnzL > 0 ? s->nnzL : 1));
    for (int j = 0; j < k; ++j) { /* insertion sort of each row's sources by rank (rows are short) */
      for (int t = s->Rp[j]; t < s->Rp[j + 1]; ++t) {
        const int kk = s->Rk[t], pos = s->Rpos[t];
        int q = t;
        while (q > s->Rp[j] && rank[s->RFk[q - 1]] > rank[kk]) { s->RFk[q] = s->RFk[q - 1]; s->RFpos[q] = s->RFpos[q - 1]; --q; }
        s->RFk[q] = kk; s->RFpos[q] = pos;
      }
    }
    free(head); free(next); free(stack); free(rank);
  }
  free(rp); free(rj); free(fill); free(parent); free(flag); free(lnz);
  return 0;
}

/* Numeric left-looking LDL' on the fixed pattern.  Kval: values aligned with Kp/Ki.
 * Returns 1 on success, 0 on a zero pivot (SimplicialLDLT info() == NumericalIssue). */
static int ldl_numeric(const ksym *s, const double *Kval, double *Lx, double *D, double *work)
{
  const int k = s->k;
  for (int j = 0; j < k; ++j) {
    work[j] = 0.0;
    for (int p = s->Lp[j]; p < s->Lp[j + 1]; ++p) work[s->Li[p]] = 0.0;
    for (int p = s->Kp[j]; p < s->Kp[j + 1]; ++p) work[s->Ki[p]] = Kval[p];
    for (int t = s->Rp[j]; t < s->Rp[j + 1]; ++t) { /* source columns kk < j, in postorder of the elimination tree */
      const int kk = s->RFk[t], pos = s->RFpos[t];
      const double w = Lx[pos] * D[kk]; /* L(j,kk) * D(kk) */
      for (int p = pos; p < s->Lp[kk + 1]; ++p) work[s->Li[p]] = fma(-Lx[p], w, work[s->Li[p]]);
    }
    const double d = work[j];
    D[j]           = d;
    if (d == 0.0) return 0;
    for (int p = s->Lp[j]; p < s->Lp[j + 1]; ++p) Lx[p] = work[s->Li[p]] / d;
  }
  return 1;
}

/* qp_solver.hpp:456-460: pt = P p; L solve; D^-1 (reciprocal then multiply); L' solve; p = P^-1 pt.
 * b in original order (length k), overwritten with the solution. t: work (k). */
static void ldl_solve(const ksym *s, const double *Lx, const double *Dinv, double *b, double *t)
{
  const int k = s->k;
  for (int i = 0; i < k; ++i) t[i] = b[s->perm[i]];
  for (int j = 0; j < k; ++j) { /* forward, column oriented */
    const double tj = t[j];
    for (int p = s->Lp[j]; p < s->Lp[j + 1]; ++p) t[s->Li[p]] = fma(-Lx[p], tj, t[s->Li[p]]);
  }
  for (int j = 0; j < k; ++j) t[j] = Dinv[j] * t[j];
  for (int j = k - 1; j >= 0; --j) { /* backward: row j pushes into its columns */
    const double tj = t[j];
    for (int r = s->Rp[j]; r < s->Rp[j + 1]; ++r) t[s->Rk[r]] = fma(-Lx[s->Rpos[r]], tj, t[s->Rk[r]]);
  }
  for (int i = 0; i < k; ++i) b[s->perm[i]] = t[i];
}

/* ---------------- one sparse solve ---------------- */
typedef struct {
  const ksym *s;
  int n, m, k;
  const int32_t *Pp, *Pi, *Ap, *Aj;
  int *Acp, *Aci, *Acpos; /* CSC view of A: column j -> (row ascending, position in Ax) */
  int *Prp, *Prj, *Prpos; /* CSR view of P: row i -> (col ascending, position in Px) */
  int *Sp, *Sj, *Spos;    /* symmetric view of triu(P): row i -> (col ascending, position of the upper entry) */
} sp_shared;

static void build_transposed(int nrows_out, int nnz, const int32_t *outer_ptr, int nouter, const int32_t *inner,
                             int **tp, int **ti, int **tpos)
{
  /* input: compressed by `outer` (nouter), inner indices in [0, nrows_out); output compressed by inner */
  *tp   = (int *)calloc((size_t)nrows_out + 1, sizeof(int));
  *ti   = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
  *tpos = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
  for (int p = 0; p < nnz; ++p) (*tp)[inner[p] + 1]++;
  for (int i = 0; i < nrows_out; ++i) (*tp)[i + 1] += (*tp)[i];
  int *fill = (int *)calloc((size_t)nrows_out, sizeof(int));
  for (int o = 0; o < nouter; ++o)
    for (int p = outer_ptr[o]; p < outer_ptr[o + 1]; ++p) {
      const int i = inner[p];
      (*ti)[(*tp)[i] + fill[i]]   = o;
      (*tpos)[(*tp)[i] + fill[i]] = p;
      fill[i]++;
    }
  free(fill);
}

typedef struct {
  const sp_shared *sh;
  oracle_qp_params prm;
  const double *Px, *q, *Ax, *l, *u;
  double c, *sx, *sy, *sx_inc, *sy_inc, *rho, *z, *z_next, *p, *t;
  double *x_us, *dx_us, *Pxv, *Aty, *Axv, *y_us, *z_us, *dy_us;
  double *Kval, *Lx, *D, *Dinv, *work;
  double *primal, *dual;
  double *trace; /* verbose table (qp_solver.hpp:490-501), one row (iter, obj, pri_res, dua_res) per stopping check */
  int trace_cap, trace_rows;
} sp_work;

/* The rows of the reference's verbose table (qp_solver.hpp:409-420 header, :490-501 rows) instead of printing them:
 * trace[batch][cap][6] = (ITER, OBJ, PRI_RES, DUA_RES, tolerance of PRI_RES, tolerance of DUA_RES) of every stopping
 * check, rows beyond the last check hold ITER = -1.  Process-global switch (test infrastructure; set before a batch call, cleared with NULL). */
static double *g_trace   = NULL;
static int g_trace_cap   = 0;
void oracle_qp_sparse_set_trace(double *trace, int cap) { g_trace = trace; g_trace_cap = cap; }

static double norm_inf(const double *v, int len)
{
  double r = 0.0;
  for (int i = 0; i < len; ++i) r = dmax(r, fabs(v[i]));
  return r;
}
/* A v: row-major sparse times dense, per row in storage order */
static void sp_mv_A(const sp_work *w, const double *v, double *out)
{
  const sp_shared *sh = w->sh;
  for (int i = 0; i < sh->m; ++i) {
    double s = 0.0;
    for (int p = sh->Ap[i]; p < sh->Ap[i + 1]; ++p) s = fma(w->Ax[p], v[sh->Aj[p]], s);
    out[i] = s;
  }
}
/* A' v: res[j] accumulates over rows i ascending */
static void sp_mv_At(const sp_work *w, const double *v, double *out)
{
  const sp_shared *sh = w->sh;
  for (int j = 0; j < sh->n; ++j) {
    double s = 0.0;
    for (int p = sh->Acp[j]; p < sh->Acp[j + 1]; ++p) s = fma(w->Ax[sh->Acpos[p]], v[sh->Aci[p]], s);
    out[j] = s;
  }
}
/* P v with P as stored (CSC): res[i] accumulates over columns j ascending */
static void sp_mv_P(const sp_work *w, const double *v, double *out)
{
  const sp_shared *sh = w->sh;
  for (int i = 0; i < sh->n; ++i) {
    double s = 0.0;
    for (int p = sh->Prp[i]; p < sh->Prp[i + 1]; ++p) s = fma(w->Px[sh->Prpos[p]], v[sh->Prj[p]], s);
    out[i] = s;
  }
}

static void sp_scale(sp_work *w) /* qp_solver.hpp:673-730 */
{
  const sp_shared *sh = w->sh;
  const int n = sh->n, m = sh->m;
  for (int j = 0; j < n; ++j) { w->sx[j] = 1.0; w->sx_inc[j] = 0.0; }
  for (int i = 0; i < m; ++i) w->sy[i] = 1.0;
  for (int cidx = 0; cidx < n; ++cidx)
    for (int p = sh->Pp[cidx]; p < sh->Pp[cidx + 1]; ++p) w->sx_inc[cidx] = dmax(w->sx_inc[cidx], fabs(w->Px[p]));
  for (int j = 0; j < n; ++j)
    if (w->sx_inc[j] == 0.0) w->sx_inc[j] = 1.0;
  double sum = w->sx_inc[0];
  for (int j = 1; j < n; ++j) sum += w->sx_inc[j];
  w->c = 1.0 / dmax(dmax(1e-6, sum / (double)n), norm_inf(w->q, n));
  int iter = 0;
  double crit;
  do {
    for (int j = 0; j < n; ++j) w->sx_inc[j] = 0.0;
    for (int i = 0; i < m; ++i) w->sy_inc[i] = 0.0;
    for (int cidx = 0; cidx < n; ++cidx)
      for (int p = sh->Pp[cidx]; p < sh->Pp[cidx + 1]; ++p)
        w->sx_inc[cidx] = dmax(w->sx_inc[cidx], fabs(w->c * w->sx[sh->Pi[p]] * w->sx[cidx] * w->Px[p]));
    for (int r = 0; r < m; ++r)
      for (int p = sh->Ap[r]; p < sh->Ap[r + 1]; ++p) {
        const int cidx   = sh->Aj[p];
        const double Aij = fabs(w->sy[r] * w->sx[cidx] * w->Ax[p]);
        w->sx_inc[cidx]  = dmax(w->sx_inc[cidx], Aij);
        w->sy_inc[r]     = dmax(w->sy_inc[r], Aij);
      }
    for (int j = 0; j < n; ++j)
      if (w->sx_inc[j] == 0.0) w->sx_inc[j] = 1.0;
    for (int i = 0; i < m; ++i)
      if (w->sy_inc[i] == 0.0) w->sy_inc[i] = 1.0;
    for (int j = 0; j < n; ++j) w->sx[j] = sqrt(1.0 / dmax(w->sx_inc[j], 1e-8)) * w->sx[j];
    for (int i = 0; i < m; ++i) w->sy[i] = sqrt(1.0 / dmax(w->sy_inc[i], 1e-8)) * w->sy[i];
    double a = 0.0, b = 0.0;
    for (int j = 0; j < n; ++j) a = dmax(a, fabs(w->sx_inc[j] - 1.0));
    for (int i = 0; i < m; ++i) b = dmax(b, fabs(w->sy_inc[i] - 1.0));
    crit = dmax(a, b);
  } while (iter++ < 10 && crit > 0.1);
}

static int sp_check_stopping(sp_work *w) /* qp_solver.hpp:574-644, identical logic to the dense oracle */
{
  const sp_shared *sh = w->sh;
  const int n = sh->n, m = sh->m;
  const double inf          = INFINITY;
  const oracle_qp_params *p = &w->prm;
  sp_mv_A(w, w->x_us, w->Axv);
  const double Ax_norm = norm_inf(w->Axv, m);
  for (int i = 0; i < m; ++i) w->Axv[i] -= w->z_us[i];
  if (norm_inf(w->Axv, m) <= (double)p->eps_abs + (double)p->eps_rel * dmax(Ax_norm, norm_inf(w->z_us, m))) {
    sp_mv_P(w, w->x_us, w->Pxv);
    sp_mv_At(w, w->y_us, w->Aty);
    const double dual_scale = dmax(dmax(norm_inf(w->Pxv, n), norm_inf(w->q, n)), norm_inf(w->Aty, n));
    for (int j = 0; j < n; ++j) w->Pxv[j] += w->q[j] + w->Aty[j];
    if (norm_inf(w->Pxv, n) <= (double)p->eps_abs + (double)p->eps_rel * dual_scale) return ORACLE_QP_OPTIMAL;
  }
  sp_mv_At(w, w->dy_us, w->Aty);
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
  sp_mv_A(w, w->dx_us, w->Axv);
  const double dx_norm = norm_inf(w->dx_us, n);
  sp_mv_P(w, w->dx_us, w->Pxv);
  double qdx = 0.0;
  for (int j = 0; j < n; ++j) qdx = fma(w->q[j], w->dx_us[j], qdx);
  const double thr = (double)p->eps_dual_inf * dx_norm;
  int dual_inf     = (norm_inf(w->Pxv, n) <= thr) && (qdx <= thr);
  for (int i = 0; i < m && dual_inf; ++i) {
    if (w->u[i] == inf) {
      dual_inf &= (w->Axv[i] >= (double)(-p->eps_dual_inf) * dx_norm);
    } else if (w->l[i] == -inf) {
      dual_inf &= (w->Axv[i] <= thr);
    } else {
      dual_inf &= (fabs(w->Axv[i]) < thr);
    }
  }
  if (dual_inf) return ORACLE_QP_DUAL_INFEASIBLE;
  return -1;
}

/* KKT values on the permuted lower pattern. mode 0: ADMM matrix (:382-395); mode 1: polish matrix
 * H + diag(delta, -delta) embedded in the full pattern (act[r] != 0 marks active rows). */
typedef struct { int *Pcol; int *Arow; } sp_aux;

static void sp_fill_K2(const sp_work *w, const sp_aux *aux, int mode, const char *act, double *Kval)
{
  const ksym *s       = w->sh->s;
  const sp_shared *sh = w->sh;
  const double sigma = (double)w->prm.sigma, delta = (double)w->prm.delta;
  for (int p = 0; p < s->nnzK; ++p) {
    const int kind = s->Kkind[p], idx = s->Kidx[p];
    double v;
    if (kind == K_P) {
      const int r = sh->Pi[idx], cidx = aux->Pcol[idx];
      if (mode == 0) v = w->c * w->sx[r] * w->sx[cidx] * w->Px[idx]; /* :385  c*sx(row)*sx(col)*P */
      else v = w->c * w->sx[cidx] * w->sx[r] * w->Px[idx];           /* :145  c*sx(col)*sx(row)*P */
      if (r == cidx) {
        if (mode == 0) v += sigma;      /* :389 */
        else if (mode == 1) v += delta; /* :170 */
      }
    } else if (kind == K_A) {
      const int r = aux->Arow[idx], cidx = sh->Aj[idx];
      v = w->sy[r] * w->sx[cidx] * w->Ax[idx]; /* :392 / :154 */
      if (mode != 0 && !act[r]) v = 0.0;
    } else if (kind == K_SIGMA) {
      v = (mode == 0) ? sigma : 0.0 + delta;
    } else { /* K_RHO */
      v = (mode == 0) ? (-1.0 / w->rho[idx]) : 0.0 - delta; /* :395 / :171 */
    }
    Kval[p] = v;
  }
}

static int sp_polish(sp_work *w, const sp_aux *aux) /* qp_solver.hpp:92-204, sparse, embedded */
{
  const sp_shared *sh = w->sh;
  const ksym *s       = sh->s;
  const int n = sh->n, m = sh->m, k = s->k;
  const double inf = INFINITY, eps = DBL_EPSILON;
  char *act   = (char *)calloc((size_t)m, 1); /* 1 lower-active, 2 upper-active */
  for (int i = 0; i < m; ++i) {
    if (w->dual[i] < -100 * eps && w->l[i] != -inf) act[i] = 1;
    if (w->dual[i] > 100 * eps && w->u[i] != inf) act[i] = 2; /* (:115-116; both cannot hold) */
  }
  double *h  = (double *)calloc((size_t)k, sizeof(double));
  double *tv = (double *)calloc((size_t)k, sizeof(double));
  sp_fill_K2(w, aux, 1, act, w->Kval);
  for (int j = 0; j < n; ++j) h[j] = -w->c * (w->sx[j] * w->q[j]); /* :180 */
  for (int i = 0; i < m; ++i) {                                   /* :181-182 */
    if (act[i] == 1) h[n + i] = w->sy[i] * w->l[i];
    else if (act[i] == 2) h[n + i] = w->sy[i] * w->u[i];
  }
  int ok = ldl_numeric(s, w->Kval, w->Lx, w->D, w->work); /* :187-190 */
  if (ok) {
    for (int j = 0; j < k; ++j) w->Dinv[j] = 1.0 / w->D[j];
    for (uint32_t it = 0; it != w->prm.polish_iter; ++it) { /* :193-195  t += Hp^-1 (h - Hsym t) */
      /* Hsym = selfadjointView<Upper>(H): rows of the ORIGINAL ordering, one fma chain per row:
       * primal row i: sym(triu(P)) terms (columns ascending) then active A' terms (rows ascending);
       * constraint row r: active ? A row terms in storage order : nothing. */
      for (int i = 0; i < n; ++i) {
        double acc = 0.0;
        for (int p = sh->Sp[i]; p < sh->Sp[i + 1]; ++p) {
          const int j = sh->Sj[p], e = sh->Spos[p];
          const int er = sh->Pi[e], ec = aux->Pcol[e];
          acc = fma(w->c * w->sx[ec] * w->sx[er] * w->Px[e], tv[j], acc);
        }
        for (int p = sh->Acp[i]; p < sh->Acp[i + 1]; ++p) {
          const int rr = sh->Aci[p], e = sh->Acpos[p];
          if (act[rr]) acc = fma(w->sy[rr] * w->sx[i] * w->Ax[e], tv[n + rr], acc);
        }
        w->p[i] = h[i] - acc;
      }
      for (int rr = 0; rr < m; ++rr) {
        double acc = 0.0;
        if (act[rr])
          for (int p = sh->Ap[rr]; p < sh->Ap[rr + 1]; ++p)
            acc = fma(w->sy[rr] * w->sx[sh->Aj[p]] * w->Ax[p], tv[sh->Aj[p]], acc);
        w->p[n + rr] = h[n + rr] - acc;
      }
      ldl_solve(s, w->Lx, w->Dinv, w->p, w->t);
      for (int i = 0; i < k; ++i) tv[i] += w->p[i];
    }
    for (int j = 0; j < n; ++j) w->primal[j] = tv[j];
    for (int i = 0; i < m; ++i)
      if (act[i]) w->dual[i] = tv[n + i];
  }
  free(act); free(h); free(tv);
  return ok;
}

static int sp_solve_one(const sp_shared *sh, const sp_aux *aux, const oracle_qp_params *prm, const double *Px,
                        const double *q, const double *Ax, const double *l, const double *u, const double *warm_x,
                        const double *warm_y, double *x, double *y, double *obj, uint32_t *iter_out, int32_t *code_out,
                        double *trace, int trace_cap)
{
  const ksym *s = sh->s;
  const int n = sh->n, m = sh->m, k = s->k;
  sp_work W;
  sp_work *w = &W;
  memset(w, 0, sizeof(*w));
  w->sh = sh; w->prm = *prm; w->Px = Px; w->q = q; w->Ax = Ax; w->l = l; w->u = u;
  w->trace = trace; w->trace_cap = trace_cap;
  for (int r = 0; trace && r < trace_cap; ++r) trace[6 * r] = -1.0;
  const size_t nd = (size_t)(6 * n + 10 * m + 5 * k) + (size_t)s->nnzK + (size_t)s->nnzL + 16;
  double *mem     = (double *)calloc(nd, sizeof(double));
  if (!mem) return -1;
  double *ptr = mem;
  w->sx = ptr; ptr += n; w->sx_inc = ptr; ptr += n; w->x_us = ptr; ptr += n; w->dx_us = ptr; ptr += n;
  w->Pxv = ptr; ptr += n; w->Aty = ptr; ptr += n;
  w->sy = ptr; ptr += m; w->sy_inc = ptr; ptr += m; w->rho = ptr; ptr += m; w->z = ptr; ptr += m;
  w->z_next = ptr; ptr += m; w->Axv = ptr; ptr += m; w->y_us = ptr; ptr += m; w->z_us = ptr; ptr += m;
  w->dy_us = ptr; ptr += m; ptr += m;
  w->p = ptr; ptr += k; w->t = ptr; ptr += k; w->D = ptr; ptr += k; w->Dinv = ptr; ptr += k; w->work = ptr; ptr += k;
  w->Kval = ptr; ptr += s->nnzK; w->Lx = ptr;
  w->primal = x; w->dual = y;
  w->c = 1.0;
  for (int j = 0; j < n; ++j) w->sx[j] = 1.0;
  for (int i = 0; i < m; ++i) w->sy[i] = 1.0;
  const double inf = INFINITY;

  if (prm->scaling) sp_scale(w);

  const double rho_bar = (double)prm->rho, alpha = (double)prm->alpha, alpha_comp = 1.0 - alpha;
  const double sigma = (double)prm->sigma;
  int ret_code = -1;
  for (int i = 0; i < m; ++i) {
    if (l[i] == inf || u[i] == -inf || u[i] - l[i] < 0.0) ret_code = ORACLE_QP_PRIMAL_INFEASIBLE;
    if (l[i] == -inf && u[i] == inf) w->rho[i] = 1e-6;
    else if (w->sy[i] * fabs(l[i] - u[i]) < 1e-5) w->rho[i] = 1e3 * rho_bar;
    else w->rho[i] = rho_bar;
  }
  struct timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0); /* :376 */
  sp_fill_K2(w, aux, 0, NULL, w->Kval);
  if (!ldl_numeric(s, w->Kval, w->Lx, w->D, w->work)) ret_code = ORACLE_QP_UNKNOWN; /* :423-433 */
  for (int j = 0; j < k; ++j) w->Dinv[j] = 1.0 / w->D[j]; /* vectorD().cwiseInverse() :458 */

  if (warm_x) { /* :436-440 */
    for (int j = 0; j < n; ++j) w->primal[j] = (1.0 / w->sx[j]) * warm_x[j];
    for (int i = 0; i < m; ++i) w->dual[i] = w->c * ((1.0 / w->sy[i]) * warm_y[i]);
    for (int i = 0; i < m; ++i) {
      double sacc = 0.0;
      for (int p = sh->Ap[i]; p < sh->Ap[i + 1]; ++p) sacc = fma(w->sy[i] * Ax[p], warm_x[sh->Aj[p]], sacc);
      w->z[i] = sacc;
    }
  } else {
    for (int j = 0; j < n; ++j) w->primal[j] = 0.0;
    for (int i = 0; i < m; ++i) { w->dual[i] = 0.0; w->z[i] = 0.0; }
  }

  uint32_t iter      = 0;
  const uint32_t sci = prm->stop_check_iter;
  for (; (prm->max_iter < 0 || (int64_t)iter != prm->max_iter) && ret_code < 0; ++iter) {
    for (int j = 0; j < n; ++j) w->p[j] = sigma * w->primal[j] - w->c * w->sx[j] * q[j];
    for (int i = 0; i < m; ++i) w->p[n + i] = w->z[i] - (1.0 / w->rho[i]) * w->dual[i];
    ldl_solve(s, w->Lx, w->Dinv, w->p, w->t);
    const int chk = (sci != 0) && (iter % sci == 1);
    if (chk) {
      memcpy(w->dx_us, w->primal, sizeof(double) * (size_t)n);
      memcpy(w->dy_us, w->dual, sizeof(double) * (size_t)m);
    }
    for (int j = 0; j < n; ++j) w->primal[j] = alpha * w->p[j] + alpha_comp * w->primal[j];
    for (int i = 0; i < m; ++i) {
      const double rinv = 1.0 / w->rho[i], nu = w->p[n + i];
      double zn = alpha * (rinv * nu) + alpha_comp * (rinv * w->dual[i]) + w->z[i];
      zn        = dmax(zn, w->sy[i] * l[i]);
      zn        = dmin(zn, w->sy[i] * u[i]);
      w->z_next[i] = zn;
      w->dual[i]   = alpha_comp * w->dual[i] + alpha * nu + w->rho[i] * w->z[i] - w->rho[i] * zn;
    }
    { double *tmp = w->z; w->z = w->z_next; w->z_next = tmp; }
    if (chk) {
      for (int j = 0; j < n; ++j) w->x_us[j] = w->sx[j] * w->primal[j];
      for (int i = 0; i < m; ++i) w->y_us[i] = w->sy[i] * w->dual[i] / w->c;
      for (int i = 0; i < m; ++i) w->z_us[i] = (1.0 / w->sy[i]) * w->z[i];
      for (int j = 0; j < n; ++j) w->dx_us[j] = w->sx[j] * (w->primal[j] - w->dx_us[j]);
      for (int i = 0; i < m; ++i) w->dy_us[i] = w->sy[i] * (w->dual[i] - w->dy_us[i]) / w->c;
      ret_code = sp_check_stopping(w);
      if (w->trace && w->trace_rows < w->trace_cap) { /* :490-501, the three columns in the reference's expressions */
        double *row = w->trace + 6 * (size_t)w->trace_rows++;
        double o = 0.0, pri = 0.0, dua = 0.0;
        sp_mv_P(w, w->x_us, w->Pxv);
        for (int j = 0; j < n; ++j) o += (0.5 * w->Pxv[j] + q[j]) * w->x_us[j];
        sp_mv_A(w, w->x_us, w->Axv);
        for (int i = 0; i < m; ++i) pri = dmax(pri, fabs(w->Axv[i] - w->z_us[i]));
        sp_mv_At(w, w->y_us, w->Aty);
        for (int j = 0; j < n; ++j) dua = dmax(dua, fabs(w->Pxv[j] + q[j] + w->Aty[j]));
        row[0] = (double)iter; row[1] = o; row[2] = pri; row[3] = dua;
        /* ... and the tolerances the two residuals are tested against (:580-590) */
        row[4] = (double)prm->eps_abs + (double)prm->eps_rel * dmax(norm_inf(w->Axv, m), norm_inf(w->z_us, m));
        row[5] = (double)prm->eps_abs + (double)prm->eps_rel * dmax(dmax(norm_inf(w->Pxv, n), norm_inf(q, n)), norm_inf(w->Aty, n));
      }
      if (ret_code < 0 && prm->max_time_ns >= 0) { /* :504-507 */
        struct timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        const int64_t el = (int64_t)(t1.tv_sec - t0.tv_sec) * 1000000000LL + (t1.tv_nsec - t0.tv_nsec);
        if (el > prm->max_time_ns) ret_code = ORACLE_QP_MAX_TIME;
      }
    }
  }
  if (ret_code == ORACLE_QP_OPTIMAL && prm->polish) (void)sp_polish(w, aux);

  *code_out = (ret_code >= 0) ? ret_code : ORACLE_QP_MAX_ITERATIONS;
  for (int j = 0; j < n; ++j) w->primal[j] = w->sx[j] * w->primal[j];
  for (int i = 0; i < m; ++i) w->dual[i] = w->sy[i] * w->dual[i] / w->c;
  if (obj) { /* primal.dot(0.5*P*primal + q), P as stored */
    double o = 0.0;
    for (int i = 0; i < n; ++i) {
      double sacc = 0.0;
      for (int p = sh->Prp[i]; p < sh->Prp[i + 1]; ++p) sacc = fma(0.5 * Px[sh->Prpos[p]], w->primal[sh->Prj[p]], sacc);
      o = fma(w->primal[i], sacc + q[i], o);
    }
    *obj = o;
  }
  if (iter_out) *iter_out = iter;
  free(mem);
  return 0;
}

/* ---------------- batch driver ---------------- */
typedef struct {
  const sp_shared *sh; const sp_aux *aux; const oracle_qp_params *prm;
  int64_t batch; int64_t *next; int nnzP, nnzA; /* items are handed out one at a time (iteration counts are heavy-tailed) */
  const double *Px, *q, *Ax, *l, *u, *wx, *wy;
  double *x, *y, *obj; uint32_t *iter; int32_t *code; int rc;
} sp_job;

static void *sp_worker(void *arg)
{
  sp_job *j = (sp_job *)arg;
  const size_t n = (size_t)j->sh->n, m = (size_t)j->sh->m;
  for (int64_t b; (b = __atomic_fetch_add(j->next, 1, __ATOMIC_RELAXED)) < j->batch;) {
    const size_t sb = (size_t)b;
    int rc = sp_solve_one(j->sh, j->aux, j->prm, j->Px + sb * (size_t)j->nnzP, j->q + sb * n,
                          j->Ax + sb * (size_t)j->nnzA, j->l + sb * m, j->u + sb * m,
                          j->wx ? j->wx + sb * n : NULL, j->wy ? j->wy + sb * m : NULL, j->x + sb * n,
                          j->y + sb * m, j->obj ? j->obj + sb : NULL, j->iter ? j->iter + sb : NULL, j->code + sb,
                          g_trace ? g_trace + 6 * sb * (size_t)g_trace_cap : NULL, g_trace_cap);
    if (rc) j->rc = rc;
  }
  return NULL;
}

int oracle_qp_sparse_solve_batch(const oracle_qp_params *prm, int64_t batch, int n, int m, const int32_t *Pp,
                                 const int32_t *Pi, const double *Px, const double *q, const int32_t *Ap,
                                 const int32_t *Aj, const double *Ax, const double *l, const double *u,
                                 const int32_t *perm, const double *warm_x, const double *warm_y, double *x,
                                 double *y, double *obj, uint32_t *iter, int32_t *code, int nthreads,
                                 int64_t *nnzL_out)
{
  return oracle_qp_sparse_solve_batch_ordered(prm, batch, n, m, Pp, Pi, Px, q, Ap, Aj, Ax, l, u, perm, NULL, warm_x, warm_y, x,
                                              y, obj, iter, code, nthreads, nnzL_out);
}

int oracle_qp_sparse_solve_batch_ordered(const oracle_qp_params *prm, int64_t batch, int n, int m, const int32_t *Pp,
                                         const int32_t *Pi, const double *Px, const double *q, const int32_t *Ap,
                                         const int32_t *Aj, const double *Ax, const double *l, const double *u,
                                         const int32_t *perm, const int32_t *forder, const double *warm_x,
                                         const double *warm_y, double *x, double *y, double *obj, uint32_t *iter,
                                         int32_t *code, int nthreads, int64_t *nnzL_out)
{
  if (!prm || n < 1 || m < 1 || batch < 0 || !Pp || !Pi || !Ap || !Aj || !code) return -1;
  ksym S;
  if (ksym_build(&S, n, m, Pp, Pi, Ap, Aj, perm, forder)) return -1;
  if (nnzL_out) *nnzL_out = S.nnzL;
  sp_shared sh;
  memset(&sh, 0, sizeof(sh));
  sh.s = &S; sh.n = n; sh.m = m; sh.k = n + m; sh.Pp = Pp; sh.Pi = Pi; sh.Ap = Ap; sh.Aj = Aj;
  build_transposed(n, Ap[m], Ap, m, Aj, &sh.Acp, &sh.Aci, &sh.Acpos);
  build_transposed(n, Pp[n], Pp, n, Pi, &sh.Prp, &sh.Prj, &sh.Prpos);
  { /* symmetric view of the upper-stored entries of P (col >= row), cf. selfadjointView<Upper> */
    int *cnt = (int *)calloc((size_t)n + 1, sizeof(int));
    for (int cidx = 0; cidx < n; ++cidx)
      for (int p = Pp[cidx]; p < Pp[cidx + 1]; ++p) {
        const int r = Pi[p];
        if (cidx >= r) { cnt[r + 1]++; if (r != cidx) cnt[cidx + 1]++; }
      }
    for (int i = 0; i < n; ++i) cnt[i + 1] += cnt[i];
    sh.Sp   = cnt;
    sh.Sj   = (int *)malloc(sizeof(int) * (size_t)(cnt[n] > 0 ? cnt[n] : 1));
    sh.Spos = (int *)malloc(sizeof(int) * (size_t)(cnt[n] > 0 ? cnt[n] : 1));
    int *fill = (int *)calloc((size_t)n, sizeof(int));
    /* row i gets: mirrored entries (j < i) from column i's upper entries... build by two passes so
     * that columns come out ascending: pass over columns ascending, each column c contributes
     * (row r, col c) to row r [c >= r] and (row c, col r) to row c [r < c]. Row c's mirrored
     * entries (cols r < c) must precede its own upper entries (cols >= c): do mirrored first. */
    for (int cidx = 0; cidx < n; ++cidx) /* mirrored: row cidx, col r < cidx (r ascending in CSC) */
      for (int p = Pp[cidx]; p < Pp[cidx + 1]; ++p) {
        const int r = Pi[p];
        if (r < cidx) { sh.Sj[sh.Sp[cidx] + fill[cidx]] = r; sh.Spos[sh.Sp[cidx] + fill[cidx]] = p; fill[cidx]++; }
      }
    for (int cidx = 0; cidx < n; ++cidx) /* upper: row r, col cidx >= r (cidx ascending) */
      for (int p = Pp[cidx]; p < Pp[cidx + 1]; ++p) {
        const int r = Pi[p];
        if (cidx >= r) { sh.Sj[sh.Sp[r] + fill[r]] = cidx; sh.Spos[sh.Sp[r] + fill[r]] = p; fill[r]++; }
      }
    free(fill);
  }
  sp_aux aux;
  aux.Pcol = (int *)malloc(sizeof(int) * (size_t)(Pp[n] > 0 ? Pp[n] : 1));
  aux.Arow = (int *)malloc(sizeof(int) * (size_t)(Ap[m] > 0 ? Ap[m] : 1));
  for (int cidx = 0; cidx < n; ++cidx)
    for (int p = Pp[cidx]; p < Pp[cidx + 1]; ++p) aux.Pcol[p] = cidx;
  for (int r = 0; r < m; ++r)
    for (int p = Ap[r]; p < Ap[r + 1]; ++p) aux.Arow[p] = r;

  if (nthreads < 1) nthreads = 1;
  if ((int64_t)nthreads > batch) nthreads = (int)(batch > 0 ? batch : 1);
  sp_job *jobs  = (sp_job *)calloc((size_t)nthreads, sizeof(sp_job));
  pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
  int64_t next_item = 0;
  for (int t = 0; t < nthreads; ++t) {
    sp_job *j = &jobs[t];
    j->sh = &sh; j->aux = &aux; j->prm = prm; j->nnzP = Pp[n]; j->nnzA = Ap[m];
    j->batch = batch; j->next = &next_item;
    j->Px = Px; j->q = q; j->Ax = Ax; j->l = l; j->u = u; j->wx = warm_x; j->wy = warm_y;
    j->x = x; j->y = y; j->obj = obj; j->iter = iter; j->code = code;
  }
  if (nthreads == 1) sp_worker(&jobs[0]);
  else {
    for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, sp_worker, &jobs[t]);
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
  }
  int rc = 0;
  for (int t = 0; t < nthreads; ++t)
    if (jobs[t].rc) rc = jobs[t].rc;
  free(jobs); free(th); free(aux.Pcol); free(aux.Arow);
  free(sh.Acp); free(sh.Aci); free(sh.Acpos); free(sh.Prp); free(sh.Prj); free(sh.Prpos); free(sh.Sp); free(sh.Sj); free(sh.Spos);
  ksym_free(&S);
  return rc;
}
