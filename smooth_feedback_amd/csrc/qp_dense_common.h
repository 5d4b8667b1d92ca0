// Building blocks shared by the dense QP kernels (qp_dense.hip: one QP per wavefront, k <= 64;
// qp_dense4.hip: four QPs per wavefront, k <= 32).  All of them are WAVE-WIDE routines working on one
// problem whose data sits in LDS (lane i owns row/column i); arithmetic order is the oracle's
// (oracle/qp_oracle.c), see the header of qp_dense.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>

#include "../../include/sfb.h"
#include "ldlt_wave.h"
#include "qp_dense_kernel.h"
#include "wave_util.h"

namespace sfb {

struct Lds {
  double *W;
  const double *P, *A;  // the problem's P and A: LDS copies, or (GPA kernels) the caller's arrays in global memory
  double *q, *l, *u, *sx, *sy, *rho, *xv, *yv, *zus, *dxus, *dyus, *temp;
  int *perm, *LU;
};

// f(e, base[e * stride]) for e = 0 .. count-1 IN ORDER.  GPA == false: the plain loop (P / A in LDS).  GPA == true: P / A
// are read from global memory -- the loads of eight elements are issued together and the calls follow in order, so a
// sequential chain over a row costs count / 8 memory round trips instead of count.
template<bool GPA, class F>
__device__ __forceinline__ void pa_run(const double *base, const int stride, const int count, F &&f)
{
  if constexpr (!GPA) {
    for (int e = 0; e < count; ++e) f(e, base[e * stride]);
  } else {
    constexpr int U = 8;
    for (int e0 = 0; e0 < count; e0 += U) {
      double a[U];
#pragma unroll
      for (int u = 0; u < U; ++u) a[u] = (e0 + u < count) ? base[(e0 + u) * stride] : 0.0;
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (e0 + u < count) f(e0 + u, a[u]);
    }
  }
}

// pa_global: the layout without LDS copies of P and A (s.P / s.A are set by qp_setup<true> to the caller's arrays)
__device__ __forceinline__ Lds carve(double *base, int n, int m, int k, const bool pa_global = false)
{
  Lds s;
  double *p = base;
  s.W    = p; p += (k * (k + 1)) >> 1;
  s.P    = p; p += pa_global ? 0 : n * n;
  s.A    = p; p += pa_global ? 0 : m * n;
  s.q    = p; p += n;
  s.l    = p; p += m;
  s.u    = p; p += m;
  s.sx   = p; p += n;
  s.sy   = p; p += m;
  s.rho  = p; p += m;
  s.xv   = p; p += n;
  s.yv   = p; p += m;
  s.zus  = p; p += m;
  s.dxus = p; p += n;
  s.dyus = p; p += m;
  s.temp = p; p += k;
  s.perm = reinterpret_cast<int *>(p);
  s.LU   = s.perm + k;
  return s;
}

// QPSolver::scale, qp_solver.hpp:673-730.  Lane j<n owns column j (sx_j), lane n+i owns row i (sy_i).
template<bool GPA = false>
__device__ inline double qp_scale(const Lds &s, const int n, const int m, const int lane)
{
  const int k     = n + m;
  const bool isx  = lane < n;
  const bool isc  = lane >= n && lane < k;
  const int ci    = lane - n;
  // :675-690
  if (isx) s.sx[lane] = 1.0;
  if (isc) s.sy[ci] = 1.0;
  if (isx) {
    double t = 0.0;
    pa_run<GPA>(s.P + lane * n, 1, n, [&](int, double p) { t = fmax(t, fabs(p)); });
    if (t == 0.0) t = 1.0;
    s.temp[lane] = t;
  }
  wave_lds_fence();
  // :693
  double sum = s.temp[0];
  for (int j = 1; j < n; ++j) sum += s.temp[j];
  const double mean = sum / (double)n;
  double qn         = 0.0;
  for (int j = 0; j < n; ++j) qn = fmax(qn, fabs(s.q[j]));
  const double c = 1.0 / fmax(fmax(1e-6, mean), qn);
  wave_lds_fence();

  int iter = 0;
  double crit;
  do {  // :698-729
    double inc = 0.0;
    if (isx) {
      const double sxc = s.sx[lane];
      pa_run<GPA>(s.P + lane * n, 1, n, [&](int row, double p) { inc = fmax(inc, fabs(c * s.sx[row] * sxc * p)); });
      pa_run<GPA>(s.A + lane * m, 1, m, [&](int row, double a) { inc = fmax(inc, fabs(s.sy[row] * sxc * a)); });
    } else if (isc) {
      const double syr = s.sy[ci];
      pa_run<GPA>(s.A + ci, m, n, [&](int col, double a) { inc = fmax(inc, fabs(syr * s.sx[col] * a)); });
    }
    if (inc == 0.0) inc = 1.0;
    wave_lds_fence();  // every lane has read the old sx/sy
    const double f = sqrt(1.0 / fmax(inc, 1e-8));
    if (isx) s.sx[lane] = f * s.sx[lane];
    if (isc) s.sy[ci] = f * s.sy[ci];
    crit = wave_max((isx || isc) ? fabs(inc - 1.0) : 0.0);
    wave_lds_fence();
  } while (iter++ < 10 && crit > 0.1);
  return c;
}

// rows of the original-order mat-vecs, fixed accumulation order (ascending inner index, fma)
template<bool GPA = false>
__device__ __forceinline__ double row_A(const Lds &s, int n, int m, int i, const double *v)
{
  double r = 0.0;
  pa_run<GPA>(s.A + i, m, n, [&](int j, double a) { r = fma(a, v[j], r); });
  return r;
}
template<bool GPA = false>
__device__ __forceinline__ double row_At(const Lds &s, int n, int m, int j, const double *v)
{
  (void)n;
  double r = 0.0;
  pa_run<GPA>(s.A + j * m, 1, m, [&](int i, double a) { r = fma(a, v[i], r); });
  return r;
}
template<bool GPA = false>
__device__ __forceinline__ double row_P(const Lds &s, int n, int i, const double *v)
{
  double r = 0.0;
  pa_run<GPA>(s.P + i, n, n, [&](int j, double p) { r = fma(p, v[j], r); });
  return r;
}

// QPSolver::check_stopping, qp_solver.hpp:574-644 on xv(=x_us), yv(=y_us), zus, dxus, dyus in LDS.
// Returns a QPSolutionStatus or -1 (std::nullopt).  Wave-uniform.
template<bool GPA = false>
__device__ inline int qp_check_stopping(const Lds &s, const DenseKernelParams &kp, const int n, const int m,
                                        const int lane)
{
  const bool ln = lane < n, lm = lane < m;
  const double inf = INFINITY;

  // OPTIMALITY :584-594
  const double Ax      = lm ? row_A<GPA>(s, n, m, lane, s.xv) : 0.0;
  const double Ax_norm = wave_max(fabs(Ax));
  const double zi      = lm ? s.zus[lane] : 0.0;
  const double r_norm  = wave_max(lm ? fabs(Ax - zi) : 0.0);
  const double z_norm  = wave_max(fabs(zi));
  if (r_norm <= kp.eps_abs + kp.eps_rel * fmax(Ax_norm, z_norm)) {
    const double Px  = ln ? row_P<GPA>(s, n, lane, s.xv) : 0.0;
    const double Aty = ln ? row_At<GPA>(s, n, m, lane, s.yv) : 0.0;
    const double qi  = ln ? s.q[lane] : 0.0;
    const double dual_scale = fmax(fmax(wave_max(fabs(Px)), wave_max(fabs(qi))), wave_max(fabs(Aty)));
    const double res        = ln ? Px + (qi + Aty) : 0.0;
    if (wave_max(fabs(res)) <= kp.eps_abs + kp.eps_rel * dual_scale) return SFB_QP_OPTIMAL;
  }

  // PRIMAL INFEASIBILITY :598-621
  {
    const double Aty      = ln ? row_At<GPA>(s, n, m, lane, s.dyus) : 0.0;
    const double Aty_norm = wave_max(fabs(Aty));
    const double Edy_norm = wave_max(lm ? fabs(s.dyus[lane]) : 0.0);
    const double thr      = kp.eps_pinf * Edy_norm;
    double acc            = 0.0;  // sequential with early exit: every lane runs the same scalar loop
    for (int i = 0; i < m; ++i) {
      const double ui = s.u[i], li = s.l[i], dyi = s.dyus[i];
      if (ui != inf) {
        acc += ui * fmax(0.0, dyi);
      } else if (dyi > thr) {
        acc = inf;
        break;
      }
      if (li != -inf) {
        acc += li * fmin(0.0, dyi);
      } else if (dyi < -thr) {
        acc = inf;
        break;
      }
    }
    // std::max(a,b) = (a<b)?b:a
    const double mxv = (Aty_norm < acc) ? acc : Aty_norm;
    if (mxv < thr) return SFB_QP_PRIMAL_INFEASIBLE;
  }

  // DUAL INFEASIBILITY :625-641
  {
    const double Adx     = lm ? row_A<GPA>(s, n, m, lane, s.dxus) : 0.0;
    const double dx_norm = wave_max(ln ? fabs(s.dxus[lane]) : 0.0);
    const double Pdx     = ln ? row_P<GPA>(s, n, lane, s.dxus) : 0.0;
    const double Pdx_n   = wave_max(fabs(Pdx));
    double qdx           = 0.0;
    for (int j = 0; j < n; ++j) qdx = fma(s.q[j], s.dxus[j], qdx);
    const double thr = kp.eps_dinf * dx_norm;
    bool ok          = (Pdx_n <= thr) && (qdx <= thr);
    bool rowok       = true;
    if (lm) {
      const double ui = s.u[lane], li = s.l[lane];
      if (ui == inf) {
        rowok = Adx >= -thr;
      } else if (li == -inf) {
        rowok = Adx <= thr;
      } else {
        rowok = fabs(Adx) < thr;
      }
    }
    if (ok && !wave_ballot(!rowok)) return SFB_QP_DUAL_INFEASIBLE;
  }
  return -1;
}

// detail::polish_qp, qp_solver.hpp:92-204 (dense branch).  In: scaled primal in xv[n], scaled dual in
// yv[m] (LDS, original order).  Out: the same arrays updated on success.  Reuses W/perm/temp.
template<bool GPA = false>
__device__ inline void qp_polish(const Lds &s, const DenseKernelParams &kp, const int n, const int m,
                                 const double c, const int lane)
{
  const double inf = INFINITY, eps = DBL_EPSILON;
  // :113-123 active sets
  bool isL = false, isU = false;
  if (lane < m) {
    const double yi = s.yv[lane];
    isL             = (yi < -100 * eps) && (s.l[lane] != -inf);
    isU             = (yi > 100 * eps) && (s.u[lane] != inf);
  }
  const unsigned long long bL = wave_ballot(isL), bU = wave_ballot(isU);
  const int nl = __popcll(bL), nu = __popcll(bU);
  if (isL) s.LU[__popcll(bL & lanemask_lt(lane))] = lane;
  if (isU) s.LU[nl + __popcll(bU & lanemask_lt(lane))] = lane;
  const int na = nl + nu, K = n + na;
  wave_lds_fence();

  // Hp (lower, row r per lane) :159-177 and h :179-182
  double h = 0.0;
  if (lane < n) {
    const int r      = lane;
    const double sxr = s.sx[r];
    pa_run<GPA>(s.P + r * n, 1, r + 1, [&](int cc, double p) {
      double v = c * s.sx[cc] * p * sxr;
      if (cc == r) v += kp.delta;
      s.W[tri(r, cc)] = v;
    });
    h = -c * (sxr * s.q[r]);
  } else if (lane < K) {
    const int a = lane - n, row = s.LU[a];
    const double syr = s.sy[row];
    pa_run<GPA>(s.A + row, m, n, [&](int j, double av) { s.W[tri(lane, j)] = syr * av * s.sx[j]; });
    for (int j = n; j < lane; ++j) s.W[tri(lane, j)] = 0.0;
    s.W[tri(lane, lane)] = 0.0 - kp.delta;
    h                     = (a < nl) ? syr * s.l[row] : syr * s.u[row];
  }
  wave_lds_fence();

  if (!ldlt_factor_lds(K, s.W, s.perm, s.temp, lane)) return;  // :187-190

  // :192-195  t += Hp^-1 (h - Hsym t); Hsym entries are recomputed (same products as above)
  double t = 0.0;
  double *tv = s.dxus;   // K <= k scratch: dxus(n)+dyus(m) are contiguous
  double *xch = s.temp;
  for (uint32_t it = 0; it != kp.polish_iter; ++it) {
    if (lane < K) tv[lane] = t;
    wave_lds_fence();
    double res = 0.0;
    if (lane < n) {
      const int r      = lane;
      const double sxr = s.sx[r];
      double acc       = 0.0;
      // upper entry (a, b) of P: (j, r) for j < r -- column r, contiguous -- then (r, j) for j >= r -- row r, stride n
      pa_run<GPA>(s.P + r * n, 1, r, [&](int j, double p) { acc = fma(c * s.sx[j] * p * s.sx[r], tv[j], acc); });
      pa_run<GPA>(s.P + r + r * n, n, n - r, [&](int e, double p) { acc = fma(c * s.sx[r] * p * s.sx[r + e], tv[r + e], acc); });
      if constexpr (!GPA) {
        for (int a = 0; a < na; ++a) {
          const int row = s.LU[a];
          acc           = fma(s.sy[row] * s.A[row + r * m] * sxr, tv[n + a], acc);
        }
      } else {
        constexpr int U = 8;
        for (int a0 = 0; a0 < na; a0 += U) {  // gather of the active rows' entries of column r, eight at a time
          int row[U];
          double av[U];
#pragma unroll
          for (int u = 0; u < U; ++u) row[u] = (a0 + u < na) ? s.LU[a0 + u] : 0;
#pragma unroll
          for (int u = 0; u < U; ++u) av[u] = s.A[row[u] + r * m];
#pragma unroll
          for (int u = 0; u < U; ++u)
            if (a0 + u < na) acc = fma(s.sy[row[u]] * av[u] * sxr, tv[n + a0 + u], acc);
        }
      }
      res = h - acc;
    } else if (lane < K) {
      const int row    = s.LU[lane - n];
      const double syr = s.sy[row];
      double acc       = 0.0;
      pa_run<GPA>(s.A + row, m, n, [&](int j, double av) { acc = fma(syr * av * s.sx[j], tv[j], acc); });
      res = h - acc;
    }
    wave_lds_fence();
    const double d = ldlt_solve_lds(K, s.W, s.perm, xch, res, lane);
    t += d;
  }
  // :199-201
  if (lane < n) s.xv[lane] = t;
  else if (lane < K) s.yv[s.LU[lane - n]] = t;
  wave_lds_fence();
}


// Problem b -> LDS, Ruiz scaling (:347), feasibility pre-check and rho (:361-374), KKT matrix
// (:399-404), pivoted LDL' (:428-433).  Returns the status if the solve ends before the first
// iteration (PrimalInfeasible / Unknown), else -1.  c = cost scaling.
template<bool GPA = false>
__device__ inline int qp_setup(Lds &s, const DenseKernelParams &kp, const int n, const int m, const size_t b,
                               const QpBatch &g, const int lane, double &c)
{
  const int k      = n + m;
  const double inf = INFINITY;
  // ---- load the problem (coalesced, batch-major contiguous) ----
  {
    const double *P = g.P + b * (size_t)(n * n);
    const double *A = g.A + b * (size_t)(m * n);
    if constexpr (GPA) {  // no LDS copies: the routines below read the caller's arrays (pa_run)
      s.P = P;
      s.A = A;
    } else {
      double *Pl = const_cast<double *>(s.P), *Al = const_cast<double *>(s.A);  // (carve() gave them LDS room)
      for (int i = lane; i < n * n; i += kWave) Pl[i] = P[i];
      for (int i = lane; i < m * n; i += kWave) Al[i] = A[i];
    }
    if (lane < n) s.q[lane] = g.q[b * n + lane];
    if (lane < m) {
      s.l[lane] = g.l[b * m + lane];
      s.u[lane] = g.u[b * m + lane];
    }
    if (lane < n) s.sx[lane] = 1.0;  // analyze(): :306-308
    if (lane < m) s.sy[lane] = 1.0;
  }
  wave_lds_fence();

  // ---- scaling :347 ----
  c = 1.0;
  if (kp.scaling) c = qp_scale<GPA>(s, n, m, lane);

  // ---- feasibility pre-check and rho :361-374 (lane i < m owns constraint i) ----
  int ret_code = -1;
  {
    bool bad = false;
    if (lane < m) {
      const double li = s.l[lane], ui = s.u[lane];
      bad = (li == inf) || (ui == -inf) || (ui - li < 0.0);
      double rho;
      if (li == -inf && ui == inf) {
        rho = 1e-6;
      } else if (s.sy[lane] * fabs(li - ui) < 1e-5) {
        rho = 1e3 * kp.rho_bar;
      } else {
        rho = kp.rho_bar;
      }
      s.rho[lane] = rho;
    }
    if (wave_ballot(bad)) ret_code = SFB_QP_PRIMAL_INFEASIBLE;
  }
  wave_lds_fence();

  // ---- KKT (lower triangle, row r per lane) :399-404 ----
  if (lane < n) {
    const int r      = lane;
    const double sxr = s.sx[r];
    pa_run<GPA>(s.P + r * n, 1, r + 1, [&](int cc, double p) {
      double v = c * s.sx[cc] * p * sxr;
      if (cc == r) v += kp.sigma;
      s.W[tri(r, cc)] = v;
    });
  } else if (lane < k) {
    const int i      = lane - n;
    const double syi = s.sy[i];
    pa_run<GPA>(s.A + i, m, n, [&](int j, double av) { s.W[tri(lane, j)] = syi * av * s.sx[j]; });
    for (int j = n; j < lane; ++j) s.W[tri(lane, j)] = 0.0;
    s.W[tri(lane, lane)] = 1.0 / (-s.rho[i]);
  }
  wave_lds_fence();

  // ---- pivoted LDL' :428-433 ----
  if (!ldlt_factor_lds(k, s.W, s.perm, s.temp, lane)) ret_code = SFB_QP_UNKNOWN;
  return ret_code;
}

// End of QPSolver::solve (:515-548).  In: the scaled iterate in s.xv / s.yv (original order).
template<bool GPA = false>
__device__ inline void qp_finish(const Lds &s, const DenseKernelParams &kp, const int n, const int m, const double c,
                                 const size_t b, const QpBatch &g, const int lane, const int ret_code,
                                 const uint32_t iter)
{
  // ---- polish :515-539 (a failed polish leaves Optimal, cf. :537 vs :544) ----
  if (ret_code == SFB_QP_OPTIMAL && kp.polish) qp_polish<GPA>(s, kp, n, m, c, lane);

  // ---- un-scale and report :544-548 ----
  double xo = 0.0;
  if (lane < n) {
    xo              = s.sx[lane] * s.xv[lane];
    g.x[b * n + lane] = xo;
    s.dxus[lane]    = xo;
  }
  if (lane < m) g.y[b * m + lane] = s.sy[lane] * s.yv[lane] / c;
  wave_lds_fence();
  if (g.obj != nullptr) {
    if (lane < n) {
      double acc = 0.0;
      pa_run<GPA>(s.P + lane, n, n, [&](int j, double p) { acc = fma(0.5 * p, s.dxus[j], acc); });
      s.temp[lane] = acc + s.q[lane];
    }
    wave_lds_fence();
    if (lane == 0) {
      double o = 0.0;
      for (int i = 0; i < n; ++i) o = fma(s.dxus[i], s.temp[i], o);
      g.obj[b] = o;
    }
  }
  if (lane == 0) {
    g.code[b] = (ret_code >= 0) ? ret_code : SFB_QP_MAX_ITERATIONS;
    if (g.iter != nullptr) g.iter[b] = iter;
  }
  wave_lds_fence();
}

}  // namespace sfb
