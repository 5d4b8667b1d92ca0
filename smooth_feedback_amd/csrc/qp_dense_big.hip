// Dense ADMM QP solver for gfx950, problems with 64 < n+m <= 1024: ONE QP PER WAVEFRONT, the KKT matrix and its
// pivoted LDL' factor in the QP's HBM workspace (they no longer fit a lane-per-row register / LDS layout), vectors in
// LDS, every lane owning rows lane, lane + 64, ...
//
// Replaces, per batch item, smooth::feedback::solve_qp for QuadraticProgram<M,N,double> (reference
// qp_solver.hpp:343-568, :574-644, :673-730, :92-204) with the DENSE branch's Eigen::LDLT<.,Upper> (:259,:428,:462):
// the reference's own ASIF example and test live at these sizes (examples/mpc_asif_vehicle.cpp:105-129: n = 3,
// m = 203; tests/test_asif.cpp:103-131: n = 4, m = 301).
//
// Arithmetic = oracle/qp_oracle.c, operation for operation (bit-identical results): association of every
// element-wise expression, k-ascending fma chains for every dot product, Eigen 3.4's unblocked pivoted LDL'
// (left-looking, pivot = first largest |entry| of the NOT yet updated trailing diagonal, dot-then-subtract, true
// divisions, zero-pivot bookkeeping), triangular solves in the oracle's per-row order -- executed column by column,
// which gives every row the same ascending (forward) / descending (backward) update sequence.  Compiled with
// -ffp-contract=off; fma() only where the oracle spells it.
//
// Parallelism inside a QP: rows over lanes for the dot products of the factorisation (each row a sequential chain),
// entries over lanes for fills and element-wise phases.  The triangular sweeps are k dependent steps each; the
// forward sweep reads a transposed copy of L (column j contiguous), the backward sweep L itself (row j contiguous),
// with the next step's entries in flight while the current one updates the LDS-resident vector.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>

#include "../../include/sfb.h"
#include "qp_dense_kernel.h"
#include "wave_util.h"

namespace sfb {

namespace {

constexpr int kBigMaxK = 1024;
constexpr int kRB      = kBigMaxK / kWave;  // row blocks per lane at the largest size

__device__ __forceinline__ int ubig(const int v) { return __builtin_amdgcn_readfirstlane(v); }

// smallest index among the lanes' candidates (cand >= 0), wave-uniform
__device__ __forceinline__ int wave_min_index(const int cand)
{
  return (int)(-wave_max(-(double)cand));  // exact: indices are far below 2^53
}

// Eigen 3.4 LDLT<., Upper> == oracle_ldlt_factor (oracle/qp_oracle.c:67-151) on the row-major lower triangle W
// (leading dimension ld, K x K), in global memory.  perm[K] (LDS): composed transpositions, (P b)[i] = b[perm[i]].
// temp[K] (LDS) scratch.  Returns 1 on success, 0 on failure (info() == NumericalIssue).  Wave-uniform.
__device__ inline int big_ldlt_factor(const int K, double *__restrict__ W, const int ld, int *perm, double *temp,
                                      const int lane)
{
#define WB(i, j) W[(size_t)(i) * (size_t)ld + (size_t)(j)]
  for (int i = lane; i < K; i += kWave) perm[i] = i;
  wave_sync();
  if (K <= 1) return 1;
  int found_zero = 0, ret = 1;
  for (int kk = 0; kk < K; ++kk) {
    // pivot: first index of the largest |diag| among rows kk..K-1 (strict '>' scan; a NaN at kk keeps kk)
    const double dkk = fabs(WB(kk, kk));
    int p            = kk;
    if (!(dkk != dkk)) {
      double best = -1.0;
      int bi      = K;
      for (int i = kk + lane; i < K; i += kWave) {
        const double a = fabs(WB(i, i));
        if (a > best) { best = a; bi = i; }
      }
      const double mx = wave_max(best);
      p               = wave_min_index((best == mx) ? bi : K);
      if (p >= K) p = kk;
    }
    if (p != kk) {
      for (int t = lane; t < kk; t += kWave) { const double a = WB(kk, t); WB(kk, t) = WB(p, t); WB(p, t) = a; }
      for (int i = p + 1 + lane; i < K; i += kWave) { const double a = WB(i, kk); WB(i, kk) = WB(i, p); WB(i, p) = a; }
      for (int i = kk + 1 + lane; i < p; i += kWave) { const double a = WB(i, kk); WB(i, kk) = WB(p, i); WB(p, i) = a; }
      if (lane == 0) {
        const double a = WB(kk, kk); WB(kk, kk) = WB(p, p); WB(p, p) = a;
        const int t = perm[kk]; perm[kk] = perm[p]; perm[p] = t;
      }
      wave_sync();
    }
    // temp(j) = D(j) * L(kk, j), j < kk
    for (int j = lane; j < kk; j += kWave) temp[j] = WB(j, j) * WB(kk, j);
    wave_sync();
    if (kk > 0) {
      for (int i = kk + lane; i < K; i += kWave) {  // rows kk (the diagonal) .. K-1: dot, then subtract
        const double *row = &WB(i, 0);
        double s          = 0.0;
        for (int j = 0; j < kk; ++j) s = fma(row[j], temp[j], s);
        WB(i, kk) -= s;
      }
      wave_sync();
    }
    const double akk = WB(kk, kk);
    const bool valid = fabs(akk) > 0.0;
    if (kk == 0 && !valid) {  // whole diagonal zero: success iff the strictly lower triangle is zero (perm = identity)
      bool nz = false;
      for (int i = lane; i < K; i += kWave)
        for (int j = 0; j < i; ++j) nz = nz || !(WB(i, j) == 0.0);
      return wave_ballot(nz) ? 0 : 1;
    }
    bool nzcol = false;
    for (int i = kk + 1 + lane; i < K; i += kWave) {
      if (valid) WB(i, kk) /= akk;
      else nzcol = nzcol || !(WB(i, kk) == 0.0);
    }
    if (!valid && wave_ballot(nzcol)) ret = 0;
    if (found_zero && valid) ret = 0;
    else if (!valid) found_zero = 1;
    wave_sync();
  }
  return ret;
#undef WB
}

// LT(j, i) = W(i, j) for i > j (column j of L contiguous), Dg[i] = W(i, i)
__device__ inline void big_transpose(const int K, const double *__restrict__ W, const int ld, double *__restrict__ LT,
                                     double *__restrict__ Dg, const int lane)
{
  for (int i = 0; i < K; ++i)
    for (int j = lane; j < i; j += kWave) LT[(size_t)j * ld + i] = W[(size_t)i * ld + j];
  for (int i = lane; i < K; i += kWave) Dg[i] = W[(size_t)i * ld + i];
  wave_sync();
}

// LDLT::_solve_impl == oracle_ldlt_solve: t (LDS, K entries, original order) <- P^T L^-T D^-1 L^-1 P t.
// Forward: column j ascending pushes into rows i > j (every row sees j ascending); backward: row j descending
// pushes into columns i < j (every entry sees j descending); |d| <= DBL_MIN -> 0, true division.
template<int RB>
__device__ inline void big_solve(const int K, const double *__restrict__ W, const double *__restrict__ LT,
                                 const double *__restrict__ Dg, const int ld, const int *perm, double *t, double *temp,
                                 const int lane)
{
  for (int i = lane; i < K; i += kWave) temp[i] = t[perm[i]];
  wave_sync();
  for (int i = lane; i < K; i += kWave) t[i] = temp[i];
  wave_sync();
  double cur[RB], nxt[RB];
  auto load_col = [&](double (&v)[RB], const int j) {  // L(i, j), i = lane + 64 r, i > j
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int i = lane + kWave * r;
      v[r]        = (i > j && i < K) ? LT[(size_t)j * ld + i] : 0.0;
    }
  };
  load_col(cur, 0);
  for (int j = 0; j < K - 1; ++j) {
    if (j + 1 < K - 1) load_col(nxt, j + 1);
    const double tj = t[j];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int i = lane + kWave * r;
      if (i > j && i < K) t[i] = fma(-cur[r], tj, t[i]);
    }
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < RB; ++r) cur[r] = nxt[r];
  }
  for (int i = lane; i < K; i += kWave) {
    const double d = Dg[i];
    t[i]           = (fabs(d) > DBL_MIN) ? t[i] / d : 0.0;
  }
  wave_lds_fence();
  auto load_row = [&](double (&v)[RB], const int j) {  // L(j, i), i = lane + 64 r, i < j
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int i = lane + kWave * r;
      v[r]        = (i < j) ? W[(size_t)j * ld + i] : 0.0;
    }
  };
  load_row(cur, K - 1);
  for (int j = K - 1; j > 0; --j) {
    if (j - 1 > 0) load_row(nxt, j - 1);
    const double tj = t[j];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int i = lane + kWave * r;
      if (i < j) t[i] = fma(-cur[r], tj, t[i]);
    }
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < RB; ++r) cur[r] = nxt[r];
  }
  for (int i = lane; i < K; i += kWave) temp[perm[i]] = t[i];
  wave_sync();
  for (int i = lane; i < K; i += kWave) t[i] = temp[i];
  wave_sync();
}

__device__ __forceinline__ double big_norm_inf(const double *v, const int len, const int lane)
{
  double r = 0.0;
  for (int e = lane; e < len; e += kWave) r = fmax(r, fabs(v[e]));
  return wave_max(r);
}

struct BigWs {
  double *H, *LT, *Hs, *Dg;                                                // k*k each, k
  double *sx, *xus, *dxus, *Px, *Aty, *hx;                                 // n each
  double *sy, *rho, *rinv, *lo, *hi, *yus, *zus, *dyus, *Ax, *act, *pos;   // m each
};

}  // namespace

size_t qp_dense_big_ws_doubles(int n, int m)
{
  const size_t k = (size_t)n + m;
  return 3 * k * k + k + 6 * (size_t)n + 11 * (size_t)m + 8;
}
size_t qp_dense_big_lds_bytes(int n, int m)
{
  const size_t k = (size_t)n + m;
  return (3 * k + (size_t)n + 2 * (size_t)m + 8) * sizeof(double) + ((k + 1) / 2 * 2) * sizeof(int);
}

template<int RB>
__global__ void __launch_bounds__(64) qp_dense_big_kernel(const DenseKernelParams kp, const QpBatch g, double *__restrict__ gws,
                                                         const size_t wsd)
{
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int lane = threadIdx.x;
  const int n = kp.n, m = kp.m, k = n + m;
  const size_t b = blockIdx.x;
  const double *P = g.P + b * (size_t)n * n, *q = g.q + b * (size_t)n, *A = g.A + b * (size_t)m * n;
  const double *l = g.l + b * (size_t)m, *u = g.u + b * (size_t)m;
  // LDS: work vector t (k), scratch temp (k + 8: the polish system is never larger than k), second scratch (k),
  // iterate x (n), y (m), z (m), permutation (k ints)
  double *t = sm, *temp = t + k, *aux = temp + k + 8, *xs = aux + k, *ys = xs + n, *zs = ys + m;
  int *perm = reinterpret_cast<int *>(zs + m);
  BigWs w;
  {
    double *p = gws + b * wsd;
    const size_t kk2 = (size_t)k * k;
    w.H = p; p += kk2;  w.LT = p; p += kk2;  w.Hs = p; p += kk2;  w.Dg = p; p += k;
    w.sx = p; p += n;   w.xus = p; p += n;   w.dxus = p; p += n;  w.Px = p; p += n;  w.Aty = p; p += n;  w.hx = p; p += n;
    w.sy = p; p += m;   w.rho = p; p += m;   w.rinv = p; p += m;  w.lo = p; p += m;  w.hi = p; p += m;
    w.yus = p; p += m;  w.zus = p; p += m;   w.dyus = p; p += m;  w.Ax = p; p += m;  w.act = p; p += m;  w.pos = p; p += m;
  }
  const double inf = INFINITY;
#define PM(i, j) P[(size_t)(i) + (size_t)(j) * (size_t)n]
#define AM(i, j) A[(size_t)(i) + (size_t)(j) * (size_t)m]

  // ---- analyze(): :306-308 ----
  for (int j = lane; j < n; j += kWave) w.sx[j] = 1.0;
  for (int i = lane; i < m; i += kWave) w.sy[i] = 1.0;
  wave_sync();
  double c = 1.0;

  // ---- scale :673-730 ----
  if (kp.scaling) {
    for (int col = 0; col < n; ++col) {  // :681-690 column inf-norms of P
      double v = 0.0;
      for (int row = lane; row < n; row += kWave) v = fmax(v, fabs(PM(row, col)));
      v = wave_max(v);
      if (v == 0.0) v = 1.0;
      if (lane == 0) temp[col] = v;
    }
    wave_sync();
    double sum = temp[0];
    for (int j = 1; j < n; ++j) sum += temp[j];
    const double qn = big_norm_inf(q, n, lane);
    c               = 1.0 / fmax(fmax(1e-6, sum / (double)n), qn);  // :693
    wave_sync();
    int pass = 0;
    double crit;
    do {  // :698-729: increments into temp[0..n) (columns) and aux[0..m) (rows) from the OLD sx, sy
      for (int i = lane; i < m; i += kWave) aux[i] = 0.0;
      wave_sync();
      for (int col = 0; col < n; ++col) {
        const double sxc = w.sx[col];
        double v         = 0.0;
        for (int row = lane; row < n; row += kWave) v = fmax(v, fabs(c * w.sx[row] * sxc * PM(row, col)));  // :704-707
        for (int row = lane; row < m; row += kWave) {                                                        // :712-714
          const double a = fabs(w.sy[row] * sxc * AM(row, col));
          v              = fmax(v, a);
          aux[row]       = fmax(aux[row], a);
        }
        v = wave_max(v);
        if (lane == 0) temp[col] = v;
      }
      wave_sync();
      double cm = 0.0;
      for (int j = lane; j < n; j += kWave) {
        double inc = temp[j];
        if (inc == 0.0) inc = 1.0;
        cm      = fmax(cm, fabs(inc - 1.0));
        w.sx[j] = sqrt(1.0 / fmax(inc, 1e-8)) * w.sx[j];
      }
      for (int i = lane; i < m; i += kWave) {
        double inc = aux[i];
        if (inc == 0.0) inc = 1.0;
        cm      = fmax(cm, fabs(inc - 1.0));
        w.sy[i] = sqrt(1.0 / fmax(inc, 1e-8)) * w.sy[i];
      }
      crit = wave_max(cm);
      wave_sync();
    } while (pass++ < 10 && crit > 0.1);
  }

  // ---- pre-check and rho :361-374 ----
  int ret_code = -1;
  {
    bool bad = false;
    for (int i = lane; i < m; i += kWave) {
      const double li = l[i], ui = u[i], syi = w.sy[i];
      bad = bad || (li == inf) || (ui == -inf) || (ui - li < 0.0);
      double rho;
      if (li == -inf && ui == inf) rho = 1e-6;
      else if (syi * fabs(li - ui) < 1e-5) rho = 1e3 * kp.rho_bar;
      else rho = kp.rho_bar;
      w.rho[i]  = rho;
      w.rinv[i] = 1.0 / rho;
      w.lo[i]   = syi * li;
      w.hi[i]   = syi * ui;
    }
    if (wave_ballot(bad)) ret_code = SFB_QP_PRIMAL_INFEASIBLE;
  }
  wave_sync();

  const unsigned long long t0_ticks = wall_clock64();  // :376
  // ---- dense KKT fill :399-404 (row-major lower triangle of the k x k array H) ----
  for (int r = 0; r < n; ++r)
    for (int cc = lane; cc <= r; cc += kWave) {
      double v = c * w.sx[cc] * PM(cc, r) * w.sx[r];
      if (cc == r) v += kp.sigma;
      w.H[(size_t)r * k + cc] = v;
    }
  for (int e = lane; e < m * n; e += kWave) {
    const int i = e % m, j = e / m;
    w.H[(size_t)(n + i) * k + j] = w.sy[i] * A[e] * w.sx[j];
  }
  for (int i = 0; i < m; ++i) {
    for (int i2 = lane; i2 < i; i2 += kWave) w.H[(size_t)(n + i) * k + (n + i2)] = 0.0;
    if (lane == 0) w.H[(size_t)(n + i) * k + (n + i)] = 1.0 / (-w.rho[i]);
  }
  wave_sync();
  if (!big_ldlt_factor(k, w.H, k, perm, temp, lane)) ret_code = SFB_QP_UNKNOWN;  // :428-433
  big_transpose(k, w.H, k, w.LT, w.Dg, lane);

  // ---- initial iterate :436-445 ----
  if (g.wx != nullptr) {
    const double *wx = g.wx + b * (size_t)n, *wy = g.wy + b * (size_t)m;
    for (int j = lane; j < n; j += kWave) xs[j] = (1.0 / w.sx[j]) * wx[j];
    for (int i = lane; i < m; i += kWave) {
      ys[i]     = c * ((1.0 / w.sy[i]) * wy[i]);
      double s  = 0.0;
      for (int j = 0; j < n; ++j) s = fma(w.sy[i] * AM(i, j), wx[j], s);
      zs[i] = s;
    }
  } else {
    for (int j = lane; j < n; j += kWave) xs[j] = 0.0;
    for (int i = lane; i < m; i += kWave) { ys[i] = 0.0; zs[i] = 0.0; }
  }
  wave_sync();

  // mat-vecs of check_stopping in the oracle's order: s = 0, inner index ascending, fma
  auto mv_A = [&](const double *v, double *out) {
    for (int i = lane; i < m; i += kWave) {
      double s = 0.0;
      for (int j = 0; j < n; ++j) s = fma(AM(i, j), v[j], s);
      out[i] = s;
    }
  };
  auto mv_At = [&](const double *v, double *out) {
    for (int j = lane; j < n; j += kWave) {
      double s = 0.0;
      for (int i = 0; i < m; ++i) s = fma(AM(i, j), v[i], s);
      out[j] = s;
    }
  };
  auto mv_P = [&](const double *v, double *out) {
    for (int i = lane; i < n; i += kWave) {
      double s = 0.0;
      for (int j = 0; j < n; ++j) s = fma(PM(i, j), v[j], s);
      out[i] = s;
    }
  };

  // ---- ADMM loop :447-510 ----
  uint32_t iter        = 0;
  const uint32_t sci   = kp.stop_check_iter;
  const uint32_t maxit = kp.max_iter;
  for (; iter != maxit && ret_code < 0; ++iter) {
    for (int j = lane; j < n; j += kWave) t[j] = kp.sigma * xs[j] - c * w.sx[j] * q[j];      // :450
    for (int i = lane; i < m; i += kWave) t[n + i] = zs[i] - w.rinv[i] * ys[i];             // :451
    wave_sync();
    big_solve<RB>(k, w.H, w.LT, w.Dg, k, perm, t, temp, lane);                              // :462
    const bool chk = (sci != 0) && (iter % sci == 1);                                       // :465
    for (int j = lane; j < n; j += kWave) {                                                 // :470
      const double xo = xs[j], xn = kp.alpha * t[j] + kp.alpha_comp * xo;
      xs[j] = xn;
      if (chk) {
        w.xus[j]  = w.sx[j] * xn;
        w.dxus[j] = w.sx[j] * (xn - xo);
      }
    }
    for (int i = lane; i < m; i += kWave) {                                                 // :471-477
      const double ri = w.rinv[i], rh = w.rho[i], yo = ys[i], zo = zs[i], nu = t[n + i];
      double zn = kp.alpha * (ri * nu) + kp.alpha_comp * (ri * yo) + zo;
      zn        = (zn < w.lo[i]) ? w.lo[i] : zn;
      zn        = (w.hi[i] < zn) ? w.hi[i] : zn;
      const double yn = kp.alpha_comp * yo + kp.alpha * nu + rh * zo - rh * zn;
      ys[i] = yn;
      zs[i] = zn;
      if (chk) {
        const double syi = w.sy[i];
        w.yus[i]  = syi * yn / c;
        w.zus[i]  = (1.0 / syi) * zn;
        w.dyus[i] = syi * (yn - yo) / c;
      }
    }
    wave_sync();
    if (chk) {  // check_stopping :574-644
      int code = -1;
      mv_A(w.xus, w.Ax);
      wave_sync();
      const double Ax_norm = big_norm_inf(w.Ax, m, lane);
      double rn = 0.0;
      for (int i = lane; i < m; i += kWave) rn = fmax(rn, fabs(w.Ax[i] - w.zus[i]));
      rn = wave_max(rn);
      if (rn <= kp.eps_abs + kp.eps_rel * fmax(Ax_norm, big_norm_inf(w.zus, m, lane))) {
        mv_P(w.xus, w.Px);
        mv_At(w.yus, w.Aty);
        wave_sync();
        const double dual_scale = fmax(fmax(big_norm_inf(w.Px, n, lane), big_norm_inf(q, n, lane)), big_norm_inf(w.Aty, n, lane));
        double dn = 0.0;
        for (int j = lane; j < n; j += kWave) dn = fmax(dn, fabs(w.Px[j] + (q[j] + w.Aty[j])));  // :592
        if (wave_max(dn) <= kp.eps_abs + kp.eps_rel * dual_scale) code = SFB_QP_OPTIMAL;
      }
      if (code < 0) {  // primal infeasibility :598-621
        mv_At(w.dyus, w.Aty);
        wave_sync();
        const double Edy = big_norm_inf(w.dyus, m, lane);
        const double thr = kp.eps_pinf * Edy;
        // the ordered sum with its early exit to +inf: +inf iff some row has an unbounded side beyond the threshold
        // BEFORE ... no: the oracle breaks at the first such row, having added the finite terms of the rows before
        // it and possibly the u-term of that row -- but then overwrites the sum with +inf, so only "any such row"
        // matters; otherwise the sum is the ordered one (a skipped term adds +0.0, exact: the sum is never -0.0)
        bool brk = false;
        for (int i = lane; i < m; i += kWave) {
          const double ui = u[i], li = l[i], dyi = w.dyus[i];
          aux[i]  = (ui != inf) ? ui * fmax(0.0, dyi) : 0.0;
          temp[i] = (li != -inf) ? li * fmin(0.0, dyi) : 0.0;
          brk = brk || (ui == inf && dyi > thr) || (li == -inf && dyi < -thr);
        }
        wave_sync();
        double s = 0.0;
        for (int i = 0; i < m; ++i) { s += aux[i]; s += temp[i]; }
        if (wave_ballot(brk)) s = inf;
        const double an = big_norm_inf(w.Aty, n, lane);
        if (((an < s) ? s : an) < thr) code = SFB_QP_PRIMAL_INFEASIBLE;
        wave_sync();
      }
      if (code < 0) {  // dual infeasibility :625-641
        mv_A(w.dxus, w.Ax);
        mv_P(w.dxus, w.Px);
        wave_sync();
        const double dxn = big_norm_inf(w.dxus, n, lane);
        const double thr = kp.eps_dinf * dxn;
        for (int j = lane; j < n; j += kWave) { aux[j] = q[j]; temp[j] = w.dxus[j]; }
        wave_sync();
        double qdx = 0.0;
        for (int j = 0; j < n; ++j) qdx = fma(aux[j], temp[j], qdx);
        bool ok = (big_norm_inf(w.Px, n, lane) <= thr) && (qdx <= thr);
        bool rowok = true;
        for (int i = lane; i < m; i += kWave) {
          const double Adx = w.Ax[i];
          if (u[i] == inf) rowok = rowok && (Adx >= -thr);
          else if (l[i] == -inf) rowok = rowok && (Adx <= thr);
          else rowok = rowok && (fabs(Adx) < thr);
        }
        if (ok && !wave_ballot(!rowok)) code = SFB_QP_DUAL_INFEASIBLE;
        wave_sync();
      }
      ret_code = code;
      if (ret_code < 0 && max_time_exceeded(kp.max_time_ns, t0_ticks)) ret_code = SFB_QP_MAX_TIME;  // :504-507
    }
  }

  // ---- polish :92-204, :515-539 (on the scaled iterate; a failed factorisation leaves the ADMM solution) ----
  if (ret_code == SFB_QP_OPTIMAL && kp.polish) {
    const double eps = DBL_EPSILON;
    // active sets: lower indices first, then upper, each ascending (:113-123)
    int nl = 0, nu = 0;
    for (int c0 = 0; c0 < m; c0 += kWave) {
      const int i = c0 + lane;
      int a       = 0;
      if (i < m) {
        const double yi = ys[i];
        if (yi < -100 * eps && l[i] != -inf) a = 1;
        if (yi > 100 * eps && u[i] != inf) a = 2;
      }
      const unsigned long long bl = wave_ballot(a == 1), bu = wave_ballot(a == 2);
      if (i < m) {
        w.act[i] = (double)a;
        w.pos[i] = (a == 1) ? (double)(nl + __popcll(bl & lanemask_lt(lane))) : (double)(nu + __popcll(bu & lanemask_lt(lane)));
      }
      nl += __popcll(bl);
      nu += __popcll(bu);
    }
    wave_sync();
    const int na = nl + nu, K = n + na;
    // Hs: the symmetric K x K matrix of the residual; Hp = Hs + diag(delta, -delta) lower row-major for the LDLT (:159-177)
    double *Hs = w.Hs, *Hp = w.H;
    for (int e = lane; e < K * K; e += kWave) { Hs[e] = 0.0; Hp[e] = 0.0; }
    wave_sync();
    for (int i = 0; i < n; ++i)
      for (int j = i + lane; j < n; j += kWave) {
        const double v           = c * w.sx[i] * PM(i, j) * w.sx[j];  // :161 upper entry (i, j)
        Hs[(size_t)i * K + j]    = v;
        Hs[(size_t)j * K + i]    = v;
        Hp[(size_t)j * K + i]    = v;
      }
    for (int e = lane; e < m * n; e += kWave) {
      const int row = e % m, j = e / m;
      const int a   = (int)w.act[row];
      if (a != 0) {
        const int col  = n + (int)w.pos[row] + (a == 2 ? nl : 0);
        const double v = w.sy[row] * A[e] * w.sx[j];  // :163
        Hs[(size_t)j * K + col] = v;
        Hs[(size_t)col * K + j] = v;
        Hp[(size_t)col * K + j] = v;
      }
    }
    wave_sync();
    for (int i = lane; i < n; i += kWave) Hp[(size_t)i * K + i] += kp.delta;
    for (int a = lane; a < na; a += kWave) Hp[(size_t)(n + a) * K + (n + a)] -= kp.delta;
    // h (:179-182) and t = 0
    for (int j = lane; j < n; j += kWave) w.hx[j] = -c * (w.sx[j] * q[j]);
    for (int i = lane; i < m; i += kWave) {
      const int a = (int)w.act[i];
      if (a == 1) w.Ax[(int)w.pos[i]] = w.sy[i] * l[i];
      else if (a == 2) w.Ax[nl + (int)w.pos[i]] = w.sy[i] * u[i];
    }
    for (int e = lane; e < K; e += kWave) aux[e] = 0.0;  // aux = t of the refinement
    wave_sync();
    if (big_ldlt_factor(K, Hp, K, perm, temp, lane)) {
      big_transpose(K, Hp, K, w.LT, w.Dg, lane);
      for (uint32_t it = 0; it != kp.polish_iter; ++it) {  // :193-195  t += Hp^-1 (h - Hs t)
        for (int i = lane; i < K; i += kWave) {
          const double *row = Hs + (size_t)i * K;
          double s          = 0.0;
          for (int j = 0; j < K; ++j) s = fma(row[j], aux[j], s);
          t[i] = ((i < n) ? w.hx[i] : w.Ax[i - n]) - s;
        }
        wave_sync();
        big_solve<RB>(K, Hp, w.LT, w.Dg, K, perm, t, temp, lane);
        for (int i = lane; i < K; i += kWave) aux[i] += t[i];
        wave_sync();
      }
      for (int j = lane; j < n; j += kWave) xs[j] = aux[j];  // :199
      for (int i = lane; i < m; i += kWave) {                // :200-201
        const int a = (int)w.act[i];
        if (a == 1) ys[i] = aux[n + (int)w.pos[i]];
        else if (a == 2) ys[i] = aux[n + nl + (int)w.pos[i]];
      }
    }
    wave_sync();
  }

  // ---- un-scale and report :544-548 ----
  double *ox = g.x + b * (size_t)n, *oy = g.y + b * (size_t)m;
  for (int j = lane; j < n; j += kWave) {
    const double v = w.sx[j] * xs[j];
    ox[j] = v;
    t[j]  = v;
  }
  for (int i = lane; i < m; i += kWave) oy[i] = w.sy[i] * ys[i] / c;
  wave_sync();
  if (g.obj != nullptr) {
    for (int i = lane; i < n; i += kWave) {
      double s = 0.0;
      for (int j = 0; j < n; ++j) s = fma(0.5 * PM(i, j), t[j], s);
      temp[i] = s + q[i];
    }
    wave_sync();
    if (lane == 0) {
      double o = 0.0;
      for (int i = 0; i < n; ++i) o = fma(t[i], temp[i], o);
      g.obj[b] = o;
    }
  }
  if (lane == 0) {
    g.code[b] = (ret_code >= 0) ? ret_code : SFB_QP_MAX_ITERATIONS;
    if (g.iter != nullptr) g.iter[b] = iter;
  }
#undef PM
#undef AM
}

hipError_t qp_dense_big_launch(const DenseKernelParams &kp, int64_t batch, const QpBatch &g, double *workspace, hipStream_t stream)
{
  const int k = kp.n + kp.m;
  if (k > kBigMaxK) return hipErrorInvalidValue;
  const size_t lds = qp_dense_big_lds_bytes(kp.n, kp.m);
  const size_t wsd = qp_dense_big_ws_doubles(kp.n, kp.m);
  const dim3 grid((unsigned)batch), block(kWave);
  const int rb = (k + kWave - 1) / kWave;
  if (rb <= 2) hipLaunchKernelGGL((qp_dense_big_kernel<2>), grid, block, lds, stream, kp, g, workspace, wsd);
  else if (rb <= 4) hipLaunchKernelGGL((qp_dense_big_kernel<4>), grid, block, lds, stream, kp, g, workspace, wsd);
  else if (rb <= 8) hipLaunchKernelGGL((qp_dense_big_kernel<8>), grid, block, lds, stream, kp, g, workspace, wsd);
  else hipLaunchKernelGGL((qp_dense_big_kernel<kRB>), grid, block, lds, stream, kp, g, workspace, wsd);
  return hipGetLastError();
}

}  // namespace sfb
