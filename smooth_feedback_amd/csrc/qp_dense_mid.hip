// Batched dense ADMM QP solver for gfx950, 32 < n+m <= 128: ONE QP PER WAVEFRONT, everything on chip.
//
// Replaces, per batch item, smooth::feedback::solve_qp for QuadraticProgram<M,N,double> (reference
// qp_solver.hpp:343-568 solve, :574-644 check_stopping, :673-730 scale, :92-204 polish_qp; Eigen 3.4 LDLT<Upper> at
// :259,428,462 and :187-195).  Arithmetic -- association of every element-wise expression, accumulation order of every
// dot product and triangular solve, true divisions, sqrt -- is that of oracle/qp_oracle.c: the two agree bit for bit.
// Compiled with -ffp-contract=off; fma() only where the oracle spells it.
//
// What is where
//   LDS (per wave = per QP): the KKT matrix / its factor L as a PACKED lower triangle T (entry (i, j), j <= i, at
//     i (i + 1) / 2 + j), the diagonal D, one scratch vector, the permutation, and ONE shared area V that holds in turn
//     the scaling vectors (setup), the un-scaled iterates of a stopping check (loop), and scaling vectors + iterate +
//     active sets (polish): 40.2 KB at (32, 64) -- four QPs per CU --, 17 KB at (20, 40) -- nine.
//   Global memory: P, q, A, l, u of the QP are read where they lie (coalesced; batches of loads in flight, pa_run).
//   Registers: lane l carries rows l and l + 64 of the PERMUTED system through the whole ADMM loop (variable, iterate,
//     constants): an iteration is right-hand side -> block sweeps (sweep_rows.h) -> update, no LDS vector traffic.
//
// Pivoting without row swaps.  Eigen's LDLT picks, at step kk, the first largest |diagonal| among rows kk.. of the NOT
// yet updated trailing matrix (oracle_ldlt_factor): the diagonal entries it compares are the ORIGINAL ones, so the whole
// sequence of transpositions follows from the original diagonal alone.  It is simulated on the diagonal held in
// registers (k wave-wide argmax steps), the KKT matrix is written into T in its final order, and the factorisation runs
// without a single swap: same entries, same operations, same order -- the arithmetic of a row never depended on where
// the row was stored.
//
// Factorisation.  Left-looking like Eigen's: column kk = W(., kk) - sum_j L(., j) temp(j), s = 0, j ascending,
// s = fma(L(i, j), temp(j), s), one row (two for k > 64) per lane.  The dot products run in chunks of 8 terms with the
// loads of the next chunk in flight; a chunk reaches up to 7 terms beyond kk, where temp is 0 and the row holds entries
// not yet eliminated: fma(finite, 0, s) == s (the running sum is never -0) -- exact as long as every value in T is
// finite, which is tracked as values are written; from the first non-finite value on the plain loops run.
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>

#include <cfloat>
#include <cmath>
#include <cstdlib>

#include "../../include/sfb.h"
#include "knobs.h"
#include "qp_dense_kernel.h"
#include "sweep_rows.h"
#include "wave_util.h"

namespace sfb {
namespace {

using rows::lds_d;
using lds_i = __attribute__((address_space(3))) int;
using lds_b = __attribute__((address_space(3))) unsigned char;

__device__ __forceinline__ int mtri(const int i) { return (i * (i + 1)) >> 1; }
__device__ __forceinline__ int muni(const int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ bool mfinite(const double v) { return fabs(v) < INFINITY; }
// the lane number from the hardware (workgroups of one wave), opaque: a value made where it is used instead of one copy of
// threadIdx.x that lives -- in scratch memory, when registers are short -- across the persistent loops of the launches below
__device__ __forceinline__ int mid_lane_now()
{
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
// a wave-uniform double as a scalar (two SGPRs instead of two VGPRs; the bits are untouched)
__device__ __forceinline__ double muni_d(const double v)
{
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)muni((int)(unsigned)b), hi = (unsigned)muni((int)(unsigned)(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// profiling build (scripts/r4/build_variant.sh prof qp_dense_mid.hip -DSFB_MID_PROF): block 0 prints where its time went
// Stamps at the phase boundaries of a solve.  C.stamps (nullptr outside the TRACE instance: the branch folds away at compile
// time, everything is inlined) receives the wall clock at points 0 .. 9 -- the per-phase times of the reference's verbose
// summary as data (qp_solver.hpp:550-565; sfb_qp_dense_solve_batch_phases).
#ifdef SFB_MID_PROF
__device__ unsigned long long g_midprof[12];
#define MP_T(i) { g_midprof[i] = wall_clock64(); if (C.stamps != nullptr && C.lane == 0) C.stamps[i] = (double)wall_clock64(); }
#else
#define MP_T(i) { if (C.stamps != nullptr && C.lane == 0) C.stamps[i] = (double)wall_clock64(); }
#endif

// waves per SIMD the k <= 48 / k <= 64 instances are compiled for (VGPR budget 168 at 3, 256 at 2)
#ifndef SFB_MID_W3
#define SFB_MID_W3 3
#endif
#ifndef SFB_MID_W4
#define SFB_MID_W4 2
#endif

// registers-only sweeps for k <= 64 (0: the LDS block engine for every size; A/B builds)
#ifndef SFB_MID_REGS
#define SFB_MID_REGS 1
#endif

#ifndef SFB_MID_CHK_KEEP
#define SFB_MID_CHK_KEEP 1
#endif

constexpr int kMidPadT = 8;  // zeros behind the packed triangle (the chunked dot products read up to 7 entries past a row)

// LDS layout in doubles (host and device)
struct MidLayout {
  int T, Dg, tmp, V, perm, total;
};
__host__ __device__ inline MidLayout mid_layout(const int n, const int m)
{
  const int k = n + m;
  MidLayout L;
  int p   = 0;
  L.T     = p;  p += (k * (k + 1)) / 2 + kMidPadT;  p = (p + 1) & ~1;
  L.Dg    = p;  p += k;                             p = (p + 1) & ~1;
  L.tmp   = p;  p += k + 8;                         p = (p + 1) & ~1;  // temp of the factorisation (padded to 8), iperm (ints), dy of a stopping check, exchange vector
  L.V     = p;  p += 2 * k + (m + 3) / 4;           p = (p + 1) & ~1;  // [sx sy | x y | active-set lists (bytes)] or [x dx y z] of a stopping check
  L.perm  = p;  p += (k + 7) / 8;                   p = (p + 1) & ~1;  // k bytes
  L.total = p;
  return L;
}

// f(e, base[e * stride]) for e = 0 .. count-1 IN ORDER; the loads of U elements are issued together (P and A are read
// from global memory: a sequential chain over a row costs count / U memory round trips).
template<int U = 16, class F>
__device__ __forceinline__ void mrun(const double *base, const int stride, const int count, F &&f)
{
  for (int e0 = 0; e0 < count; e0 += U) {
    double a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = (e0 + u < count) ? base[(size_t)(e0 + u) * stride] : 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (e0 + u < count) f(e0 + u, a[u]);
  }
}
// f(e, base[e]) for the elements e = lane, lane + 64, ... < count of a contiguous array (coalesced), U loads in flight per lane
template<int U = 8, class F>
__device__ __forceinline__ void mstream(const double *base, const int count, const int lane, F &&f)
{
  for (int b0 = 0; b0 < count; b0 += kWave * U) {
    double a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = b0 + lane + kWave * u;
      a[u]        = (e < count) ? base[e] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = b0 + lane + kWave * u;
      if (e < count) f(e, a[u]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Eigen's pivot order from the diagonal alone (see the header).  d[r] / id[r]: value and index held at position
// lane + 64 r; on return position p holds the entry Eigen's transpositions bring there (id = its original index).
template<int R>
__device__ inline void mid_pivot_order(const int K_, double (&d)[R], int (&id)[R], const int lane)
{
  const int K = muni(K_);
  // mx: the largest |d| among the positions >= kk (NaNs skipped).  It only changes when the positions holding it are used
  // up, so the wave-wide maximum is recomputed when the search for it comes back empty, not at every step.
  double mx   = 0.0;
  bool mx_set = false;
  for (int kk = 0; kk + 1 < K; ++kk) {
    const int hk = kk >> 6, lk = kk & 63;
    const double dkk = lane_bcast((R > 1 && hk) ? d[R - 1] : d[0], lk);
    int p = kk;
    if (!(dkk != dkk)) {  // (a NaN AT kk stays the maximum of Eigen's strict '>' scan; NaNs further down are skipped)
      for (int attempt = 0; attempt < 2; ++attempt) {
        if (!mx_set) {
          double am = -1.0;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const int pos = lane + kWave * r;
            am            = fmax(am, (pos >= kk && pos < K) ? fabs(d[r]) : -1.0);
          }
          mx     = wave_max(am);
          mx_set = true;
        }
        bool found = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int pos = lane + kWave * r;
          const unsigned long long bal = wave_ballot(pos >= kk && pos < K && fabs(d[r]) == mx);
          if (!found && bal) {
            p     = kWave * r + (int)__builtin_ctzll(bal);
            found = true;
          }
        }
        if (found) break;
        mx_set = false;  // the positions that held mx are all in front of kk now: next value
      }
    }
    if (p != kk) {
      const int hp = p >> 6, lp = p & 63;
      const double dp = lane_bcast((R > 1 && hp) ? d[R - 1] : d[0], lp);
      const int ip    = __builtin_amdgcn_readlane((R > 1 && hp) ? id[R - 1] : id[0], lp);
      const int ik    = __builtin_amdgcn_readlane((R > 1 && hk) ? id[R - 1] : id[0], lk);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int pos = lane + kWave * r;
        if (pos == kk) { d[r] = dp; id[r] = ip; }
        else if (pos == p) { d[r] = dkk; id[r] = ik; }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Dot products of one factorisation step: s[r] = sum_{j < 8 nch} fma(row_r[j], tmp[j], s[r]), j ascending, for the rows
// R0 .. R-1 of the lane (R0 = 1: only rows >= 64 are still being eliminated).  Chunks of 8 terms, two chunks per trip
// with the operands of the following chunk requested before the FMAs of the current one (the request past the last
// chunk is clamped to it: a harmless re-read instead of a conditional the compiler would turn into register copies).
template<int R, int R0>
__device__ __forceinline__ void mid_dots(const lds_d *const tmp, const lds_d *const (&rp)[R], const int nch, double (&s)[R])
{
  double ta[8], tb[8], va[R][8], vb[R][8];
  auto load = [&](double (&tv)[8], double (&vv)[R][8], const int c) {
#pragma unroll
    for (int e = 0; e < 8; ++e) tv[e] = tmp[8 * c + e];
#pragma unroll
    for (int r = R0; r < R; ++r) {
#pragma unroll
      for (int e = 0; e < 8; ++e) vv[r][e] = rp[r][8 * c + e];
    }
  };
  auto fmas = [&](const double (&tv)[8], const double (&vv)[R][8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int r = R0; r < R; ++r) s[r] = fma(vv[r][e], tv[e], s[r]);
    }
  };
  const int last = nch - 1;
  load(ta, va, 0);
  int c = 0;
  for (; c + 1 < nch; c += 2) {
    load(tb, vb, c + 1);
    fmas(ta, va);
    load(ta, va, (c + 2 < last) ? c + 2 : last);
    fmas(tb, vb);
  }
  if (c < nch) fmas(ta, va);
}

// Unpivoted LDL' (Eigen 3.4's unblocked algorithm = oracle_ldlt_factor without its swaps) of the K x K matrix in the
// packed triangle T, in place; D goes to Dg and the diagonal slots of T end at -0.0 (sweep_rows.h).  tmp: K + 16 doubles.
// Returns 1 on success, 0 on failure (info() == NumericalIssue).  Wave-uniform.  `fin`: every value in T (pad included) is finite.
template<int R>
__device__ inline int mid_ldlt(const int K_, lds_d *const T, lds_d *const Dg, lds_d *const tmp, const int lane, bool fin)
{
  const int K = muni(K_);
  int ret = 1, found_zero = 0;
  int ri[R];
  const lds_d *rp[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    ri[r] = lane + kWave * r;
    rp[r] = T + mtri(ri[r] < K ? ri[r] : 0);
  }
  auto finish = [&]() {
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (ri[r] < K) T[mtri(ri[r]) + ri[r]] = -0.0;
    wave_lds_fence();
  };
  if (K <= 1) {
    if (K == 1 && lane == 0) Dg[0] = T[0];
    finish();
    return 1;
  }
  for (int kk = 0; kk < K; ++kk) {
    // temp(j) = D(j) * L(kk, j), j < kk; zeros up to the next multiple of 8
    const int kk8      = (kk + 7) & ~7;
    const lds_d *const rk = T + mtri(kk);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int j = lane + kWave * r;
      if (j < kk) tmp[j] = Dg[j] * rk[j];
      else if (j < kk8) tmp[j] = 0.0;
    }
    wave_lds_fence();
    double s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) s[r] = 0.0;
    if (kk > 0) {
      if (fin) {  // (rows below kk -- already final -- ride along: their sums are not used)
        if (R > 1 && kk >= kWave) mid_dots<R, R - 1>(tmp, rp, kk8 >> 3, s);  // rows < 64 are all final
        else mid_dots<R, 0>(tmp, rp, kk8 >> 3, s);
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r)
          if (ri[r] >= kk && ri[r] < K) {
            double t = 0.0;
            for (int j = 0; j < kk; ++j) t = fma(rp[r][j], tmp[j], t);
            s[r] = t;
          }
      }
    }
    double val[R];
    bool cand[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      cand[r] = ri[r] >= kk && ri[r] < K;
      val[r]  = cand[r] ? rp[r][kk] : 0.0;
      if (kk > 0 && cand[r]) val[r] -= s[r];
    }
    const double akk = lane_bcast((R > 1 && (kk >> 6)) ? val[R - 1] : val[0], kk & 63);
    const bool valid = fabs(akk) > 0.0;
    if (kk == 0 && !valid) {  // whole diagonal zero: success iff the strictly lower triangle is zero
      bool nz = false;
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (ri[r] < K)
          for (int j = 0; j < ri[r]; ++j) nz = nz || !(rp[r][j] == 0.0);
      const int ok = wave_ballot(nz) ? 0 : 1;
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (ri[r] < K) Dg[ri[r]] = rp[r][ri[r]];
      wave_lds_fence();
      finish();
      return ok;
    }
    bool nzcol = false, f2 = mfinite(akk);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (ri[r] == kk) Dg[kk] = akk;
      if (cand[r] && ri[r] > kk) {
        double v = val[r];
        if (valid) {
          v = v / akk;
        } else {
          nzcol = nzcol || !(v == 0.0);
        }
        const_cast<lds_d *>(rp[r])[kk] = v;
        f2 = f2 && mfinite(v);
      }
    }
    if (fin && wave_ballot(!f2)) fin = false;
    if (!valid && wave_ballot(nzcol)) ret = 0;
    if (found_zero && valid) ret = 0;
    else if (!valid) found_zero = 1;
    wave_lds_fence();
  }
  finish();
  return ret;
}

// one row of the loop's state: the variable perm[row] of the KKT system, its iterate and constants
struct MidRow {
  bool isx, isc;
  int xi, ci;
  double qc, sc, rho, rinv, lo, hi, x, y, z;  // sc: the row's scale factor, sx_j or sy_i (1 for a row that does not exist) -- every use is under isx / isc
};

// ---------------------------------------------------------------------------------------------------------------
// Ordered dot products against a matrix in global memory: f(j, base[j * stride], v0[j], v1[j]) for j = 0 .. count-1 IN
// ORDER, the matrix entries and the LDS vector entries of a batch requested together (no wait, no branch per element);
// batches of U, the remainder in batches of U/2, U/4, ... 1.
template<int U, class F>
__device__ __forceinline__ void mvec_batch(const double *const base, const size_t stride, const lds_d *const v0, const lds_d *const v1, const int j0, F &&f)
{
  double a[U], x0[U], x1[U];
#pragma unroll
  for (int u = 0; u < U; ++u) a[u] = base[(size_t)(j0 + u) * stride];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    x0[u] = v0[j0 + u];
    x1[u] = v1[j0 + u];
  }
#pragma unroll
  for (int u = 0; u < U; ++u) f(j0 + u, a[u], x0[u], x1[u]);
}
template<int U = 16, class F>
__device__ __forceinline__ void mvec(const double *const base, const size_t stride, const lds_d *const v0, const lds_d *const v1, const int count, F &&f)
{
  int j = 0;
  for (; j + U <= count; j += U) mvec_batch<U>(base, stride, v0, v1, j, f);
  if constexpr (U >= 16) if (count - j >= 8) { mvec_batch<8>(base, stride, v0, v1, j, f); j += 8; }
  if constexpr (U >= 8) if (count - j >= 4) { mvec_batch<4>(base, stride, v0, v1, j, f); j += 4; }
  if constexpr (U >= 4) if (count - j >= 2) { mvec_batch<2>(base, stride, v0, v1, j, f); j += 2; }
  if (count - j >= 1) mvec_batch<1>(base, stride, v0, v1, j, f);
}

// QPSolver::check_stopping, qp_solver.hpp:574-644 (oracle qp_check_stopping) on the un-scaled iterates
// V = [x (n) | dx (n) | y (m) | z (m)], dy in tmp.  Returns a QPSolutionStatus or -1 (std::nullopt).  Wave-uniform.
// Outlined: it runs once per stop_check_iter iterations and its registers stay out of the ADMM loop's budget.
// Rows of the mat-vecs: constraint i = lane + 64 r (< m), variable j = lane + 64 r (< n); s = 0, inner index ascending, fma.
template<int R, int MU = 16>  // MU: matrix entries requested together per row (16 when the check has the registers to itself)
__device__ __forceinline__ int mid_stop_check_body(const int n_, const int m_, const int lane, const double *const P, const double *const q,
                                                        const double *const A, const double *const l, const double *const u, lds_d *const V,
                                                        lds_d *const tmp, const double eps_abs, const double eps_rel, const double eps_pinf,
                                                        const double eps_dinf)
{
  const int n = muni(n_), m = muni(m_);
  const double inf = INFINITY;
  lds_d *const xus = V, *const dxus = V + n, *const yus = V + 2 * n, *const zus = yus + m, *const dyus = tmp;
  int ci_[R], vj_[R];
  bool con[R], var[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    ci_[r] = lane + kWave * r;  con[r] = ci_[r] < m;
    vj_[r] = lane + kWave * r;  var[r] = vj_[r] < n;
  }
  // rows that do not exist read row 0 (valid memory) and drop the result: no divergent control flow around the batches
  auto mv_A2 = [&](const lds_d *v0, double (&o0)[R], const lds_d *v1, double (&o1)[R]) {  // A v0 and A v1, one pass over A
#pragma unroll
    for (int r = 0; r < R; ++r) {
      double s0 = 0.0, s1 = 0.0;
      mvec<MU>(A + (con[r] ? ci_[r] : 0), (size_t)m, v0, v1, n, [&](int, double a, double x0, double x1) { s0 = fma(a, x0, s0); s1 = fma(a, x1, s1); });
      o0[r] = con[r] ? s0 : 0.0;
      o1[r] = con[r] ? s1 : 0.0;
    }
  };
  auto mv_At = [&](const lds_d *v, double (&o)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      double s = 0.0;
      mvec<MU>(A + (size_t)(var[r] ? vj_[r] : 0) * m, 1, v, v, m, [&](int, double a, double x0, double) { s = fma(a, x0, s); });
      o[r] = var[r] ? s : 0.0;
    }
  };
  auto mv_P = [&](const lds_d *v, double (&o)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      double s = 0.0;
      mvec<MU>(P + (var[r] ? vj_[r] : 0), (size_t)n, v, v, n, [&](int, double p, double x0, double) { s = fma(p, x0, s); });
      o[r] = var[r] ? s : 0.0;
    }
  };
  auto nrm = [&](const double (&v)[R], const bool (&on)[R]) {
    double a = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) a = fmax(a, on[r] ? fabs(v[r]) : 0.0);
    return wave_max(a);
  };
  auto nrm_lds = [&](const lds_d *v, const int len) {
    double a = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int e = lane + kWave * r;
      if (e < len) a = fmax(a, fabs(v[e]));
    }
    return wave_max(a);
  };
  double Ax[R], Adx[R];
  mv_A2(xus, Ax, dxus, Adx);  // (A dx is needed by every check that does not end in Optimal / PrimalInfeasible: one pass over A for both)
  // OPTIMALITY :584-594
  {
    const double Ax_norm = nrm(Ax, con);
    double rs[R];
#pragma unroll
    for (int r = 0; r < R; ++r) rs[r] = con[r] ? Ax[r] - zus[ci_[r]] : 0.0;
    if (nrm(rs, con) <= eps_abs + eps_rel * fmax(Ax_norm, nrm_lds(zus, m))) {
      double Px[R], Aty[R], qv[R], res[R];
      mv_P(xus, Px);
      mv_At(yus, Aty);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        qv[r]  = var[r] ? q[vj_[r]] : 0.0;
        res[r] = var[r] ? Px[r] + (qv[r] + Aty[r]) : 0.0;  // :592
      }
      const double dual_scale = fmax(fmax(nrm(Px, var), nrm(qv, var)), nrm(Aty, var));
      if (nrm(res, var) <= eps_abs + eps_rel * dual_scale) return SFB_QP_OPTIMAL;
    }
  }
  // PRIMAL INFEASIBILITY :598-621
  // (the certificate's cheap parts first, the mat-vec A' dy only while the verdict is still open: below)
  {
    const double Edy = nrm_lds(dyus, m);
    const double thr = eps_pinf * Edy;
    // the ordered sum with its early exit to +inf: the oracle breaks at the first row with an unbounded side beyond the
    // threshold and the sum becomes +inf, so only "any such row" matters; otherwise the sum is the ordered one (a skipped
    // term adds +0.0, exact: the sum is never -0.0).  The terms go through y / z (dead after the optimality test).
    lds_d *const tu = yus, *const tl = zus;
    bool brk = false;
    double ul_[R][2];
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (con[r]) {
        const double ui = u[ci_[r]], li = l[ci_[r]], dyi = dyus[ci_[r]];
        ul_[r][0] = (ui != inf) ? ui * fmax(0.0, dyi) : 0.0;
        ul_[r][1] = (li != -inf) ? li * fmin(0.0, dyi) : 0.0;
        brk = brk || (ui == inf && dyi > thr) || (li == -inf && dyi < -thr);
      }
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (con[r]) {
        tu[ci_[r]] = ul_[r][0];
        tl[ci_[r]] = ul_[r][1];
      }
    wave_lds_fence();
    double s = 0.0;
    {
      int i = 0;
      for (; i + 8 <= m; i += 8) {  // (eight terms of each side requested together; same order of additions)
        double a[8], b[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[e] = tu[i + e]; b[e] = tl[i + e]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) { s += a[e]; s += b[e]; }
      }
      for (; i < m; ++i) { s += tu[i]; s += tl[i]; }
    }
    if (wave_ballot(brk)) s = inf;
    // :616-620 is max(||A' dy||, s) < thr with max(a, s) = (a < s) ? s : a.  With s >= thr (a true comparison: s is no NaN) the
    // maximum is s or an a >= s or a NaN, never below thr -- whatever A' dy is, so it is not formed.
    if (!(s >= thr)) {
      double Aty[R];
      mv_At(dyus, Aty);
      const double an = nrm(Aty, var);
      if (((an < s) ? s : an) < thr) return SFB_QP_PRIMAL_INFEASIBLE;
    }
  }
  // DUAL INFEASIBILITY :625-641
  // (q' dx and the row conditions on A dx first; P dx only when they hold)
  {
    const double dxn = nrm_lds(dxus, n);
    const double thr = eps_dinf * dxn;
    lds_d *const qs = xus;  // q next to dx for the ordered dot product
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (var[r]) qs[vj_[r]] = q[vj_[r]];
    wave_lds_fence();
    double qdx = 0.0;
    {
      int j = 0;
      for (; j + 8 <= n; j += 8) {
        double a[8], b[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[e] = qs[j + e]; b[e] = dxus[j + e]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) qdx = fma(a[e], b[e], qdx);
      }
      for (; j < n; ++j) qdx = fma(qs[j], dxus[j], qdx);
    }
    bool rowok = true;
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (con[r]) {
        const double ui = u[ci_[r]], li = l[ci_[r]];
        if (ui == inf) rowok = rowok && (Adx[r] >= -thr);
        else if (li == -inf) rowok = rowok && (Adx[r] <= thr);
        else rowok = rowok && (fabs(Adx[r]) < thr);
      }
    if ((qdx <= thr) && !wave_ballot(!rowok)) {
      double Pdx[R];
      mv_P(dxus, Pdx);
      if (nrm(Pdx, var) <= thr) return SFB_QP_DUAL_INFEASIBLE;
    }
  }
  return -1;
}

// One row of the reference's verbose table (qp_solver.hpp:490-501) at a stopping check: ITER, OBJ = (0.5 P x + q) . x,
// PRI_RES = |A x - z|_inf, DUA_RES = |P x + q + A' y|_inf on the un-scaled iterate (V = [x | dx | y | z], still intact:
// called in FRONT of the check, which re-uses y, z and x as scratch), TIME in microseconds of the device clock since the
// solve began.  The three products row by row like the check's (k ascending fma chains), the dot product a sequential
// mul + add chain: the columns of oracle/qp_oracle.c's trace, bit for bit.  Only the TRACE instance calls this.
// sc: n doubles of LDS scratch behind the kernel's regular areas.
template<int R>
__device__ __attribute__((noinline)) void mid_trace_row(const int n_, const int m_, const int lane, const double *const P, const double *const q,
                                                       const double *const A, lds_d *const V, lds_d *const sc, const uint32_t iter,
                                                       const unsigned long long t0_ticks, double *const row)
{
  const int n = muni(n_), m = muni(m_);
  lds_d *const xus = V, *const yus = V + 2 * n, *const zus = yus + m;
  double Px[R], Aty[R], Ax[R], qv[R];
  double pri = 0.0, dua = 0.0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int e = lane + kWave * r;
    const bool var = e < n, con = e < m;
    double sp = 0.0, sa = 0.0, st = 0.0;
    mvec<8>(P + (var ? e : 0), (size_t)n, xus, xus, n, [&](int, double p, double x0, double) { sp = fma(p, x0, sp); });
    mvec<8>(A + (size_t)(var ? e : 0) * m, 1, yus, yus, m, [&](int, double a, double y0, double) { st = fma(a, y0, st); });
    mvec<8>(A + (con ? e : 0), (size_t)m, xus, xus, n, [&](int, double a, double x0, double) { sa = fma(a, x0, sa); });
    Px[r] = var ? sp : 0.0;  Aty[r] = var ? st : 0.0;  Ax[r] = con ? sa : 0.0;
    qv[r] = var ? q[e] : 0.0;
    if (var) {
      sc[e] = (0.5 * Px[r] + qv[r]) * xus[e];
      dua   = fmax(dua, fabs(Px[r] + qv[r] + Aty[r]));
    }
    if (con) pri = fmax(pri, fabs(Ax[r] - zus[e]));
  }
  wave_lds_fence();
  double o = 0.0;
  for (int j = 0; j < n; ++j) o += sc[j];
  pri = wave_max(pri);
  dua = wave_max(dua);
  if (lane == 0) {
    row[0] = (double)iter;  row[1] = o;  row[2] = pri;  row[3] = dua;
    row[4] = (double)((wall_clock64() - t0_ticks) / 100ull);
  }
  wave_lds_fence();
}

// outlined instance (the LDS block engine: the check's registers stay out of the ADMM loop's budget)
template<int R>
__device__ __attribute__((noinline)) int mid_stop_check(const int n, const int m, const int lane, const double *const P, const double *const q,
                                                        const double *const A, const double *const l, const double *const u, lds_d *const V,
                                                        lds_d *const tmp, const double eps_abs, const double eps_rel, const double eps_pinf,
                                                        const double eps_dinf)
{
  return mid_stop_check_body<R>(n, m, lane, P, q, A, l, u, V, tmp, eps_abs, eps_rel, eps_pinf, eps_dinf);
}

// ---------------------------------------------------------------------------------------------------------------
// One QP as seen by the phases below: its LDS regions, its arrays in global memory, where its results go.
struct MidC {
  int n, m, k, lane, tsz;
  lds_d *T, *Dg, *tmp, *V;
  lds_b *perm;
  lds_i *iperm;
  const double *P, *q, *A, *l, *u, *wx, *wy;
  double *ox, *oy, *oobj;
  uint32_t *oiter;
  int32_t *ocode;
  double *stamps;  // TRACE instance: ten wall-clock stamps of this QP (see MP_T), else nullptr
};
__device__ __forceinline__ MidC mid_context(double *sm, const DenseKernelParams &kp, const QpBatch &g, const size_t b, const int lane)
{
  const int n = kp.n, m = kp.m, k = n + m;
  const MidLayout L = mid_layout(n, m);
  MidC C;
  C.n = n;  C.m = m;  C.k = k;  C.lane = lane;  C.tsz = (k * (k + 1)) / 2 + kMidPadT;
  C.T = (lds_d *)(sm + L.T);  C.Dg = (lds_d *)(sm + L.Dg);  C.tmp = (lds_d *)(sm + L.tmp);  C.V = (lds_d *)(sm + L.V);
  C.perm  = (lds_b *)(sm + L.perm);
  C.iperm = (lds_i *)(sm + L.tmp);
  C.P = g.P + b * (size_t)n * n;  C.q = g.q + b * (size_t)n;  C.A = g.A + b * (size_t)m * n;
  C.l = g.l + b * (size_t)m;      C.u = g.u + b * (size_t)m;
  C.wx = g.wx ? g.wx + b * (size_t)n : nullptr;
  C.wy = g.wy ? g.wy + b * (size_t)m : nullptr;
  C.ox = g.x + b * (size_t)n;  C.oy = g.y + b * (size_t)m;
  C.oobj  = g.obj ? g.obj + b : nullptr;
  C.oiter = g.iter ? g.iter + b : nullptr;
  C.ocode = g.code + b;
  C.stamps = nullptr;
  return C;
}

// Setup of QPSolver::solve (:347-445): scaling, pre-check and rho, KKT matrix in pivot order, LDL', the lanes' rows of the
// permuted system with their constants and the initial iterate.  Returns the status if the solve ends before the first
// iteration (PrimalInfeasible / Unknown), else -1.
template<int R, int kFillU>
__device__ __forceinline__ int mid_setup(const MidC &C, const DenseKernelParams &kp, MidRow (&h)[R], double &c, unsigned long long &t0_ticks)
{
  const int n = C.n, m = C.m, k = C.k, lane = C.lane, tsz = C.tsz;
  lds_d *const T = C.T, *const Dg = C.Dg, *const tmp = C.tmp, *const V = C.V;
  lds_b *const perm = C.perm;
  lds_i *const iperm = C.iperm;
  lds_d *const SX = V, *const SY = V + n;
  const double *const P = C.P, *const q = C.q, *const A = C.A, *const l = C.l, *const u = C.u;
  const double inf = INFINITY;
  (void)tsz; (void)T; (void)Dg; (void)tmp; (void)perm; (void)iperm; (void)SX; (void)SY; (void)P; (void)q; (void)A; (void)l; (void)u; (void)inf; (void)k; (void)m; (void)lane;
  int ret_code = -1;
  c            = 1.0;
  // ================= setup: natural order, unit e = lane + 64 r: e < n variable e, else constraint e - n =================
#pragma unroll
  for (int r = 0; r < R; ++r) {  // analyze(): :306-308
    const int e = lane + kWave * r;
    if (e < k) V[e] = 1.0;
  }
  wave_lds_fence();
  if (kp.scaling) {  // :673-730
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int e = lane + kWave * r;
      if (e < n) {  // :681-690 column inf-norms of P
        double t = 0.0;
        mrun(P + (size_t)e * n, 1, n, [&](int, double p) { t = fmax(t, fabs(p)); });
        if (t == 0.0) t = 1.0;
        tmp[e] = t;
      }
    }
    wave_lds_fence();
    double sum = tmp[0];  // :693 mean(): sequential sum
    for (int j = 1; j < n; ++j) sum += tmp[j];
    double qv = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int e = lane + kWave * r;
      if (e < n) qv = fmax(qv, fabs(q[e]));
    }
    const double qn = wave_max(qv);
    c               = 1.0 / fmax(fmax(1e-6, sum / (double)n), qn);
    wave_lds_fence();
    int pass = 0;
    double crit;
    do {  // :698-729
      double inc[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int e = lane + kWave * r;
        double v    = 0.0;
        if (e < n) {
          const double sxc = SX[e];
          mrun(P + (size_t)e * n, 1, n, [&](int row, double p) { v = fmax(v, fabs(c * SX[row] * sxc * p)); });  // :704-707
          mrun(A + (size_t)e * m, 1, m, [&](int row, double a) { v = fmax(v, fabs(SY[row] * sxc * a)); });      // :712-714
        } else if (e < k) {
          const double syr = SY[e - n];
          mrun(A + (e - n), m, n, [&](int col, double a) { v = fmax(v, fabs(syr * SX[col] * a)); });
        }
        if (v == 0.0) v = 1.0;
        inc[r] = v;
      }
      wave_lds_fence();  // every lane has read the old sx / sy
      double cm = 0.0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int e = lane + kWave * r;
        if (e < k) {
          V[e] = sqrt(1.0 / fmax(inc[r], 1e-8)) * V[e];
          cm   = fmax(cm, fabs(inc[r] - 1.0));
        }
      }
      crit = wave_max(cm);
      wave_lds_fence();
    } while (pass++ < 10 && crit > 0.1);
  }

  MP_T(1);
  // ---- pre-check and rho :361-374, the diagonal of the KKT matrix :399-404 ----
  double dgn[R];
  int idn[R];
  {
    bool bad = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int e = lane + kWave * r;
      idn[r]      = e;
      dgn[r]      = 0.0;
      if (e < n) {
        const double sxe = SX[e];
        dgn[r]           = c * sxe * P[(size_t)e * n + e] * sxe + kp.sigma;
      } else if (e < k) {
        const double li = l[e - n], ui = u[e - n];
        bad = bad || (li == inf) || (ui == -inf) || (ui - li < 0.0);
        double rho;
        if (li == -inf && ui == inf) rho = 1e-6;
        else if (SY[e - n] * fabs(li - ui) < 1e-5) rho = 1e3 * kp.rho_bar;
        else rho = kp.rho_bar;
        dgn[r] = 1.0 / (-rho);
      }
    }
    if (wave_ballot(bad)) ret_code = SFB_QP_PRIMAL_INFEASIBLE;
  }
  t0_ticks = wall_clock64();  // :376

  // ---- Eigen's pivot order (from the diagonal), then the KKT matrix straight into its final positions ----
  MP_T(2);
  mid_pivot_order<R>(k, dgn, idn, lane);
  MP_T(3);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int pos = lane + kWave * r;
    if (pos < k) {
      perm[pos]      = (unsigned char)idn[r];
      iperm[idn[r]] = pos;
    }
  }
  for (int e = lane; e < tsz; e += kWave) T[e] = 0.0;
  wave_lds_fence();
  bool fin = true;
  {
    const float rn = 1.0f / (float)n, rm = 1.0f / (float)m;
    mstream<kFillU>(P, n * n, lane, [&](const int e, const double pv) {  // upper entries (a, bb), a < bb, of P: ((c sx_a) P_ab) sx_b
      const int bb = (int)(((float)e + 0.5f) * rn), a = e - bb * n;
      if (a < bb) {
        const double v = c * SX[a] * pv * SX[bb];
        const int ra = iperm[a], rb = iperm[bb];
        T[mtri(ra > rb ? ra : rb) + (ra > rb ? rb : ra)] = v;
        fin = fin && mfinite(v);
      }
    });
    mstream<kFillU>(A, m * n, lane, [&](const int e, const double av) {  // (sy_i A_ij) sx_j
      const int j = (int)(((float)e + 0.5f) * rm), i = e - j * m;
      const double v = SY[i] * av * SX[j];
      const int ra = iperm[n + i], rb = iperm[j];
      T[mtri(ra > rb ? ra : rb) + (ra > rb ? rb : ra)] = v;
      fin = fin && mfinite(v);
    });
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int pos = lane + kWave * r;
      if (pos < k) {
        T[mtri(pos) + pos] = dgn[r];
        fin = fin && mfinite(dgn[r]);
      }
    }
    fin = !wave_ballot(!fin);
  }
  wave_lds_fence();

  MP_T(4);
  // ---- LDL' :428-433 ----
  if (!mid_ldlt<R>(k, T, Dg, tmp, lane, fin)) ret_code = SFB_QP_UNKNOWN;
  MP_T(5);

  // ================= lane roles for the loop: rows lane, lane + 64 of the PERMUTED system =================
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = lane + kWave * r;
    const bool in = row < k;
    const int v   = in ? perm[row] : 0;
    MidRow &w = h[r];
    w.isx = in && v < n;
    w.isc = in && v >= n;
    w.xi  = w.isx ? v : 0;
    w.ci  = w.isc ? v - n : 0;
    w.sc  = w.isx ? SX[w.xi] : (w.isc ? SY[w.ci] : 1.0);
    w.qc  = w.isx ? c * w.sc * q[w.xi] : 0.0;  // (c sx_j) q_j of the right-hand side :450
    const double li = w.isc ? l[w.ci] : 0.0, ui = w.isc ? u[w.ci] : 0.0;
    double rho = 1.0;
    if (w.isc) {
      if (li == -inf && ui == inf) rho = 1e-6;
      else if (w.sc * fabs(li - ui) < 1e-5) rho = 1e3 * kp.rho_bar;
      else rho = kp.rho_bar;
    }
    w.rho  = rho;
    w.rinv = 1.0 / rho;
    w.lo   = w.isc ? w.sc * li : 0.0;
    w.hi   = w.isc ? w.sc * ui : 0.0;
    w.x = w.y = w.z = 0.0;
  }
  // ---- initial iterate :436-445 ----
  if (C.wx != nullptr) {
    const double *const wx = C.wx, *const wy = C.wy;
    lds_d *const WX = tmp;  // the warm primal, broadcast operand of z = (Sy A) x_ws
    for (int j = lane; j < n; j += kWave) WX[j] = wx[j];
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < R; ++r) {
      MidRow &w = h[r];
      if (w.isx) w.x = (1.0 / w.sc) * wx[w.xi];
      if (w.isc) {
        w.y      = c * ((1.0 / w.sc) * wy[w.ci]);
        double s = 0.0;
        const double syv = w.sc;
        mrun(A + w.ci, m, n, [&](int j, double a) { s = fma(syv * a, WX[j], s); });
        w.z = s;
      }
    }
    wave_lds_fence();
  }
  return ret_code;
}

// The record of a QP in the workspace (doubles): [T (tsz) | D (k) | per row of the permuted system: scale factor, x or y, z
// (3 k) | permutation (k bytes) | header: c, iter, next check, start ticks, status].  A QP travels through it between the
// setup, loop and finish launches and when it is suspended by a time-sliced loop launch.
constexpr int kMidHdr = 8;
__host__ __device__ inline size_t mid_save_doubles(const int n, const int m)
{
  const size_t k = (size_t)n + m;
  return (k * (k + 1)) / 2 + kMidPadT + k + 3 * k + (k + 7) / 8 + kMidHdr;
}
struct MidRec {
  double *T, *Dg, *rows, *perm, *hdr;
};
__device__ __forceinline__ MidRec mid_record(double *const sv, const int k, const int tsz)
{
  MidRec r;
  r.T = sv;  r.Dg = sv + tsz;  r.rows = r.Dg + k;  r.perm = r.rows + 3 * k;  r.hdr = r.perm + (k + 7) / 8;
  return r;
}
template<int R>
__device__ __forceinline__ void mid_save_state(const MidC &C, const MidRec rec, const MidRow (&h)[R], const double c, const uint32_t iter,
                                               const uint32_t next_chk, const unsigned long long t0_ticks, const int code, const bool with_factor)
{
  const int k = C.k, lane = C.lane, tsz = C.tsz;
  wave_lds_fence();
  if (with_factor) {
    for (int e = lane; e < tsz; e += kWave) rec.T[e] = C.T[e];
    for (int e = lane; e < k; e += kWave) rec.Dg[e] = C.Dg[e];
    unsigned char *const pb = reinterpret_cast<unsigned char *>(rec.perm);
    for (int e = lane; e < k; e += kWave) pb[e] = C.perm[e];
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = lane + kWave * r;
    if (row < k) {
      const MidRow &w = h[r];
      rec.rows[3 * row]     = w.sc;
      rec.rows[3 * row + 1] = w.isx ? w.x : w.y;
      rec.rows[3 * row + 2] = w.z;
    }
  }
  if (lane == 0) {
    rec.hdr[0] = c;
    rec.hdr[1] = (double)iter;
    rec.hdr[2] = (double)next_chk;
    rec.hdr[3] = __longlong_as_double((long long)t0_ticks);
    rec.hdr[4] = (double)code;
  }
}
// with_factor: the factor and D come back to LDS (loop launch); otherwise only what the finish phase reads
template<int R>
__device__ __forceinline__ void mid_load_state(const MidC &C, const DenseKernelParams &kp, const MidRec rec, MidRow (&h)[R], double &c, uint32_t &iter,
                                               uint32_t &next_chk, unsigned long long &t0_ticks, int &code, const bool with_factor)
{
  const int n = C.n, m = C.m, k = C.k, lane = C.lane, tsz = C.tsz;
  lds_d *const T = C.T, *const Dg = C.Dg, *const tmp = C.tmp, *const V = C.V;
  lds_b *const perm = C.perm;
  lds_i *const iperm = C.iperm;
  lds_d *const SX = V, *const SY = V + n;
  const double *const P = C.P, *const q = C.q, *const A = C.A, *const l = C.l, *const u = C.u;
  const double inf = INFINITY;
  (void)tsz; (void)T; (void)Dg; (void)tmp; (void)perm; (void)iperm; (void)SX; (void)SY; (void)P; (void)q; (void)A; (void)l; (void)u; (void)inf; (void)k; (void)m; (void)lane;
  const double *const sv_rows = rec.rows, *const sv_hdr = rec.hdr;
  if (with_factor) {
    for (int e = lane; e < tsz; e += kWave) T[e] = rec.T[e];
    for (int e = lane; e < k; e += kWave) Dg[e] = rec.Dg[e];
  }
  {
    const unsigned char *const pb = reinterpret_cast<const unsigned char *>(rec.perm);
    for (int e = lane; e < k; e += kWave) perm[e] = pb[e];
  }
  // (the header is the same for every lane: scalars -- as vector values they were live in VGPRs across the whole ADMM loop)
  c        = muni_d(sv_hdr[0]);
  iter     = (uint32_t)muni((int)(uint32_t)sv_hdr[1]);
  next_chk = (uint32_t)muni((int)(uint32_t)sv_hdr[2]);
  t0_ticks = (unsigned long long)__double_as_longlong(muni_d(sv_hdr[3]));
  code     = muni((int)sv_hdr[4]);
  wave_lds_fence();
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = lane + kWave * r;
    const bool in = row < k;
    const int v   = in ? perm[row] : 0;
    MidRow &w = h[r];
    w.isx = in && v < n;
    w.isc = in && v >= n;
    w.xi  = w.isx ? v : 0;
    w.ci  = w.isc ? v - n : 0;
    const double sc = in ? sv_rows[3 * row] : 1.0, a0 = in ? sv_rows[3 * row + 1] : 0.0, a1 = in ? sv_rows[3 * row + 2] : 0.0;
    double rho_bar = kp.rho_bar, rho_free = 1e-6;  // (opaque: 1e3 rho_bar and the literal are made here, per QP, not once per kernel and then held in scratch memory)
    asm volatile("" : "+s"(rho_bar), "+s"(rho_free));
    w.sc  = sc;
    w.qc  = w.isx ? c * w.sc * q[w.xi] : 0.0;
    const double li = w.isc ? l[w.ci] : 0.0, ui = w.isc ? u[w.ci] : 0.0;
    double rho = 1.0;
    if (w.isc) {
      if (li == -inf && ui == inf) rho = rho_free;
      else if (w.sc * fabs(li - ui) < 1e-5) rho = 1e3 * rho_bar;
      else rho = rho_bar;
    }
    w.rho  = rho;
    w.rinv = 1.0 / rho;
    w.lo   = w.isc ? w.sc * li : 0.0;
    w.hi   = w.isc ? w.sc * ui : 0.0;
    w.x = w.isx ? a0 : 0.0;
    w.y = w.isc ? a0 : 0.0;
    w.z = w.isc ? a1 : 0.0;
  }
}

// The end of QPSolver::solve (:515-548): polish, un-scale, report.
template<int R, int NB, int kFillU>
__device__ __forceinline__ void mid_finish(const MidC &C, const DenseKernelParams &kp, const MidRow (&h)[R], const double c, const int ret_code,
                                           const uint32_t iter)
{
  const int n = C.n, m = C.m, k = C.k, lane = C.lane, tsz = C.tsz;
  lds_d *const T = C.T, *const Dg = C.Dg, *const tmp = C.tmp, *const V = C.V;
  lds_b *const perm = C.perm;
  lds_i *const iperm = C.iperm;
  lds_d *const SX = V, *const SY = V + n;
  const double *const P = C.P, *const q = C.q, *const A = C.A, *const l = C.l, *const u = C.u;
  const double inf = INFINITY;
  (void)tsz; (void)T; (void)Dg; (void)tmp; (void)perm; (void)iperm; (void)SX; (void)SY; (void)P; (void)q; (void)A; (void)l; (void)u; (void)inf; (void)k; (void)m; (void)lane;
  // ================= the end of solve() :515-548: V = [sx | sy | x | y | lists] in natural order =================
  lds_d *const XS = V + k, *const YS = V + k + n;
  wave_lds_fence();
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const MidRow &w = h[r];
    if (w.isx) { SX[w.xi] = w.sc; XS[w.xi] = w.x; }
    if (w.isc) { SY[w.ci] = w.sc; YS[w.ci] = w.y; }
  }
  wave_lds_fence();

  // ---- polish :92-204 (on the scaled iterate; a failed factorisation leaves the ADMM solution) ----
  if (ret_code == SFB_QP_OPTIMAL && kp.polish) {
    const double eps = DBL_EPSILON;
    lds_b *const colof = (lds_b *)(V + 2 * k);  // per constraint: its row n + a of the polish system, or 255
    lds_b *const LU    = colof + m;             // row n + a -> constraint
    // active sets: lower indices first, then upper, each ascending (:113-123)
    int nl = 0, nu = 0;
    int act[R], pos[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = lane + kWave * r;
      int a       = 0;
      if (i < m) {
        const double yi = YS[i];
        if (yi < -100 * eps && l[i] != -inf) a = 1;
        if (yi > 100 * eps && u[i] != inf) a = 2;
      }
      const unsigned long long bl = wave_ballot(a == 1), bu = wave_ballot(a == 2);
      act[r] = a;
      pos[r] = (a == 1) ? nl + __popcll(bl & lanemask_lt(lane)) : nu + __popcll(bu & lanemask_lt(lane));
      nl += __popcll(bl);
      nu += __popcll(bu);
    }
    const int na = nl + nu, K = n + na;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = lane + kWave * r;
      if (i < m) {
        const int a = act[r] == 0 ? -1 : pos[r] + (act[r] == 2 ? nl : 0);
        colof[i]    = (unsigned char)(a < 0 ? 255 : n + a);
        if (a >= 0) LU[a] = (unsigned char)i;
      }
    }
    wave_lds_fence();
    // rows of the polish system: e = lane + 64 r < K: variable e, or active constraint LU[e - n]; h :179-182; diagonal of Hp :174-177
    double hh[R], dgp[R];
    int idp[R], prow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int e = lane + kWave * r;
      idp[r] = e;  hh[r] = 0.0;  dgp[r] = 0.0;  prow[r] = 0;
      if (e < n) {
        const double sxe = SX[e];
        dgp[r] = c * sxe * P[(size_t)e * n + e] * sxe + kp.delta;
        hh[r]  = -c * (sxe * q[e]);
      } else if (e < K) {
        const int row = LU[e - n];
        prow[r]       = row;
        dgp[r]        = 0.0 - kp.delta;
        hh[r]         = (e - n < nl) ? SY[row] * l[row] : SY[row] * u[row];
      }
    }
    mid_pivot_order<R>(K, dgp, idp, lane);
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int p = lane + kWave * r;
      if (p < K) {
        perm[p]        = (unsigned char)idp[r];
        iperm[idp[r]] = p;
      }
    }
    const int tszp = (K * (K + 1)) / 2 + kMidPadT;
    for (int e = lane; e < tszp; e += kWave) T[e] = 0.0;
    wave_lds_fence();
    bool finp = true;
    {
      const float rn = 1.0f / (float)n, rm = 1.0f / (float)m;
      mstream<kFillU>(P, n * n, lane, [&](const int e, const double pv) {
        const int bb = (int)(((float)e + 0.5f) * rn), a = e - bb * n;
        if (a < bb) {
          const double v = c * SX[a] * pv * SX[bb];  // :161
          const int ra = iperm[a], rb = iperm[bb];
          T[mtri(ra > rb ? ra : rb) + (ra > rb ? rb : ra)] = v;
          finp = finp && mfinite(v);
        }
      });
      mstream<kFillU>(A, m * n, lane, [&](const int e, const double av) {
        const int j = (int)(((float)e + 0.5f) * rm), i = e - j * m;
        const int col = colof[i];
        if (col != 255) {
          const double v = SY[i] * av * SX[j];  // :163
          const int ra = iperm[col], rb = iperm[j];
          T[mtri(ra > rb ? ra : rb) + (ra > rb ? rb : ra)] = v;
          finp = finp && mfinite(v);
        }
      });
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int p = lane + kWave * r;
        if (p < K) {
          T[mtri(p) + p] = dgp[r];
          finp = finp && mfinite(dgp[r]);
        }
      }
      finp = !wave_ballot(!finp);
    }
    wave_lds_fence();
    if (mid_ldlt<R>(K, T, Dg, tmp, lane, finp)) {  // :187-190
      double tt[R];
#pragma unroll
      for (int r = 0; r < R; ++r) tt[r] = 0.0;
      lds_d *const tv = tmp;
      for (uint32_t it = 0; it != kp.polish_iter; ++it) {  // :193-195  t += Hp^-1 (h - H t); the entries of H are recomputed (same products)
        bool tfin = true;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int e = lane + kWave * r;
          if (e < K) tv[e] = tt[r];
          tfin = tfin && mfinite(tt[r]);
        }
        tfin = !wave_ballot(!tfin);
        wave_lds_fence();
        double res[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int e = lane + kWave * r;
          double acc  = 0.0;
          if (e < n) {
            const double sxr = SX[e];
            // upper entry (a, bb) of P: (j, e) for j < e -- column e, contiguous -- then (e, j) for j >= e -- row e, stride n
            mrun(P + (size_t)e * n, 1, e, [&](int j, double p) { acc = fma(c * SX[j] * p * sxr, tv[j], acc); });
            mrun(P + e + (size_t)e * n, n, n - e, [&](int d, double p) { acc = fma(c * sxr * p * SX[e + d], tv[e + d], acc); });
            for (int a0 = 0; a0 < na; a0 += 8) {  // the active rows' entries of column e, eight at a time
              int row[8];
              double av[8];
#pragma unroll
              for (int uu = 0; uu < 8; ++uu) row[uu] = (a0 + uu < na) ? LU[a0 + uu] : 0;
#pragma unroll
              for (int uu = 0; uu < 8; ++uu) av[uu] = A[row[uu] + (size_t)e * m];
#pragma unroll
              for (int uu = 0; uu < 8; ++uu)
                if (a0 + uu < na) acc = fma(SY[row[uu]] * av[uu] * sxr, tv[n + a0 + uu], acc);
            }
          } else if (e < K) {
            const double syr = SY[prow[r]];
            mrun(A + prow[r], m, n, [&](int j, double av) { acc = fma(syr * av * SX[j], tv[j], acc); });
            // the zero (2,2) block of H: fma(0, t_j, acc) leaves acc unchanged for finite t_j (acc is never -0)
            if (!tfin)
              for (int j = n; j < K; ++j) acc = fma(0.0, tv[j], acc);
          }
          res[r] = hh[r] - acc;
        }
        wave_lds_fence();
        // oracle_ldlt_solve: P b -> sweeps -> P^T, through the exchange vector
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int e = lane + kWave * r;
          if (e < K) tv[e] = res[r];
        }
        wave_lds_fence();
        rows::Pair pr{0.0, 0.0};
        {
          const int p0 = lane, p1 = lane + kWave;
          if (p0 < K) pr.lo = tv[perm[p0]];
          if (R > 1 && p1 < K) pr.hi = tv[perm[p1]];
        }
        wave_lds_fence();
        pr = rows::row_sweeps<NB>(K, (const double *)T, (const double *)Dg, pr, lane);
        {
          const int p0 = lane, p1 = lane + kWave;
          if (p0 < K) tv[perm[p0]] = pr.lo;
          if (R > 1 && p1 < K) tv[perm[p1]] = pr.hi;
        }
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int e = lane + kWave * r;
          if (e < K) tt[r] += tv[e];
        }
        wave_lds_fence();
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {  // :199-201
        const int e = lane + kWave * r;
        if (e < n) XS[e] = tt[r];
        else if (e < K) YS[prow[r]] = tt[r];
      }
    }
    wave_lds_fence();
  }

  MP_T(8);
  // ---- un-scale and report :544-548 ----
  double *const ox = C.ox, *const oy = C.oy;
  lds_d *const xo = tmp, *const pv = Dg;  // (both dead by now)
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int e = lane + kWave * r;
    if (e < n) {
      const double v = SX[e] * XS[e];
      ox[e] = v;
      xo[e] = v;
    }
    if (e < m) oy[e] = SY[e] * YS[e] / c;
  }
  wave_lds_fence();
  if (C.oobj != nullptr) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int e = lane + kWave * r;
      if (e < n) {
        double s = 0.0;
        mrun(P + e, n, n, [&](int j, double p) { s = fma(0.5 * p, xo[j], s); });
        pv[e] = s + q[e];
      }
    }
    wave_lds_fence();
    if (lane == 0) {
      double o = 0.0;
      for (int i = 0; i < n; ++i) o = fma(xo[i], pv[i], o);
      *C.oobj = o;
    }
  }
  if (lane == 0) {
    *C.ocode = (ret_code >= 0) ? ret_code : SFB_QP_MAX_ITERATIONS;
    if (C.oiter != nullptr) *C.oiter = iter;
  }
}

// One ADMM iteration state machine (:447-510) on the rows in registers.  QUEUE: time-sliced -- returns true when the QP
// has used its slice while others are waiting (the caller suspends it), false when the loop has ended.
template<int NB, int R, bool QUEUE, bool TRACE = false>
__device__ __forceinline__ bool mid_admm(const MidC &C, const DenseKernelParams &kp, MidRow (&h)[R], const double c, uint32_t &iter, uint32_t &next_chk,
                                         int &ret_code, const unsigned long long t0_ticks, const uint32_t slice, const unsigned *const q_fresh,
                                         const unsigned *const q_rhead, const unsigned *const q_tail, const unsigned batch,
                                         double *const trace = nullptr, const int trace_cap = 0, lds_d *const trace_sc = nullptr)
{
  [[maybe_unused]] int trace_rows = 0;
  const int n = C.n, m = C.m, k = C.k, lane = C.lane, tsz = C.tsz;
  lds_d *const T = C.T, *const Dg = C.Dg, *const tmp = C.tmp, *const V = C.V;
  lds_b *const perm = C.perm;
  lds_i *const iperm = C.iperm;
  lds_d *const SX = V, *const SY = V + n;
  const double *const P = C.P, *const q = C.q, *const A = C.A, *const l = C.l, *const u = C.u;
  const double inf = INFINITY;
  (void)tsz; (void)T; (void)Dg; (void)tmp; (void)perm; (void)iperm; (void)SX; (void)SY; (void)P; (void)q; (void)A; (void)l; (void)u; (void)inf; (void)k; (void)m; (void)lane;
  const uint32_t sci = kp.stop_check_iter, maxit = kp.max_iter;
  lds_d *const xus = V, *const dxus = V + n, *const yus = V + 2 * n, *const zus = yus + m, *const dyus = tmp;  // the un-scaled iterates of a stopping check
  auto rhs = [&](const MidRow &w) { return w.isx ? (kp.sigma * w.x - w.qc) : (w.isc ? (w.z - w.rinv * w.y) : 0.0); };  // :450-451
  auto upd = [&](MidRow &w, const double t, const bool chk) {                                                              // :470-477
    const double xo = w.x, yo = w.y;
    w.x       = kp.alpha * t + kp.alpha_comp * w.x;
    double zn = kp.alpha * (w.rinv * t) + kp.alpha_comp * (w.rinv * w.y) + w.z;
    zn        = (zn < w.lo) ? w.lo : zn;
    zn        = (w.hi < zn) ? w.hi : zn;
    w.y       = kp.alpha_comp * w.y + kp.alpha * t + w.rho * w.z - w.rho * zn;
    w.z       = zn;
    if (chk) {  // :481-485
      if (w.isx) {
        xus[w.xi]  = w.sc * w.x;
        dxus[w.xi] = w.sc * (w.x - xo);
      }
      if (w.isc) {
        yus[w.ci]  = w.sc * w.y / c;
        zus[w.ci]  = (1.0 / w.sc) * w.z;
        dyus[w.ci] = w.sc * (w.y - yo) / c;
      }
    }
  };
  // k <= 64: the factor moves into registers for the duration of the loop (sweep_rows.h, registers-only engine)
  // (the TRACE instance serves every k <= 16 NB, also below the class of its block count: the LDS engine in its any-K form)
  constexpr bool kRegs = NB <= 4 && SFB_MID_REGS != 0 && !TRACE;
  constexpr bool kChkKeep = SFB_MID_CHK_KEEP != 0;
  rows::Masks masks{};
  if constexpr (!kRegs) masks = rows::make_masks();  // (once, in front of the loop: see sweep_rows.h)
  rows::RegFactor<kRegs ? NB : 2> F;
  if constexpr (kRegs) {
    wave_lds_fence();
    rows::reg_factor_load<NB>(F, k, (const double *)T, (const double *)Dg, lane);
  }
  uint32_t it0 = iter;  // iteration at which this wave took the QP
  for (; iter != maxit && ret_code < 0; ++iter) {
    if constexpr (QUEUE) {
      if (iter - it0 >= slice) {  // the slice is used up: hand the QP back if others are waiting
        unsigned waiting = 0;
        if (lane == 0) {
          const unsigned fr = __hip_atomic_load(q_fresh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned hd = __hip_atomic_load(q_rhead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned tl = __hip_atomic_load(q_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          waiting = (fr < batch || (int)(tl - hd) > 0) ? 1u : 0u;
        }
        if (muni((int)waiting)) return true;
        it0 = iter;  // nobody waits: a fresh slice
      }
    }
    rows::Pair t{rhs(h[0]), R > 1 ? rhs(h[R - 1]) : 0.0};
    if constexpr (kRegs) t.lo = rows::row_sweeps_reg<NB>(k, F, t.lo, lane);                              // :462
    else t = rows::row_sweeps_inl<NB, TRACE>(k, (const double *)T, (const double *)Dg, t, lane, masks);
    const bool chk = (iter == next_chk);                                                            // :465
    if (chk) next_chk += sci;
    upd(h[0], t.lo, chk);
    if constexpr (R > 1) upd(h[R - 1], t.hi, chk);
    if (chk) {
      wave_lds_fence();
      if constexpr (TRACE) {  // the reference's verbose table as data (:490-501)
        if (trace != nullptr && trace_rows < trace_cap) mid_trace_row<R>(n, m, lane, P, q, A, V, trace_sc, iter, t0_ticks, trace + 5 * (size_t)trace_rows);
        ++trace_rows;
      }
      // registers-only engine: the check is INLINED (an outlined call would save and restore the live factor registers
      // through scratch memory); SFB_MID_CHK_KEEP: the factor registers stay live across it and the check requests the
      // matrix entries in short batches that fit next to them, otherwise they are dead across it and re-filled from LDS behind it
      const int lane_c = mid_lane_now();  // (the same per check: its addresses are made inside it instead of living across the iterations)
      if constexpr (kRegs) ret_code = mid_stop_check_body<R, kChkKeep ? (NB <= 3 ? 4 : 8) : 16>(n, m, lane_c, P, q, A, l, u, V, tmp, kp.eps_abs, kp.eps_rel, kp.eps_pinf, kp.eps_dinf);
      else ret_code = mid_stop_check<R>(n, m, lane, P, q, A, l, u, V, tmp, kp.eps_abs, kp.eps_rel, kp.eps_pinf, kp.eps_dinf);
      if (ret_code < 0 && max_time_exceeded(kp.max_time_ns, t0_ticks)) ret_code = SFB_QP_MAX_TIME;  // :504-507
      wave_lds_fence();
      // the factor registers are re-filled from LDS behind the outlined check instead of living across the call (the
      // compiler would save and restore all of them through scratch memory at the call site)
      if constexpr (kRegs && !kChkKeep) rows::reg_factor_load<NB>(F, k, (const double *)T, (const double *)Dg, lane);
    }
  }
  return false;
}

// ---------------------------------------------------------------------------------------------------------------
// FUSED launch: one QP per workgroup from setup to report, the hardware dispatcher is the work queue, no workspace --
// batches the chip holds at once (and single QPs: one launch).
template<int NB, int WPE>
__global__ void __launch_bounds__(64, WPE) qp_dense_mid_kernel(const DenseKernelParams kp, const QpBatch g)
{
  constexpr int R = NB > 4 ? 2 : 1;
  constexpr int kFillU = NB > 4 ? 16 : 8;  // loads in flight per lane while P and A are streamed into the KKT matrix
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int lane = threadIdx.x;
  const MidC C   = mid_context(sm, kp, g, blockIdx.x, lane);
  MidRow h[R];
  double c = 1.0;
  unsigned long long t0_ticks = 0;
  MP_T(0);
  int ret_code      = mid_setup<R, kFillU>(C, kp, h, c, t0_ticks);
  uint32_t iter     = 0;
  uint32_t next_chk = (kp.stop_check_iter >= 2) ? 1u : 0xFFFFFFFFu;  // iter % sci == 1 (:465) without a division per iteration
  MP_T(6);
  mid_admm<NB, R, false>(C, kp, h, c, iter, next_chk, ret_code, t0_ticks, 0, nullptr, nullptr, nullptr, 0);
  MP_T(7);
  mid_finish<R, NB, kFillU>(C, kp, h, c, ret_code, iter);
#ifdef SFB_MID_PROF
  MP_T(9);
  if (lane == 0 && blockIdx.x == 0)
    printf("midprof (%d,%d) x10ns: scale %llu | rho+diag %llu | pivot order %llu | zero+fill %llu | ldlt %llu | roles+warm %llu | loop %llu (%u iterations) | polish %llu | report %llu\n",
           C.n, C.m, g_midprof[1] - g_midprof[0], g_midprof[2] - g_midprof[1], g_midprof[3] - g_midprof[2], g_midprof[4] - g_midprof[3],
           g_midprof[5] - g_midprof[4], g_midprof[6] - g_midprof[5], g_midprof[7] - g_midprof[6], iter, g_midprof[8] - g_midprof[7], g_midprof[9] - g_midprof[8]);
#endif
}

// TRACE instance of the fused launch (sfb_qp_dense_solve_batch_trace; verbose on one problem): the same solve, one block per QP
// whatever the batch size, with a row of the reference's verbose table written per stopping check (trace [batch][cap][5], rows
// beyond cap dropped).  Serves every n + m <= 16 NB (the NB = 3 instance also the sizes the four-per-wave kernel normally takes).
// phase_us (nullable, [batch][6] + 10 doubles of scratch per QP behind them, see qp_dense_mid_trace_launch): the per-phase
// microseconds of the reference's summary (:550-565) -- scaling and pre-check | matrix filling (pivot order, zero, fill) |
// factorization | iteration | polish | un-scale and report.
template<int NB, int WPE>
__global__ void __launch_bounds__(64, WPE) qp_dense_mid_trace_kernel(const DenseKernelParams kp, const QpBatch g, double *__restrict__ trace,
                                                                     const int trace_cap, double *__restrict__ phase_us)
{
  constexpr int R = NB > 4 ? 2 : 1;
  constexpr int kFillU = NB > 4 ? 16 : 8;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int lane = threadIdx.x;
  MidC C   = mid_context(sm, kp, g, blockIdx.x, lane);
  C.stamps = phase_us ? phase_us + (size_t)gridDim.x * 6 + (size_t)blockIdx.x * 10 : nullptr;
  MidRow h[R];
  double c = 1.0;
  unsigned long long t0_ticks = 0;
  MP_T(0);
  int ret_code      = mid_setup<R, kFillU>(C, kp, h, c, t0_ticks);
  uint32_t iter     = 0;
  uint32_t next_chk = (kp.stop_check_iter >= 2) ? 1u : 0xFFFFFFFFu;
  MP_T(6);
  mid_admm<NB, R, false, true>(C, kp, h, c, iter, next_chk, ret_code, t0_ticks, 0, nullptr, nullptr, nullptr, 0,
                               trace ? trace + (size_t)blockIdx.x * (size_t)trace_cap * 5 : nullptr, trace_cap, (lds_d *)(sm + mid_layout(kp.n, kp.m).total));
  MP_T(7);
  mid_finish<R, NB, kFillU>(C, kp, h, c, ret_code, iter);
  if (phase_us != nullptr) {
    __syncthreads();
    MP_T(9);
    if (lane == 0) {  // (lane 0 wrote the stamps itself)
      const double *st = C.stamps;
      double *o = phase_us + (size_t)blockIdx.x * 6;
      o[0] = (st[2] - st[0]) * 0.01;  // 100 MHz wall clock -> microseconds
      o[1] = (st[4] - st[2]) * 0.01;
      o[2] = (st[5] - st[4]) * 0.01;
      o[3] = (st[7] - st[5]) * 0.01;
      o[4] = (st[8] - st[7]) * 0.01;
      o[5] = (st[9] - st[8]) * 0.01;
    }
  }
}

// SPLIT launch for batches larger than the chip: setup (one QP per workgroup) -> records; loop (persistent, time-sliced);
// finish (one QP per workgroup).  The ADMM loop, where a batch spends its time, then runs in a kernel that carries
// nothing else: no scaling / fill / factorisation / polish code competing for its registers, and a QP that runs into
// max_iter does not hold a wave from setup to report.
template<int NB, int WPE>
__global__ void __launch_bounds__(64, WPE) qp_dense_mid_setup_kernel(const DenseKernelParams kp, const QpBatch g, double *__restrict__ gws, const size_t wsd)
{
  constexpr int R = NB > 4 ? 2 : 1;
  constexpr int kFillU = NB > 4 ? 16 : 8;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int lane = threadIdx.x;
  const MidC C   = mid_context(sm, kp, g, blockIdx.x, lane);
  MidRow h[R];
  double c = 1.0;
  unsigned long long t0_ticks = 0;
  const int ret_code = mid_setup<R, kFillU>(C, kp, h, c, t0_ticks);
  mid_save_state<R>(C, mid_record(gws + (size_t)blockIdx.x * wsd, C.k, C.tsz), h, c, 0u, (kp.stop_check_iter >= 2) ? 1u : 0xFFFFFFFFu, t0_ticks, ret_code, true);
}

template<int NB, int WPE>
__global__ void __launch_bounds__(64, WPE) qp_dense_mid_finish_kernel(const DenseKernelParams kp, const QpBatch g, double *__restrict__ gws, const size_t wsd)
{
  constexpr int R = NB > 4 ? 2 : 1;
  constexpr int kFillU = NB > 4 ? 16 : 8;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int lane = threadIdx.x;
  const MidC C   = mid_context(sm, kp, g, blockIdx.x, lane);
  MidRow h[R];
  double c = 1.0;
  unsigned long long t0_ticks = 0;
  uint32_t iter = 0, next_chk = 0;
  int ret_code = -1;
  mid_load_state<R>(C, kp, mid_record(gws + (size_t)blockIdx.x * wsd, C.k, C.tsz), h, c, iter, next_chk, t0_ticks, ret_code, false);
  mid_finish<R, NB, kFillU>(C, kp, h, c, ret_code, iter);
}

// The loop launch: a PERSISTENT grid with a device-side queue, time-sliced like the k <= 32 kernel's.  QPs are handed
// out by a ticket counter; a QP that has held its wave for `slice` iterations while others are waiting goes back into
// its record and to the back of a ring of 2 * batch tagged entries, from where any wave resumes it -- the QPs that run
// into max_iter advance together instead of the last-started one running alone at the end.  Results do not depend on
// grid, slice or order.
template<int NB, int WPE>
__global__ void __launch_bounds__(64, WPE) qp_dense_mid_loop_kernel(const DenseKernelParams kp, const QpBatch g, double *__restrict__ gws, const size_t wsd,
                                                                    unsigned *__restrict__ queue, const unsigned batch, const uint32_t slice)
{
  constexpr int R = NB > 4 ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  unsigned *const q_fresh = queue, *const q_rhead = queue + 16, *const q_tail = queue + 32, *const q_done = queue + 48;
  unsigned long long *const ring = reinterpret_cast<unsigned long long *>(queue + 64);
  const unsigned ring_n = 2u * batch;
  constexpr unsigned kNone = 0xFFFFFFFFu;
  unsigned pend   = kNone;  // ring ticket this wave is waiting for
  bool fresh_left = true;
  for (;;) {
    // (per QP: what depends on the lane alone -- index vectors and addresses of the record traffic, of the stopping check --
    // is otherwise hoisted out of this persistent loop and held in scratch memory across it)
    const int lane = mid_lane_now();
    int item = -1;
    if (pend == kNone) {
      if (fresh_left) {
        unsigned t = 0;
        if (lane == 0) t = atomicAdd(q_fresh, 1u);
        t = (unsigned)muni((int)t);
        if (t < batch) item = (int)t;
        else fresh_left = false;
      }
      if (item < 0) {
        unsigned t = 0;
        if (lane == 0) t = atomicAdd(q_rhead, 1u);
        pend = (unsigned)muni((int)t);
      }
    }
    if (item < 0) {
      unsigned long long e = 0;
      if (lane == 0) e = __hip_atomic_load(ring + (pend % ring_n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned lo32 = (unsigned)muni((int)(unsigned)e), hi32 = (unsigned)muni((int)(unsigned)(e >> 32));
      if (hi32 == pend + 1u) {  // my entry has arrived
        if (lane == 0) __hip_atomic_store(ring + (pend % ring_n), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        item = (int)(lo32 - 1u);
        pend = kNone;
        __threadfence();  // the record was written by another wave
      } else {
        unsigned d = 0;
        if (lane == 0) d = __hip_atomic_load(q_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)muni((int)d) >= batch) break;  // every QP has left the loop
        __builtin_amdgcn_s_sleep(16);
        continue;
      }
    }
    const MidC C     = mid_context(sm, kp, g, (size_t)item, lane);
    const MidRec rec = mid_record(gws + (size_t)item * wsd, C.k, C.tsz);
    MidRow h[R];
    double c = 1.0;
    unsigned long long t0_ticks = 0;
    uint32_t iter = 0, next_chk = 0;
    int ret_code = -1;
    mid_load_state<R>(C, kp, rec, h, c, iter, next_chk, t0_ticks, ret_code, true);
    const bool suspended = mid_admm<NB, R, true>(C, kp, h, c, iter, next_chk, ret_code, t0_ticks, slice, q_fresh, q_rhead, q_tail, batch);
    {  // (the record's addresses are made again behind the loop instead of living across it)
      MidC Cs = C;
      Cs.lane = mid_lane_now();
      mid_save_state<R>(Cs, rec, h, c, iter, next_chk, t0_ticks, ret_code, suspended);
    }
    const int lane_e = mid_lane_now();
    if (suspended) {
      __threadfence();  // the record is complete (device scope) before the id can be popped
      if (lane_e == 0) {
        const unsigned j           = atomicAdd(q_tail, 1u);
        unsigned long long *slot   = ring + (j % ring_n);
        const unsigned long long v = ((unsigned long long)(j + 1u) << 32) | ((unsigned)item + 1u);
        while (atomicCAS(slot, 0ull, v) != 0ull) __builtin_amdgcn_s_sleep(1);
      }
    } else if (lane_e == 0) {
      atomicAdd(q_done, 1u);
    }
  }
}

}  // namespace

size_t qp_dense_mid_lds_bytes(int n, int m) { return (size_t)mid_layout(n, m).total * sizeof(double); }

namespace {
// compile-time shape classes: NB = ceil(k / 16) blocks; waves per SIMD the LOOP kernels are compiled for (VGPR budget);
// the setup / finish kernels and the fused kernel take what the LDS leaves them anyway
template<int NBV> struct MidW { static constexpr int loop = NBV <= 3 ? SFB_MID_W3 : (NBV == 4 ? SFB_MID_W4 : (NBV == 5 ? 2 : 1)), other = NBV <= 5 ? 2 : 1; };

enum MidOp { MID_RESIDENT, MID_FUSED, MID_SETUP, MID_LOOP, MID_FINISH, MID_TRACE };
struct MidLaunch {
  MidOp op;
  unsigned grid, batch;
  double *ws;
  size_t wsd;
  unsigned *queue;
  uint32_t slice;
  int *resident;
  hipStream_t stream;
  double *trace = nullptr;
  int trace_cap = 0;
  double *phase_us = nullptr;
};
template<int NBV>
hipError_t mid_launch_nb(const DenseKernelParams &kp, const QpBatch &g, const size_t lds, const MidLaunch &a)
{
  constexpr int WL = MidW<NBV>::loop, WO = MidW<NBV>::other;
  switch (a.op) {
    case MID_RESIDENT: {  // waves of the loop kernel the device holds at once (LDS / VGPR limits)
      int per_cu = 0, dev = 0;
      hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, qp_dense_mid_loop_kernel<NBV, WL>, kWave, lds);
      if (e != hipSuccess) return e;
      hipDeviceProp_t prop;
      e = hipGetDevice(&dev);
      if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
      if (e != hipSuccess) return e;
      *a.resident = per_cu * prop.multiProcessorCount;
      return hipSuccess;
    }
    case MID_FUSED: hipLaunchKernelGGL((qp_dense_mid_kernel<NBV, WO>), dim3(a.grid), dim3(kWave), lds, a.stream, kp, g); break;
    case MID_SETUP: hipLaunchKernelGGL((qp_dense_mid_setup_kernel<NBV, WO>), dim3(a.grid), dim3(kWave), lds, a.stream, kp, g, a.ws, a.wsd); break;
    case MID_LOOP:
      hipLaunchKernelGGL((qp_dense_mid_loop_kernel<NBV, WL>), dim3(a.grid), dim3(kWave), lds, a.stream, kp, g, a.ws, a.wsd, a.queue, a.batch, a.slice);
      break;
    case MID_FINISH: hipLaunchKernelGGL((qp_dense_mid_finish_kernel<NBV, WO>), dim3(a.grid), dim3(kWave), lds, a.stream, kp, g, a.ws, a.wsd); break;
    case MID_TRACE:
      hipLaunchKernelGGL((qp_dense_mid_trace_kernel<NBV, WO>), dim3(a.grid), dim3(kWave), lds + (size_t)kp.n * sizeof(double), a.stream, kp, g, a.trace,
                         a.trace_cap, a.phase_us);
      break;
  }
  return hipGetLastError();
}
hipError_t mid_launch(const DenseKernelParams &kp, const QpBatch &g, const MidLaunch &a)
{
  const int k = kp.n + kp.m, nb = (k + 15) / 16;
  const size_t lds = qp_dense_mid_lds_bytes(kp.n, kp.m);
  switch (nb) {
    case 0: case 1: case 2: case 3: return mid_launch_nb<3>(kp, g, lds, a);
    case 4: return mid_launch_nb<4>(kp, g, lds, a);
    case 5: return mid_launch_nb<5>(kp, g, lds, a);
    case 6: return mid_launch_nb<6>(kp, g, lds, a);
    case 7: return mid_launch_nb<7>(kp, g, lds, a);
    default: return mid_launch_nb<8>(kp, g, lds, a);
  }
}
// iterations a QP may hold its wave while others wait (SFB_MID_SLICE, in stopping-check intervals; 0 = the fused launch for every batch)
uint32_t mid_slice(const DenseKernelParams &kp)
{
  const char *const v = sfb::knob("SFB_MID_SLICE");  // (read per call: the tests change it)
  const int checks    = v ? atoi(v) : 40;
  if (checks <= 0) return 0;
  return (uint32_t)checks * (kp.stop_check_iter >= 2 ? kp.stop_check_iter : 25u);
}
// waves of the loop kernel the current device holds at once
int mid_resident(const DenseKernelParams &kp)
{
  const char *const v = sfb::knob("SFB_MID_GRID");  // tests: a tiny grid forces the split, time-sliced launch
  const int forced    = v ? atoi(v) : 0;
  if (forced > 0) return forced;
  // The occupancy query and the device attributes behind it cost host time on every call (two to three per solve): cached per
  // (device, kernel instance = block count, LDS bytes), which is all the answer depends on.
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
  const int nb = (kp.n + kp.m + 15) / 16;
  const unsigned long long key = ((unsigned long long)(unsigned)dev << 48) | ((unsigned long long)(unsigned)nb << 40) |
                                 (unsigned long long)qp_dense_mid_lds_bytes(kp.n, kp.m);
  static std::mutex mu;
  static std::map<unsigned long long, int> cache;
  {
    std::lock_guard<std::mutex> lk(mu);
    const auto it = cache.find(key);
    if (it != cache.end()) return it->second;
  }
  int r = 0;
  MidLaunch a{MID_RESIDENT, 0, 0, nullptr, 0, nullptr, 0, &r, nullptr};
  if (mid_launch(kp, QpBatch{}, a) != hipSuccess) return 0;
  std::lock_guard<std::mutex> lk(mu);
  cache[key] = r;
  return r;
}
}  // namespace

// Device memory a launch needs beyond its arguments: nothing for a batch the chip holds at once (the fused kernel);
// otherwise one record per QP and the queue of the loop launch.
size_t qp_dense_mid_ws_bytes(const DenseKernelParams &kp, int64_t batch)
{
  if (mid_slice(kp) == 0) return 0;
  const int res = mid_resident(kp);
  if (res <= 0 || batch <= (int64_t)res) return 0;
  return (size_t)batch * mid_save_doubles(kp.n, kp.m) * sizeof(double) + 256 + 2 * (size_t)batch * sizeof(unsigned long long);
}

hipError_t qp_dense_mid_trace_launch(const DenseKernelParams &kp, int64_t batch, const QpBatch &g, hipStream_t stream, double *trace, int trace_cap,
                                     double *phase_us)
{
  const int k = kp.n + kp.m;
  if (k > kDenseMidMaxK || k < 1) return hipErrorInvalidValue;
  MidLaunch a{MID_TRACE, (unsigned)batch, (unsigned)batch, nullptr, 0, nullptr, 0, nullptr, stream};
  a.trace     = trace;
  a.trace_cap = trace_cap;
  a.phase_us  = phase_us;
  return mid_launch(kp, g, a);
}

hipError_t qp_dense_mid_launch(const DenseKernelParams &kp, int64_t batch, const QpBatch &g, hipStream_t stream, void *workspace)
{
  const int k = kp.n + kp.m;
  if (k > kDenseMidMaxK || k < 1) return hipErrorInvalidValue;
  const size_t need = qp_dense_mid_ws_bytes(kp, batch);
  if (need == 0) return mid_launch(kp, g, MidLaunch{MID_FUSED, (unsigned)batch, (unsigned)batch, nullptr, 0, nullptr, 0, nullptr, stream});
  // split launch: setup -> loop (persistent, time-sliced) -> finish
  char *buf        = static_cast<char *>(workspace);
  bool async_alloc = true;
  hipError_t e     = hipSuccess;
  if (buf == nullptr) {
    e = hipMallocAsync(reinterpret_cast<void **>(&buf), need, stream);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      async_alloc = false;
      e           = hipMalloc(reinterpret_cast<void **>(&buf), need);
      if (e != hipSuccess) return e;
    }
  }
  const size_t wsd  = mid_save_doubles(kp.n, kp.m);
  const size_t qoff = (size_t)batch * wsd * sizeof(double);
  unsigned *queue   = reinterpret_cast<unsigned *>(buf + qoff);
  double *ws        = reinterpret_cast<double *>(buf);
  e = hipMemsetAsync(queue, 0, need - qoff, stream);
  if (e == hipSuccess) e = mid_launch(kp, g, MidLaunch{MID_SETUP, (unsigned)batch, (unsigned)batch, ws, wsd, nullptr, 0, nullptr, stream});
  if (e == hipSuccess) {
    const int res = mid_resident(kp);
    const unsigned grid = (unsigned)((int64_t)res < batch ? res : batch);
    e = mid_launch(kp, g, MidLaunch{MID_LOOP, grid, (unsigned)batch, ws, wsd, queue, mid_slice(kp), nullptr, stream});
  }
  if (e == hipSuccess) e = mid_launch(kp, g, MidLaunch{MID_FINISH, (unsigned)batch, (unsigned)batch, ws, wsd, nullptr, 0, nullptr, stream});
  if (workspace == nullptr) {
    if (async_alloc) {
      (void)hipFreeAsync(buf, stream);
    } else {
      (void)hipStreamSynchronize(stream);
      (void)hipFree(buf);
    }
  }
  return e;
}

}  // namespace sfb
