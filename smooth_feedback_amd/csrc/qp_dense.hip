// Batched dense ADMM QP solver for gfx950: ONE QP PER WAVEFRONT, KKT factor in LDS + VGPRs.
//
// Replaces, per batch item, smooth::feedback::solve_qp for QuadraticProgram<M,N,double>
// (reference qp_solver.hpp:343-568, :574-644, :673-730, :92-204).  Arithmetic (association of every
// element-wise expression, accumulation order of every dot product / triangular solve, true
// divisions, sqrt) is kept identical to oracle/qp_oracle.c so the two agree bit-for-bit; the file
// is compiled with -ffp-contract=off and fma() appears only where the oracle spells it.
//
// Mapping
//   - workgroup = 1 wavefront = 1 QP; k = n+m <= 64; lane i owns row i of the (permuted) KKT
//     system.  Iteration counts differ per QP, so the hardware workgroup dispatcher is the work
//     queue: a wave that finishes frees its slot for the next QP.
//   - setup (scaling, KKT fill, pivoted LDL') runs on LDS-resident data;
//   - for the ADMM loop each lane keeps row i of L (forward sweep) and column i of L (backward
//     sweep) in VGPRs; the pivot of every elimination step is broadcast with v_readlane (SGPR),
//     so one sweep step is 2 x v_readlane + 1 x v_fma_f64;
//   - the row permutation of the pivoted factorisation is folded into the lane assignment: lane i
//     carries variable perm[i] for the whole loop, so no permutation is applied per iteration;
//   - stopping tests (every stop_check_iter iterations) scatter the unscaled iterates to LDS in
//     original order and run the mat-vecs one row per lane; infinity norms are DPP reductions.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdlib>

#include "../../include/sfb.h"
#include "ldlt_wave.h"
#include "qp_dense_kernel.h"
#include "wave_util.h"

namespace sfb {

namespace {

struct Lds {
  double *W, *P, *A, *q, *l, *u, *sx, *sy, *rho, *xv, *yv, *zus, *dxus, *dyus, *temp;
  int *perm, *LU;
};

__device__ __forceinline__ Lds carve(double *base, int n, int m, int k)
{
  Lds s;
  double *p = base;
  s.W    = p; p += (k * (k + 1)) >> 1;
  s.P    = p; p += n * n;
  s.A    = p; p += m * n;
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
    for (int row = 0; row < n; ++row) t = fmax(t, fabs(s.P[row + lane * n]));
    if (t == 0.0) t = 1.0;
    s.temp[lane] = t;
  }
  wave_sync();
  // :693
  double sum = s.temp[0];
  for (int j = 1; j < n; ++j) sum += s.temp[j];
  const double mean = sum / (double)n;
  double qn         = 0.0;
  for (int j = 0; j < n; ++j) qn = fmax(qn, fabs(s.q[j]));
  const double c = 1.0 / fmax(fmax(1e-6, mean), qn);
  wave_sync();

  int iter = 0;
  double crit;
  do {  // :698-729
    double inc = 0.0;
    if (isx) {
      const double sxc = s.sx[lane];
      for (int row = 0; row < n; ++row) inc = fmax(inc, fabs(c * s.sx[row] * sxc * s.P[row + lane * n]));
      for (int row = 0; row < m; ++row) inc = fmax(inc, fabs(s.sy[row] * sxc * s.A[row + lane * m]));
    } else if (isc) {
      const double syr = s.sy[ci];
      for (int col = 0; col < n; ++col) inc = fmax(inc, fabs(syr * s.sx[col] * s.A[ci + col * m]));
    }
    if (inc == 0.0) inc = 1.0;
    wave_sync();  // every lane has read the old sx/sy
    const double f = sqrt(1.0 / fmax(inc, 1e-8));
    if (isx) s.sx[lane] = f * s.sx[lane];
    if (isc) s.sy[ci] = f * s.sy[ci];
    crit = wave_max((isx || isc) ? fabs(inc - 1.0) : 0.0);
    wave_sync();
  } while (iter++ < 10 && crit > 0.1);
  return c;
}

// rows of the original-order mat-vecs, fixed accumulation order (ascending inner index, fma)
__device__ __forceinline__ double row_A(const Lds &s, int n, int m, int i, const double *v)
{
  double r = 0.0;
  for (int j = 0; j < n; ++j) r = fma(s.A[i + j * m], v[j], r);
  return r;
}
__device__ __forceinline__ double row_At(const Lds &s, int n, int m, int j, const double *v)
{
  (void)n;
  double r = 0.0;
  for (int i = 0; i < m; ++i) r = fma(s.A[i + j * m], v[i], r);
  return r;
}
__device__ __forceinline__ double row_P(const Lds &s, int n, int i, const double *v)
{
  double r = 0.0;
  for (int j = 0; j < n; ++j) r = fma(s.P[i + j * n], v[j], r);
  return r;
}

// QPSolver::check_stopping, qp_solver.hpp:574-644 on xv(=x_us), yv(=y_us), zus, dxus, dyus in LDS.
// Returns a QPSolutionStatus or -1 (std::nullopt).  Wave-uniform.
__device__ inline int qp_check_stopping(const Lds &s, const DenseKernelParams &kp, const int n, const int m,
                                        const int lane)
{
  const bool ln = lane < n, lm = lane < m;
  const double inf = INFINITY;

  // OPTIMALITY :584-594
  const double Ax      = lm ? row_A(s, n, m, lane, s.xv) : 0.0;
  const double Ax_norm = wave_max(fabs(Ax));
  const double zi      = lm ? s.zus[lane] : 0.0;
  const double r_norm  = wave_max(lm ? fabs(Ax - zi) : 0.0);
  const double z_norm  = wave_max(fabs(zi));
  if (r_norm <= kp.eps_abs + kp.eps_rel * fmax(Ax_norm, z_norm)) {
    const double Px  = ln ? row_P(s, n, lane, s.xv) : 0.0;
    const double Aty = ln ? row_At(s, n, m, lane, s.yv) : 0.0;
    const double qi  = ln ? s.q[lane] : 0.0;
    const double dual_scale = fmax(fmax(wave_max(fabs(Px)), wave_max(fabs(qi))), wave_max(fabs(Aty)));
    const double res        = ln ? Px + (qi + Aty) : 0.0;
    if (wave_max(fabs(res)) <= kp.eps_abs + kp.eps_rel * dual_scale) return SFB_QP_OPTIMAL;
  }

  // PRIMAL INFEASIBILITY :598-621
  {
    const double Aty      = ln ? row_At(s, n, m, lane, s.dyus) : 0.0;
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
    const double Adx     = lm ? row_A(s, n, m, lane, s.dxus) : 0.0;
    const double dx_norm = wave_max(ln ? fabs(s.dxus[lane]) : 0.0);
    const double Pdx     = ln ? row_P(s, n, lane, s.dxus) : 0.0;
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
  wave_sync();

  // Hp (lower, row r per lane) :159-177 and h :179-182
  double h = 0.0;
  if (lane < n) {
    const int r      = lane;
    const double sxr = s.sx[r];
    for (int cc = 0; cc <= r; ++cc) {
      double v = c * s.sx[cc] * s.P[cc + r * n] * sxr;
      if (cc == r) v += kp.delta;
      s.W[tri(r, cc)] = v;
    }
    h = -c * (sxr * s.q[r]);
  } else if (lane < K) {
    const int a = lane - n, row = s.LU[a];
    const double syr = s.sy[row];
    for (int j = 0; j < n; ++j) s.W[tri(lane, j)] = syr * s.A[row + j * m] * s.sx[j];
    for (int j = n; j < lane; ++j) s.W[tri(lane, j)] = 0.0;
    s.W[tri(lane, lane)] = 0.0 - kp.delta;
    h                     = (a < nl) ? syr * s.l[row] : syr * s.u[row];
  }
  wave_sync();

  if (!ldlt_factor_lds(K, s.W, s.perm, s.temp, lane)) return;  // :187-190

  // :192-195  t += Hp^-1 (h - Hsym t); Hsym entries are recomputed (same products as above)
  double t = 0.0;
  double *tv = s.dxus;   // K <= k scratch: dxus(n)+dyus(m) are contiguous
  double *xch = s.temp;
  for (uint32_t it = 0; it != kp.polish_iter; ++it) {
    if (lane < K) tv[lane] = t;
    wave_sync();
    double res = 0.0;
    if (lane < n) {
      const int r      = lane;
      const double sxr = s.sx[r];
      double acc       = 0.0;
      for (int j = 0; j < n; ++j) {
        const int a = (j < r) ? j : r, b = (j < r) ? r : j;  // upper entry (a,b)
        acc         = fma(c * s.sx[a] * s.P[a + b * n] * s.sx[b], tv[j], acc);
      }
      for (int a = 0; a < na; ++a) {
        const int row = s.LU[a];
        acc           = fma(s.sy[row] * s.A[row + r * m] * sxr, tv[n + a], acc);
      }
      res = h - acc;
    } else if (lane < K) {
      const int row    = s.LU[lane - n];
      const double syr = s.sy[row];
      double acc       = 0.0;
      for (int j = 0; j < n; ++j) acc = fma(syr * s.A[row + j * m] * s.sx[j], tv[j], acc);
      res = h - acc;
    }
    wave_sync();
    const double d = ldlt_solve_lds(K, s.W, s.perm, xch, res, lane);
    t += d;
  }
  // :199-201
  if (lane < n) s.xv[lane] = t;
  else if (lane < K) s.yv[s.LU[lane - n]] = t;
  wave_sync();
}

}  // namespace

// Triangular-sweep engines of the ADMM loop (all produce the same bits):
//   SWEEP_READLANE      k <= KP: rows AND columns of L in VGPRs, pivot broadcast by v_readlane
//   SWEEP_READLANE_LDS  k <= KP: rows of L in VGPRs, backward sweep reads L' from LDS
//   SWEEP_DPP32         k <= 32: 16x16 blocks of L in VGPRs, pivot broadcast fused into the FMA by
//                       DPP row_newbcast (v_fmac_f64_dpp); forward sweep on lanes 0-31 (rows 0,1),
//                       backward sweep on lanes 32-63 (rows 2,3) sharing the same registers.
enum { SWEEP_READLANE = 0, SWEEP_READLANE_LDS = 1, SWEEP_DPP32 = 2 };

#ifndef SFB_QP_WAVES_PER_EU
#define SFB_QP_WAVES_PER_EU 2
#endif

template<int J>
struct DppSweep {
  // forward, ascending j: rows 0 / 1 in-row chains and the row-1 update by block 0
  static __device__ __forceinline__ void fwd_diag0(double &t, const double (&A)[16])
  {
    if constexpr (J < 15) {
      fmac_rowbcast_self<J, 0x1>(t, A[J]);
      DppSweep<J + 1>::fwd_diag0(t, A);
    }
  }
  static __device__ __forceinline__ void fwd_diag1(double &t, const double (&A)[16])
  {
    if constexpr (J < 15) {
      fmac_rowbcast_self<J, 0x2>(t, A[J]);
      DppSweep<J + 1>::fwd_diag1(t, A);
    }
  }
  static __device__ __forceinline__ void fwd_off(double &t, const double x, const double (&B)[16])
  {
    if constexpr (J < 16) {
      fmac_rowbcast<J, 0x2>(t, x, B[J]);
      DppSweep<J + 1>::fwd_off(t, x, B);
    }
  }
  // backward, descending j (J counts 15 -> 0): rows 3 / 2
  static __device__ __forceinline__ void bwd_diag1(double &t, const double (&A)[16])
  {
    if constexpr (J >= 1) {
      fmac_rowbcast_self<J, 0x8>(t, A[J]);
      DppSweep<J - 1>::bwd_diag1(t, A);
    }
  }
  static __device__ __forceinline__ void bwd_diag0(double &t, const double (&A)[16])
  {
    if constexpr (J >= 1) {
      fmac_rowbcast_self<J, 0x4>(t, A[J]);
      DppSweep<J - 1>::bwd_diag0(t, A);
    }
  }
  static __device__ __forceinline__ void bwd_off(double &t, const double x, const double (&B)[16])
  {
    if constexpr (J >= 0) {
      fmac_rowbcast<J, 0x4>(t, x, B[J]);
      DppSweep<J - 1>::bwd_off(t, x, B);
    }
  }
};

template<int KP, int MODE>
__global__ void __launch_bounds__(64, SFB_QP_WAVES_PER_EU) qp_dense_kernel(const DenseKernelParams kp, const double *__restrict__ gP,
                                                      const double *__restrict__ gq, const double *__restrict__ gA,
                                                      const double *__restrict__ gl, const double *__restrict__ gu,
                                                      const double *__restrict__ gwx, const double *__restrict__ gwy,
                                                      double *__restrict__ gx, double *__restrict__ gy,
                                                      double *__restrict__ gobj, uint32_t *__restrict__ giter,
                                                      int32_t *__restrict__ gcode)
{
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane   = threadIdx.x;
  const int n      = kp.n, m = kp.m, k = n + m;
  const size_t b   = blockIdx.x;
  const Lds s      = carve(smem, n, m, k);
  const double inf = INFINITY;

  // ---- load the problem (coalesced, batch-major contiguous) ----
  {
    const double *P = gP + b * (size_t)(n * n);
    const double *A = gA + b * (size_t)(m * n);
    for (int i = lane; i < n * n; i += kWave) s.P[i] = P[i];
    for (int i = lane; i < m * n; i += kWave) s.A[i] = A[i];
    if (lane < n) s.q[lane] = gq[b * n + lane];
    if (lane < m) {
      s.l[lane] = gl[b * m + lane];
      s.u[lane] = gu[b * m + lane];
    }
    if (lane < n) s.sx[lane] = 1.0;  // analyze(): :306-308
    if (lane < m) s.sy[lane] = 1.0;
  }
  wave_sync();

  // ---- scaling :347 ----
  double c = 1.0;
  if (kp.scaling) c = qp_scale(s, n, m, lane);

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
  wave_sync();

  // ---- KKT (lower triangle, row r per lane) :399-404 ----
  if (lane < n) {
    const int r      = lane;
    const double sxr = s.sx[r];
    for (int cc = 0; cc <= r; ++cc) {
      double v = c * s.sx[cc] * s.P[cc + r * n] * sxr;
      if (cc == r) v += kp.sigma;
      s.W[tri(r, cc)] = v;
    }
  } else if (lane < k) {
    const int i      = lane - n;
    const double syi = s.sy[i];
    for (int j = 0; j < n; ++j) s.W[tri(lane, j)] = syi * s.A[i + j * m] * s.sx[j];
    for (int j = n; j < lane; ++j) s.W[tri(lane, j)] = 0.0;
    s.W[tri(lane, lane)] = 1.0 / (-s.rho[i]);
  }
  wave_sync();

  // ---- pivoted LDL' :428-433 ----
  if (!ldlt_factor_lds(k, s.W, s.perm, s.temp, lane)) ret_code = SFB_QP_UNKNOWN;

  // ---- register-resident factor: Lr[j] = L(i,j) (j<i), Lc[j] = L(j,i) (j>i), d = D(i) ----
  constexpr bool LC_REGS = (MODE == SWEEP_READLANE);
  constexpr bool DPP     = (MODE == SWEEP_DPP32);
  double Lr[DPP ? 1 : KP];
  double Lc[LC_REGS ? KP : 1];
  double Ad[DPP ? 16 : 1], Bo[DPP ? 16 : 1];  // negated 16x16 blocks (SWEEP_DPP32)
  double dgi = 1.0;
  const bool inmat = lane < k;
  if constexpr (!DPP) {
#pragma unroll
    for (int j = 0; j < KP; ++j) Lr[j] = (inmat && j < lane && j < k) ? s.W[tri(lane, j)] : 0.0;
    if constexpr (LC_REGS) {
#pragma unroll
      for (int j = 0; j < KP; ++j) Lc[j] = (inmat && j > lane && j < k) ? s.W[tri(j, lane)] : 0.0;
    }
  } else {
    // lane = 16*row + cc.  rows 0,1: forward, system row i = lane;  rows 2,3: backward, column
    // i = lane - 32.  Ad = diagonal block, Bo = off-diagonal block, zero outside the factor.
    const int row = lane >> 4, cc = lane & 15;
    const int i   = (row & 1) * 16 + cc;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      double a = 0.0, bb = 0.0;
      if (i < k) {
        if (row == 0) {
          if (j < i) a = s.W[tri(i, j)];
        } else if (row == 1) {
          if (16 + j < i) a = s.W[tri(i, 16 + j)];
          bb = s.W[tri(i, j)];
        } else if (row == 2) {
          if (j > i) a = s.W[tri(j, i)];                  // j < 16 <= k is not guaranteed:
          if (j >= k) a = 0.0;                             //   k < 16 leaves these outside the factor
          if (16 + j < k) bb = s.W[tri(16 + j, i)];
        } else {
          if (16 + j > i && 16 + j < k) a = s.W[tri(16 + j, i)];
        }
      }
      Ad[j] = -a;
      Bo[j] = -bb;
    }
  }
  if (inmat) dgi = s.W[tri(lane, lane)];

  // ---- lane role: variable v = perm[lane] ----
  const int v     = inmat ? s.perm[lane] : 0;
  const bool isx  = inmat && v < n;
  const bool isc  = inmat && v >= n;
  const int ci    = isc ? v - n : 0;
  const int xi    = isx ? v : 0;
  const double qc   = isx ? c * s.sx[xi] * s.q[xi] : 0.0;  // :450
  const double sxv  = isx ? s.sx[xi] : 1.0;
  const double syv  = isc ? s.sy[ci] : 1.0;
  const double rho  = isc ? s.rho[ci] : 1.0;
  const double rinv = 1.0 / rho;                           // rho_.cwiseInverse()
  const double lo   = isc ? syv * s.l[ci] : 0.0;           // :473
  const double hi   = isc ? syv * s.u[ci] : 0.0;           // :474

  // ---- initial iterate :436-445 ----
  double xs = 0.0, ys = 0.0, zs = 0.0;  // x (x-lanes); y, z (constraint lanes)
  if (gwx != nullptr) {
    if (lane < n) s.xv[lane] = gwx[b * n + lane];
    if (lane < m) s.yv[lane] = gwy[b * m + lane];
    wave_sync();
    if (isx) xs = (1.0 / sxv) * s.xv[xi];
    if (isc) {
      ys       = c * ((1.0 / syv) * s.yv[ci]);
      double t = 0.0;
      for (int j = 0; j < n; ++j) t = fma(syv * s.A[ci + j * m], s.xv[j], t);
      zs = t;
    }
    wave_sync();
  }

  // ---- ADMM loop :447-510 ----
  uint32_t iter        = 0;
  const uint32_t sci   = kp.stop_check_iter;
  const uint32_t maxit = kp.max_iter;
  // iter % sci == 1 (:465) without a per-iteration division: checks at 1, 1+sci, ...; never for sci<=1
  uint32_t next_chk = (sci >= 2) ? 1u : 0xFFFFFFFFu;
  for (; iter != maxit && ret_code < 0; ++iter) {
    // right-hand side :450-451 (already in permuted lane order)
    double t = isx ? (kp.sigma * xs - qc) : (isc ? (zs - rinv * ys) : 0.0);

    // K^-1 t : forward sweep, D^-1, backward sweep :462
    // (steps j >= k-1 multiply zero-padded factor entries: exact no-ops, no per-step branch)
    if constexpr (DPP) {
      double ev, od, lo2, hi2;
      DppSweep<0>::fwd_diag0(t, Ad);       // rows 0..15, in-row pivot broadcast
      cross_lane_fence(t);
      row_swap16(t, ev, od);               // block-0 solution -> row 1
      DppSweep<0>::fwd_off(t, ev, Bo);     // rows 16..31 -= L10 * x0 (j ascending)
      DppSweep<0>::fwd_diag1(t, Ad);       // rows 16..31
      t = (fabs(dgi) > DBL_MIN) ? t / dgi : 0.0;
      half_swap32(t, lo2, hi2);            // lanes 32..63 <- lanes 0..31
      t = lo2;
      DppSweep<15>::bwd_diag1(t, Ad);      // row 3: block 1, j descending
      cross_lane_fence(t);
      row_swap16(t, ev, od);               // block-1 solution -> row 2
      DppSweep<15>::bwd_off(t, od, Bo);    // row 2: block 0 -= L10' * x1 (j descending)
      DppSweep<15>::bwd_diag0(t, Ad);      // row 2: block 0
      cross_lane_fence(t);
      half_swap32(t, lo2, hi2);            // lanes 0..31 <- lanes 32..63
      t = hi2;
    } else {
#pragma unroll
    for (int j = 0; j < KP - 1; ++j) {
      const double tj = lane_bcast(t, j);
      t               = fma(-Lr[j], tj, t);
    }
    t = (fabs(dgi) > DBL_MIN) ? t / dgi : 0.0;
    if constexpr (LC_REGS) {
#pragma unroll
      for (int j = KP - 1; j > 0; --j) {
        const double tj = lane_bcast(t, j);
        t               = fma(-Lc[j], tj, t);
      }
    } else {
      for (int j = k - 1; j > 0; --j) {
        const double tj = lane_bcast(t, j);
        const double lj = (lane < j) ? s.W[tri(j, lane)] : 0.0;
        t               = fma(-lj, tj, t);
      }
    }
    }

    const bool chk = (iter == next_chk);  // :465
    if (chk) next_chk += sci;
    const double xold = xs, yold = ys;

    // :470-477
    xs              = kp.alpha * t + kp.alpha_comp * xs;
    double zn       = kp.alpha * (rinv * t) + kp.alpha_comp * (rinv * ys) + zs;
    zn              = (zn < lo) ? lo : zn;  // cwiseMax(sy*l):  std::max(zn, lo)
    zn              = (hi < zn) ? hi : zn;  // cwiseMin(sy*u):  std::min(zn, hi)
    ys              = kp.alpha_comp * ys + kp.alpha * t + rho * zs - rho * zn;
    zs              = zn;

    if (chk) {  // :479-509
      if (isx) {
        s.xv[xi]   = sxv * xs;
        s.dxus[xi] = sxv * (xs - xold);
      }
      if (isc) {
        s.yv[ci]   = syv * ys / c;
        s.zus[ci]  = (1.0 / syv) * zs;
        s.dyus[ci] = syv * (ys - yold) / c;
      }
      wave_sync();
      ret_code = qp_check_stopping(s, kp, n, m, lane);
      wave_sync();
    }
  }

  // ---- scaled iterate back to original order ----
  if (isx) s.xv[xi] = xs;
  if (isc) s.yv[ci] = ys;
  wave_sync();

  // ---- polish :515-539 (a failed polish leaves Optimal, cf. :537 vs :544) ----
  if (ret_code == SFB_QP_OPTIMAL && kp.polish) qp_polish(s, kp, n, m, c, lane);

  // ---- un-scale and report :544-548 ----
  double xo = 0.0;
  if (lane < n) {
    xo            = s.sx[lane] * s.xv[lane];
    gx[b * n + lane] = xo;
    s.dxus[lane]  = xo;
  }
  if (lane < m) gy[b * m + lane] = s.sy[lane] * s.yv[lane] / c;
  wave_sync();
  if (gobj != nullptr) {
    if (lane < n) {
      double acc = 0.0;
      for (int j = 0; j < n; ++j) acc = fma(0.5 * s.P[lane + j * n], s.dxus[j], acc);
      s.temp[lane] = acc + s.q[lane];
    }
    wave_sync();
    if (lane == 0) {
      double o = 0.0;
      for (int i = 0; i < n; ++i) o = fma(s.dxus[i], s.temp[i], o);
      gobj[b] = o;
    }
  }
  if (lane == 0) {
    gcode[b] = (ret_code >= 0) ? ret_code : SFB_QP_MAX_ITERATIONS;
    if (giter != nullptr) giter[b] = iter;
  }
}

size_t qp_dense_lds_bytes(int n, int m)
{
  const int k = n + m;
  const size_t doubles = ((size_t)k * (k + 1)) / 2 + (size_t)n * n + (size_t)m * n + 5 * (size_t)n + 8 * (size_t)m + (size_t)k;
  const size_t ints    = (size_t)k + (size_t)m;
  return doubles * sizeof(double) + ((ints * sizeof(int) + 15) / 16) * 16;
}

hipError_t qp_dense_launch(const DenseKernelParams &kp, int64_t batch, const double *P, const double *q,
                           const double *A, const double *l, const double *u, const double *wx, const double *wy,
                           double *x, double *y, double *obj, uint32_t *iter, int32_t *code, hipStream_t stream)
{
  const int k        = kp.n + kp.m;
  size_t lds         = qp_dense_lds_bytes(kp.n, kp.m);
  if (const char *pad = getenv("SFB_QP_LDS_PAD")) lds += (size_t)atoi(pad);  // occupancy experiments only
  const dim3 grid((unsigned)batch), block(kWave);
#define SFB_LAUNCH(KPV, MODE)                                                                                   \
  hipLaunchKernelGGL((qp_dense_kernel<KPV, MODE>), grid, block, lds, stream, kp, P, q, A, l, u, wx, wy, x, y, \
                     obj, iter, code)
  static const int force_mode = getenv("SFB_QP_SWEEP") ? atoi(getenv("SFB_QP_SWEEP")) : -1;  // A/B only
  if (k <= 32 && force_mode != SWEEP_READLANE) {
    SFB_LAUNCH(32, SWEEP_DPP32);
  } else if (k <= 16) {
    SFB_LAUNCH(16, SWEEP_READLANE);
  } else if (k <= 32) {
    SFB_LAUNCH(32, SWEEP_READLANE);
  } else if (k <= 48) {
    SFB_LAUNCH(48, SWEEP_READLANE_LDS);
  } else {
    SFB_LAUNCH(64, SWEEP_READLANE_LDS);
  }
#undef SFB_LAUNCH
  return hipGetLastError();
}

}  // namespace sfb
