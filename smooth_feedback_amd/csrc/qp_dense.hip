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

#include "knobs.h"
#include "../../include/sfb.h"
#include "qp_dense_common.h"

namespace sfb {

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

// GPA: P and A stay in global memory (qp_dense_common.h, pa_run) -- for k > 32 their LDS copies are a third to a half of
// the block's LDS and the kernel's throughput is proportional to the blocks a CU holds (measured with padded LDS requests:
// (20, 40) 5 -> 4 -> 3 blocks per CU = 3.6 -> 2.9 -> 2.2 x 10^8 QP-iterations/s).
template<int KP, int MODE, bool GPA = false>
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
  Lds s            = carve(smem, n, m, k, GPA);

  const QpBatch g{gP, gq, gA, gl, gu, gwx, gwy, gx, gy, gobj, giter, gcode};
  double c     = 1.0;
  const unsigned long long t0_ticks = wall_clock64();  // (:376 takes it after scaling; setup is microseconds here)
  int ret_code = qp_setup<GPA>(s, kp, n, m, b, g, lane, c);

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
    wave_lds_fence();
    if (isx) xs = (1.0 / sxv) * s.xv[xi];
    if (isc) {
      ys       = c * ((1.0 / syv) * s.yv[ci]);
      double t = 0.0;
      pa_run<GPA>(s.A + ci, m, n, [&](int j, double a) { t = fma(syv * a, s.xv[j], t); });
      zs = t;
    }
    wave_lds_fence();
  }

  if constexpr (MODE == SWEEP_READLANE_LDS) {  // the zero slot of the pipelined backward sweep (nothing in the loop writes s.temp)
    if (lane == 0) s.temp[k - 1] = 0.0;
    wave_lds_fence();
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
      // L' from LDS (row j of the packed factor, one entry per lane).  The loads do not depend on the chain: they are
      // issued PF steps ahead of their use, so that only [v_readlane -> v_fma] is left on the dependent path (a rolled
      // loop waits for every load: one LDS latency per step, measured 4 x slower at k = 60).  Steps j >= k and lanes
      // >= j read the zero slot: exact no-ops on finite data, like the zero-padded registers of the forward sweep.
      constexpr int PF = 6;
      const int zidx = (int)(s.temp - s.W) + k - 1;  // s.temp[k-1]: kept at 0.0 for the whole loop (see below)
      auto lidx = [&](const int J) { return (lane < J && J < k) ? tri(J, lane) : zidx; };
      double pf[PF];
#pragma unroll
      for (int e = 0; e < PF; ++e) pf[e] = s.W[lidx(KP - 1 - e)];
#pragma unroll
      for (int J = KP - 1; J > 0; --J) {
        const int slot  = (KP - 1 - J) % PF;
        const double lj = pf[slot];
        if (J - PF > 0) pf[slot] = s.W[lidx(J - PF)];
        const double tj = (J < k) ? lane_bcast(t, J < k ? J : 0) : 0.0;
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
      wave_lds_fence();
      ret_code = qp_check_stopping<GPA>(s, kp, n, m, lane);
      if (ret_code < 0 && max_time_exceeded(kp.max_time_ns, t0_ticks)) ret_code = SFB_QP_MAX_TIME;  // :504-507
      wave_lds_fence();
    }
  }

  // ---- scaled iterate back to original order ----
  if (isx) s.xv[xi] = xs;
  if (isc) s.yv[ci] = ys;
  wave_lds_fence();

  qp_finish<GPA>(s, kp, n, m, c, b, g, lane, ret_code, iter);
}

// 48 < k <= 64: the GPA kernel (no LDS copies of P and A): (20, 40) 5 -> 8 blocks per CU, 68 k -> 97 k QP/s under the reference
// benchmark's parameters, (32, 32) 4 -> 7 blocks, 0.80 -> 0.98 M QP/s.  Not for 32 < k <= 48: that kernel's 209 VGPRs allow 8
// blocks per CU, which its LDS request with the copies already reaches -- (16, 32) 155 k -> 145 k QP/s with the stopping checks'
// global reads.  SFB_QP_DENSE_PA_LDS=1 (A/B, tests): the kernel with the copies at every size.
static bool dense_pa_global(int k)
{
  static const bool off = [] { const char *v = sfb::knob("SFB_QP_DENSE_PA_LDS"); return v && v[0] == '1'; }();
  return k > 48 && !off;
}

size_t qp_dense_lds_bytes(int n, int m)
{
  const int k = n + m;
  const size_t pa = dense_pa_global(k) ? 0 : (size_t)n * n + (size_t)m * n;
  const size_t doubles = ((size_t)k * (k + 1)) / 2 + pa + 5 * (size_t)n + 8 * (size_t)m + (size_t)k;
  const size_t ints    = (size_t)k + (size_t)m;
  return doubles * sizeof(double) + ((ints * sizeof(int) + 15) / 16) * 16;
}

hipError_t qp_dense_launch(const DenseKernelParams &kp, int64_t batch, const double *P, const double *q,
                           const double *A, const double *l, const double *u, const double *wx, const double *wy,
                           double *x, double *y, double *obj, uint32_t *iter, int32_t *code, hipStream_t stream,
                           void *workspace)
{
  const int k        = kp.n + kp.m;
  const char *env4 = sfb::knob("SFB_QP_DENSE4");  // A/B and tests: 0 selects the one-QP-per-wave kernels
  const int dense4 = env4 ? atoi(env4) : 1;
  if (k <= 32 && dense4 && kp.max_time_ns < 0) {  // (a time limit is implemented by the one-QP-per-wave kernels)
    const QpBatch g{P, q, A, l, u, wx, wy, x, y, obj, iter, code};
    return qp_dense4_launch(kp, batch, g, stream, workspace);
  }
  if (k > 32 && qp_dense_mid_enabled() && !sfb::knob("SFB_QP_SWEEP")) {  // 32 < k <= 64: the on-chip block-sweep kernel (qp_dense_mid.hip)
    const QpBatch g{P, q, A, l, u, wx, wy, x, y, obj, iter, code};
    return qp_dense_mid_launch(kp, batch, g, stream, workspace);
  }
  size_t lds         = qp_dense_lds_bytes(kp.n, kp.m);
  if (const char *pad = sfb::knob("SFB_QP_LDS_PAD")) lds += (size_t)atoi(pad);  // occupancy experiments only
  const dim3 grid((unsigned)batch), block(kWave);
#define SFB_LAUNCH(KPV, MODE)                                                                                   \
  hipLaunchKernelGGL((qp_dense_kernel<KPV, MODE>), grid, block, lds, stream, kp, P, q, A, l, u, wx, wy, x, y, \
                     obj, iter, code)
  static const int force_mode = sfb::knob("SFB_QP_SWEEP") ? atoi(sfb::knob("SFB_QP_SWEEP")) : -1;  // A/B only
  if (k <= 32 && force_mode != SWEEP_READLANE) {
    SFB_LAUNCH(32, SWEEP_DPP32);
  } else if (k <= 16) {
    SFB_LAUNCH(16, SWEEP_READLANE);
  } else if (k <= 32) {
    SFB_LAUNCH(32, SWEEP_READLANE);
  } else if (k <= 48) {
    SFB_LAUNCH(48, SWEEP_READLANE_LDS);
  } else {
    if (dense_pa_global(k)) hipLaunchKernelGGL((qp_dense_kernel<64, SWEEP_READLANE_LDS, true>), grid, block, lds, stream, kp, P, q, A, l, u, wx, wy,
                                               x, y, obj, iter, code);
    else SFB_LAUNCH(64, SWEEP_READLANE_LDS);
  }
#undef SFB_LAUNCH
  return hipGetLastError();
}

}  // namespace sfb
