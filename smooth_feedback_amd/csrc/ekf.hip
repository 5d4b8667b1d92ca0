// Batched Lie-group EKF covariance kernels for gfx950: ONE FILTER PER LANE, all small matrices in
// registers, item-major (AoS) HBM layout staged through LDS so that every global access is a
// coalesced 512-byte wave transaction.
//
// Replaces the matrix part of smooth::feedback::EKF (reference ekf.hpp):
//   predict  :84-89 + euler step :96     P <- P + dt * symU(A P + P A' + Q)
//   update   :119-138                    S = triU(H symU(P) H' + R); K = (ldlt(symU(S)).solve(H P))';
//                                        delta = K r;  P <- symU((I - K H) P)
// The host keeps what needs the user's callbacks: A = -ad(f) + d^r f/dx at the estimate, H = d^r h/dx,
// r = y (-) h(g), and applies g <- g (+) delta.  Arithmetic order == oracle/ekf_oracle.c (k-ascending
// fma chains, pivoted LDL' as in ldlt of Eigen 3.4), so results are bit-identical to the oracle.
// These kernels are pure streaming: ~1.4 KB of I/O and ~1.5 kflop per item => HBM-bound.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>

#include "../../include/sfb.h"
#include "../../include/smooth_feedback_amd/detail/ekf_lane.hpp"
#include "ekf_kernel.h"
#include "knobs.h"
#include "ldlt_wave.h"
#include "wave_util.h"

namespace sfb {

namespace {

using namespace ekf_lane;  // tile I/O, SmallLdlt and the per-lane predict / update (include/smooth_feedback_amd/detail/ekf_lane.hpp)

}  // namespace

// One launch = optional predict (Euler substep) followed by optional update, fused per item.
#ifndef SFB_EKF_WPE
#define SFB_EKF_WPE 1  // waves per SIMD the fused kernel is compiled for (2: 256 registers in all, measured slower -- scripts/r4/experiments)
#endif
template<int N, int M, bool PREDICT, bool UPDATE>
__global__ void __launch_bounds__(64, SFB_EKF_WPE) ekf_kernel(const EkfArgs a)
{
  constexpr int NN = N * N, NP = NN | 1;
  constexpr int WMAX = (NN > M * N ? NN : M * N) > M * M ? (NN > M * N ? NN : M * N) : M * M;
  constexpr int TILE = (WMAX | 1) * kWave;
  __shared__ double lds[TILE];
  const int lane      = threadIdx.x;
  const int64_t item0 = (int64_t)blockIdx.x * kWave;
  const int64_t item  = item0 + lane;
  const bool live     = item < a.batch;

  double P[NN];
  tile_load<NN>(a.P, item0, a.batch, lds, lane);
  wave_sync();
#pragma unroll
  for (int e = 0; e < NN; ++e) P[e] = live ? lds[lane * NP + e] : 0.0;
  wave_sync();

  if constexpr (PREDICT) {
    double A[NN];
    tile_load<NN>(a.A, item0, a.batch, lds, lane);
    wave_sync();
#pragma unroll
    for (int e = 0; e < NN; ++e) A[e] = live ? lds[lane * NP + e] : 0.0;
    wave_sync();
    if (!a.q_shared) {
      tile_load<NN>(a.Q, item0, a.batch, lds, lane);
      wave_sync();
    }
    const double dt = live ? (a.dt_shared ? a.dt[0] : a.dt[item]) : 0.0;
    ekf_lane_predict<N>(P, A, [&](const int i, const int j) {
      return a.q_shared ? a.Q[i + j * N] : (live ? lds[lane * NP + i + j * N] : 0.0);
    }, dt);
    wave_sync();
  }

  if constexpr (UPDATE) {
    constexpr int MN = M * N, MNP = MN | 1;
    double H[MN];
    tile_load<MN>(a.H, item0, a.batch, lds, lane);
    wave_sync();
#pragma unroll
    for (int e = 0; e < MN; ++e) H[e] = live ? lds[lane * MNP + e] : 0.0;
    wave_sync();

    // R of this item in registers (upper part), then the per-lane update
    constexpr int MM = M * M, MMP = MM | 1;
    if (!a.r_shared) {
      tile_load<MM>(a.R, item0, a.batch, lds, lane);
      wave_sync();
    }
    double Rv[MM];
#pragma unroll
    for (int b = 0; b < M; ++b) {
#pragma unroll
      for (int aa = 0; aa < M; ++aa)
        Rv[aa + b * M] = (aa <= b) ? (a.r_shared ? a.R[aa + b * M] : (live ? lds[lane * MMP + aa + b * M] : ((aa == b) ? 1.0 : 0.0))) : 0.0;
    }
    wave_sync();
    double rv[M], delta[N];
#pragma unroll
    for (int aa = 0; aa < M; ++aa) rv[aa] = live ? a.r[item * M + aa] : 0.0;
    const bool ok = ekf_lane_update<N, M>(P, H, [&](const int aa, const int b) { return Rv[aa + b * M]; }, rv, delta);
    if (live) {
#pragma unroll
      for (int i = 0; i < N; ++i) a.delta[item * N + i] = delta[i];
    }
    if (a.info != nullptr && live) a.info[item] = ok ? 0 : 1;
  }

  // ---- write P back (coalesced through LDS) ----
  if (live) {
#pragma unroll
    for (int e = 0; e < NN; ++e) lds[lane * NP + e] = P[e];
  }
  wave_sync();
  tile_store<NN>(a.P, item0, a.batch, lds, lane);
}

// The fused step as a PERSISTENT wave (round 6; N * N even: 16-byte granules).  The kernel above is HBM-bound while a wave
// loads and idle towards the memory system while it computes (2 260 VALU instructions per tile of 64 filters, a quarter of a
// tile's time, with one wave per SIMD and nothing of its own in flight).  Here a wave walks over tiles and, before the update's
// arithmetic of tile i, requests the covariances of tile i + 1 STRAIGHT INTO LDS (global_load_lds_dwordx4: no register of the
// wave is involved, which is what the register-staged prefetch of round 4 ran out of): they land while the wave computes and
// stores.  Only the covariances: a request costs the wave ~100 cycles of issue per KB, and with EVERY input tile requested that
// way (built and measured: 0.298 ms per 1 048 576 filters against 0.284 for this form and 0.297-0.305 for the kernel above) the
// issue time eats what the overlap gains.  The LDS image of such a load is lane-linear (destination = base + 16 lane), so the
// odd-stride padding of the staged tiles is not available; the bank conflicts of the per-lane walk (18 granules per filter:
// lanes l and l + 8 of a 16-lane pass meet) are removed by exchanging the two granules of every pair for the filters with bit 3
// of the lane set -- on the SOURCE side, granule g of the tile is fetched into position g ^ ((g / (8 * G)) & 1), and the reader
// applies the same involution.  Same arithmetic on the same values as ekf_kernel: bit-identical.
#ifndef SFB_EKF_GLDS_AUX
#define SFB_EKF_GLDS_AUX 0  // cache policy of the requests (2: non-temporal)
#endif
#ifndef SFB_EKF_LANDED_EARLY
#define SFB_EKF_LANDED_EARLY 0  // 1: wait for the next tile's covariances BEFORE this tile's stores join the queue (0: at the next tile's start)
#endif
template<int N, int M>
__global__ void __launch_bounds__(64, 1) ekf_fused_persistent_kernel(const EkfArgs a, const int64_t ntiles)
{
  constexpr int NN = N * N, NP = NN | 1, G = NN / 2;  // G: 16-byte granules per filter
  static_assert(NN % 2 == 0, "16-byte granules");
  constexpr int WMAX = (NN > M * N ? NN : M * N) > M * M ? (NN > M * N ? NN : M * N) : M * M;
  constexpr int TILE = (WMAX | 1) * kWave;
  __shared__ double lds[TILE];
  __shared__ __attribute__((aligned(16))) double nextP[NN * kWave];  // the next tile's covariances, swizzled lane-linear image
  const int lane = threadIdx.x;
  using lds_ptr  = __attribute__((address_space(3))) void *;
  auto request_P = [&](const int64_t tile) {
    const int64_t item0 = tile * kWave;
    const int granules  = (int)(a.batch - item0 < kWave ? a.batch - item0 : kWave) * G;
    const double *src   = a.P + item0 * NN;
#pragma unroll
    for (int w = 0; w < G; ++w) {  // G windows of 64 granules (1 KB per wave instruction)
      const int p = w * kWave + lane;
      int g       = p ^ ((p / (8 * G)) & 1);
      g           = g < granules ? g : 0;  // (a partial last tile: lanes beyond it fetch something valid, nobody reads it)
      __builtin_amdgcn_global_load_lds(src + 2 * g, (lds_ptr)(nextP + 2 * w * kWave), 16, 0, SFB_EKF_GLDS_AUX);
    }
  };
  auto landed = [] { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
  int64_t tile = blockIdx.x;
  if (tile < ntiles) request_P(tile);
  landed();
  for (; tile < ntiles; tile += gridDim.x) {
    const int64_t item0 = tile * kWave;
    const int64_t item  = item0 + lane;
    const bool live     = item < a.batch;
    if constexpr (!SFB_EKF_LANDED_EARLY) landed();
    wave_sync();
    double P[NN];
    {
      const int sw = (lane >> 3) & 1;
#pragma unroll
      for (int c = 0; c < G; ++c) {
        const vd2 v  = *reinterpret_cast<const vd2 *>(nextP + 2 * ((lane * G + c) ^ sw));
        P[2 * c]     = live ? v.x : 0.0;
        P[2 * c + 1] = live ? v.y : 0.0;
      }
    }
    wave_sync();  // (every lane has its covariance: the buffer is free for the next request)

    {  // predict, as in ekf_kernel
      double A[NN];
      tile_load<NN>(a.A, item0, a.batch, lds, lane);
      wave_sync();
#pragma unroll
      for (int e = 0; e < NN; ++e) A[e] = live ? lds[lane * NP + e] : 0.0;
      wave_sync();
      if (!a.q_shared) {
        tile_load<NN>(a.Q, item0, a.batch, lds, lane);
        wave_sync();
      }
      const double dt = live ? (a.dt_shared ? a.dt[0] : a.dt[item]) : 0.0;
      ekf_lane_predict<N>(P, A, [&](const int i, const int j) {
        return a.q_shared ? a.Q[i + j * N] : (live ? lds[lane * NP + i + j * N] : 0.0);
      }, dt);
      wave_sync();
    }

    constexpr int MN = M * N, MNP = MN | 1;
    double H[MN];
    tile_load<MN>(a.H, item0, a.batch, lds, lane);
    wave_sync();
#pragma unroll
    for (int e = 0; e < MN; ++e) H[e] = live ? lds[lane * MNP + e] : 0.0;
    wave_sync();
    constexpr int MM = M * M, MMP = MM | 1;
    if (!a.r_shared) {
      tile_load<MM>(a.R, item0, a.batch, lds, lane);
      wave_sync();
    }
    double Rv[MM];
#pragma unroll
    for (int b = 0; b < M; ++b) {
#pragma unroll
      for (int aa = 0; aa < M; ++aa)
        Rv[aa + b * M] = (aa <= b) ? (a.r_shared ? a.R[aa + b * M] : (live ? lds[lane * MMP + aa + b * M] : ((aa == b) ? 1.0 : 0.0))) : 0.0;
    }
    wave_sync();
    double rv[M], delta[N];
#pragma unroll
    for (int aa = 0; aa < M; ++aa) rv[aa] = live ? a.r[item * M + aa] : 0.0;
    // every ordinary load of this tile is consumed before the next tile's covariances are requested (the compiler waits for
    // ALL outstanding loads at the first use of an ordinary one: with the request in flight that would be the request too)
#pragma unroll
    for (int aa = 0; aa < M; ++aa) asm volatile("" : "+v"(rv[aa]));
    landed();
    if (tile + gridDim.x < ntiles) request_P(tile + gridDim.x);
    const bool ok = ekf_lane_update<N, M>(P, H, [&](const int aa, const int b) { return Rv[aa + b * M]; }, rv, delta);
    if (live) {
#pragma unroll
      for (int e = 0; e < NN; ++e) lds[lane * NP + e] = P[e];
    }
    if constexpr (SFB_EKF_LANDED_EARLY) landed();  // (... so that the wait at the next tile's start is not one for these stores' acknowledgements)
    if (live) {
#pragma unroll
      for (int i = 0; i < N; ++i) a.delta[item * N + i] = delta[i];
    }
    if (a.info != nullptr && live) a.info[item] = ok ? 0 : 1;
    wave_sync();
    tile_store<NN>(a.P, item0, a.batch, lds, lane);
    wave_sync();  // (the staging tile is free again)
  }
}

// The update for 6 < dof <= 10 (ny <= 3): one filter per lane as in ekf_kernel, but P alone fills a third of the
// register file (dof = 9: 162 VGPRs), so I - K H is formed one row at a time and the new covariance goes straight
// into the output tile instead of a second register copy.  Every entry is the same k-ascending fma chain as in
// ekf_kernel / the oracle: bit-identical.  (A fused predict does not fit: with A and the slope next to P the
// compiler spills and the launch is slower than the generic kernel's predict followed by this update -- measured.)
template<int N, int M>
__global__ void __launch_bounds__(64) ekf_update_wide_kernel(const EkfArgs a)
{
  constexpr int NN = N * N, NP = NN | 1;
  constexpr int TILE = NP * kWave;
  static_assert(M * N <= NN && M * M <= NN, "the P tile is the largest");
  __shared__ double lds[TILE];
  const int lane      = threadIdx.x;
  const int64_t item0 = (int64_t)blockIdx.x * kWave;
  const int64_t item  = item0 + lane;
  const bool live     = item < a.batch;

  double P[NN];
  tile_load<NN>(a.P, item0, a.batch, lds, lane);
  wave_sync();
#pragma unroll
  for (int e = 0; e < NN; ++e) P[e] = live ? lds[lane * NP + e] : 0.0;
  wave_sync();

  constexpr int MN = M * N, MNP = MN | 1;
  double H[MN];
  tile_load<MN>(a.H, item0, a.batch, lds, lane);
  wave_sync();
#pragma unroll
  for (int e = 0; e < MN; ++e) H[e] = live ? lds[lane * MNP + e] : 0.0;
  wave_sync();
  double T[MN], HP[MN];  // H * symU(P), H * P   (ekf.hpp:129, :134)
#pragma unroll
  for (int j = 0; j < N; ++j) {
#pragma unroll
    for (int aa = 0; aa < M; ++aa) {
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k) s1 = fma(H[aa + k * M], (k <= j) ? P[k + j * N] : P[j + k * N], s1);
#pragma unroll
      for (int k = 0; k < N; ++k) s2 = fma(H[aa + k * M], P[k + j * N], s2);
      T[aa + j * M]  = s1;
      HP[aa + j * M] = s2;
    }
  }
  SmallLdlt<M> F;
  {
    constexpr int MM = M * M, MMP = MM | 1;
    if (!a.r_shared) {
      tile_load<MM>(a.R, item0, a.batch, lds, lane);
      wave_sync();
    }
#pragma unroll
    for (int b = 0; b < M; ++b) {
#pragma unroll
      for (int aa = 0; aa < M; ++aa) {
        if (aa <= b) {
          double s = 0.0;
#pragma unroll
          for (int k = 0; k < N; ++k) s = fma(T[aa + k * M], H[b + k * M], s);
          const double rr = a.r_shared ? a.R[aa + b * M] : (live ? lds[lane * MMP + aa + b * M] : ((aa == b) ? 1.0 : 0.0));
          F.W[b][aa]      = s + rr;
        } else {
          F.W[b][aa] = 0.0;
        }
      }
    }
    wave_sync();
  }
  F.factor();
  double X[MN];  // S^-1 (H P), column by column
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double col[M];
#pragma unroll
    for (int aa = 0; aa < M; ++aa) col[aa] = HP[aa + j * M];
    F.solve(col);
#pragma unroll
    for (int aa = 0; aa < M; ++aa) X[aa + j * M] = col[aa];
  }
  double rv[M];  // delta = K r, K = X'   (:137)
#pragma unroll
  for (int aa = 0; aa < M; ++aa) rv[aa] = live ? a.r[item * M + aa] : 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double s = 0.0;
#pragma unroll
    for (int aa = 0; aa < M; ++aa) s = fma(X[aa + i * M], rv[aa], s);
    if (live) a.delta[item * N + i] = s;
  }
  if (a.info != nullptr && live) a.info[item] = F.ok ? 0 : 1;
  // P = symU((I - K H) P)   (:138): row i of I - K H, then the entries (i, j >= i) of the product, mirrored, straight
  // into the output tile (the old P is still needed by the rows that follow)
  double *Pl = lds + lane * NP;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double IKi[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
      double s = 0.0;
#pragma unroll
      for (int aa = 0; aa < M; ++aa) s = fma(X[aa + i * M], H[aa + k * M], s);
      IKi[k] = ((i == k) ? 1.0 : 0.0) - s;
    }
#pragma unroll
    for (int j = i; j < N; ++j) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k) s = fma(IKi[k], P[k + j * N], s);
      if (live) {
        Pl[i + j * N] = s;
        Pl[j + i * N] = s;
      }
    }
  }
  wave_sync();
  tile_store<NN>(a.P, item0, a.batch, lds, lane);
}

// Predict with boost::numeric::odeint::runge_kutta4 on the covariance ODE (ekf.hpp:86-96 with the stepper of
// tests/test_ekf.cpp:113-115); same stage formulas, coefficients and accumulation order as oracle_ekf_predict_rk4.
// The reference re-evaluates A = -ad(f(t_s, g)) + d^r f/dx at every stage time t_s (cov_ode, :84-89; g is frozen
// during the covariance step): A at t, A_mid at t + dt/2 (stages 2, 3), A_end at t + dt; nullptr = same as A.  One filter per lane: P, the stage state, the running
// sum and the upper triangle of the current slope stay in registers; A is re-read from its LDS tile at
// every use (four right-hand sides), the second tile stages P and Q.
template<int N>
__global__ void __launch_bounds__(64) ekf_rk4_kernel(const EkfArgs a)
{
  constexpr int NN = N * N, NP = NN | 1, NT = N * (N + 1) / 2, TILE = NP * kWave;
  __shared__ double ldsA[TILE];
  __shared__ double ldsS[TILE];
  const int lane      = threadIdx.x;
  const int64_t item0 = (int64_t)blockIdx.x * kWave;
  const int64_t item  = item0 + lane;
  const bool live     = item < a.batch;

  double P[NN];
  tile_load<NN>(a.P, item0, a.batch, ldsS, lane);
  tile_load<NN>(a.A, item0, a.batch, ldsA, lane);
  wave_sync();
#pragma unroll
  for (int e = 0; e < NN; ++e) P[e] = live ? ldsS[lane * NP + e] : 0.0;
  wave_sync();
  double qv[NT];
  if (!a.q_shared) {
    tile_load<NN>(a.Q, item0, a.batch, ldsS, lane);
    wave_sync();
  }
#pragma unroll
  for (int j = 0; j < N; ++j)
#pragma unroll
    for (int i = 0; i <= j; ++i)
      qv[i + j * (j + 1) / 2] = a.q_shared ? a.Q[i + j * N] : (live ? ldsS[lane * NP + i + j * N] : 0.0);
  const double *Al = ldsA + lane * NP;  // lanes past the end of the batch read the zeros tile_load put there
  const double dt  = live ? (a.dt_shared ? a.dt[0] : a.dt[item]) : 0.0;
  const double b1 = dt * (1.0 / 6.0), b2 = dt * (1.0 / 3.0), c2 = dt * 0.5, c4 = dt * 1.0;

  double X[NN], S[NN], kv[NT];
  auto rhs = [&]() {  // kv = upper triangle of symU(A X + X A' + Q)   (ekf.hpp:88)
#pragma unroll
    for (int j = 0; j < N; ++j) {
#pragma unroll
      for (int i = 0; i <= j; ++i) {
        double m1 = 0.0, m2 = 0.0;
#pragma unroll
        for (int k = 0; k < N; ++k) m1 = fma(Al[i + k * N], X[k + j * N], m1);
#pragma unroll
        for (int k = 0; k < N; ++k) m2 = fma(X[i + k * N], Al[j + k * N], m2);
        kv[i + j * (j + 1) / 2] = (m1 + m2) + qv[i + j * (j + 1) / 2];
      }
    }
  };
  auto kof = [&](int r, int c) { return (r <= c) ? kv[r + c * (c + 1) / 2] : kv[c + r * (r + 1) / 2]; };
#pragma unroll
  for (int e = 0; e < NN; ++e) X[e] = P[e];
  rhs();  // k1
#pragma unroll
  for (int c = 0; c < N; ++c)
#pragma unroll
    for (int r = 0; r < N; ++r) {
      const double kk = kof(r, c);
      S[r + c * N]    = P[r + c * N] + b1 * kk;
      X[r + c * N]    = P[r + c * N] + c2 * kk;
    }
  if (a.A_mid != nullptr) {  // time-dependent dynamics: the linearisation at t + dt/2 (stages 2 and 3, ekf.hpp:84-89)
    wave_sync();
    tile_load<NN>(a.A_mid, item0, a.batch, ldsA, lane);
    wave_sync();
  }
  rhs();  // k2
#pragma unroll
  for (int c = 0; c < N; ++c)
#pragma unroll
    for (int r = 0; r < N; ++r) {
      const double kk = kof(r, c);
      S[r + c * N]    = S[r + c * N] + b2 * kk;
      X[r + c * N]    = P[r + c * N] + c2 * kk;
    }
  rhs();  // k3
#pragma unroll
  for (int c = 0; c < N; ++c)
#pragma unroll
    for (int r = 0; r < N; ++r) {
      const double kk = kof(r, c);
      S[r + c * N]    = S[r + c * N] + b2 * kk;
      X[r + c * N]    = P[r + c * N] + c4 * kk;
    }
  if (a.A_end != nullptr) {  // ... and at t + dt (stage 4)
    wave_sync();
    tile_load<NN>(a.A_end, item0, a.batch, ldsA, lane);
    wave_sync();
  }
  rhs();  // k4
  wave_sync();
  if (live) {
#pragma unroll
    for (int c = 0; c < N; ++c)
#pragma unroll
      for (int r = 0; r < N; ++r) ldsS[lane * NP + r + c * N] = S[r + c * N] + b1 * kof(r, c);
  }
  wave_sync();
  tile_store<NN>(a.P, item0, a.batch, ldsS, lane);
}

// ---- generic sizes: ONE FILTER PER WAVEFRONT, matrices in LDS, lanes over matrix entries ----
// Any dof, ny <= kEkfMaxDim (the reference's EKF is a template over arbitrary Dof / Ny: its own tests use (10,3),
// (3,10) and Nx = 9, tests/test_ekf.cpp:93-103,148-153).  Same operations in the same order as the per-lane
// kernels and the oracle: every entry is a k-ascending fma chain; the innovation covariance is factorised by the
// wave-wide pivoted LDL' of the dense QP kernels (ldlt_wave.h: lane i owns row i).  mode: 0 no predict, 1 Euler,
// 2 runge_kutta4.
__global__ void __launch_bounds__(64) ekf_generic_kernel(const EkfArgs a, const int N, const int M, const int mode,
                                                         const int update)
{
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.x;
  const int NN = N * N, MN = M * N, BUF = NN > MN ? NN : MN;  // (ny may exceed dof)
  double *P = sm, *A = P + BUF, *Q = A + BUF, *K = Q + BUF, *X = K + BUF, *S = X + BUF;  // predict
  double *H = K, *T = X, *HP = S;                                                        // update reuses the stage buffers
  double *Xs = S + BUF, *W = Xs + kEkfMaxDim * kEkfMaxDim, *temp = W + kEkfMaxDim * (kEkfMaxDim + 1) / 2, *xch = temp + kEkfMaxDim;
  int *perm = reinterpret_cast<int *>(xch + kEkfMaxDim);
  double *IK = reinterpret_cast<double *>(perm + kEkfMaxDim);
  double *gP = a.P + b * NN;
  for (int e = lane; e < NN; e += kWave) P[e] = gP[e];
  wave_sync();
  if (mode != 0) {
    const double *gQ = a.q_shared ? a.Q : a.Q + b * NN;
    const double *gA = a.A + b * NN;
    for (int e = lane; e < NN; e += kWave) { A[e] = gA[e]; Q[e] = gQ[e]; }
    const double dt = a.dt_shared ? a.dt[0] : a.dt[b];
    wave_sync();
    // K = symU(A Z + Z A' + Q): upper entries computed, mirrored (ekf.hpp:88)
    auto rhs = [&](const double *Z) {
      for (int e = lane; e < NN; e += kWave) {
        const int i = e % N, j = e / N;
        if (i <= j) {
          double m1 = 0.0, m2 = 0.0;
          for (int k = 0; k < N; ++k) m1 = fma(A[i + k * N], Z[k + j * N], m1);
          for (int k = 0; k < N; ++k) m2 = fma(Z[i + k * N], A[j + k * N], m2);
          const double s = (m1 + m2) + Q[i + j * N];
          K[i + j * N]   = s;
          K[j + i * N]   = s;
        }
      }
      wave_sync();
    };
    auto reload_A = [&](const double *g) {
      if (g == nullptr) return;
      wave_sync();
      for (int e = lane; e < NN; e += kWave) A[e] = g[b * NN + e];
      wave_sync();
    };
    if (mode == 1) {
      rhs(P);
      for (int e = lane; e < NN; e += kWave) P[e] = P[e] + dt * K[e];
    } else {
      const double b1 = dt * (1.0 / 6.0), b2 = dt * (1.0 / 3.0), c2 = dt * 0.5, c4 = dt * 1.0;
      rhs(P);  // k1
      for (int e = lane; e < NN; e += kWave) { S[e] = P[e] + b1 * K[e]; X[e] = P[e] + c2 * K[e]; }
      reload_A(a.A_mid);
      wave_sync();
      rhs(X);  // k2
      for (int e = lane; e < NN; e += kWave) { S[e] = S[e] + b2 * K[e]; X[e] = P[e] + c2 * K[e]; }
      wave_sync();
      rhs(X);  // k3
      for (int e = lane; e < NN; e += kWave) { S[e] = S[e] + b2 * K[e]; X[e] = P[e] + c4 * K[e]; }
      reload_A(a.A_end);
      wave_sync();
      rhs(X);  // k4
      for (int e = lane; e < NN; e += kWave) P[e] = S[e] + b1 * K[e];
    }
    wave_sync();
  }
  if (update) {
    const double *gH = a.H + b * MN, *gR = a.r_shared ? a.R : a.R + b * (size_t)(M * M);
    for (int e = lane; e < MN; e += kWave) H[e] = gH[e];
    wave_sync();
    for (int e = lane; e < MN; e += kWave) {  // T = H symU(P), HP = H P   (ekf.hpp:129, :134)
      const int aa = e % M, j = e / M;
      double s1 = 0.0, s2 = 0.0;
      for (int k = 0; k < N; ++k) s1 = fma(H[aa + k * M], (k <= j) ? P[k + j * N] : P[j + k * N], s1);
      for (int k = 0; k < N; ++k) s2 = fma(H[aa + k * M], P[k + j * N], s2);
      T[e]  = s1;
      HP[e] = s2;
    }
    wave_sync();
    for (int e = lane; e < M * M; e += kWave) {  // S = triU(H Psym H' + R), stored as the packed lower triangle
      const int aa = e % M, bb = e / M;
      if (aa <= bb) {
        double s = 0.0;
        for (int k = 0; k < N; ++k) s = fma(T[aa + k * M], H[bb + k * M], s);
        W[tri(bb, aa)] = s + gR[aa + bb * M];
      }
    }
    wave_sync();
    const int ok = ldlt_factor_lds(M, W, perm, temp, lane);
    wave_sync();
    // X = S^-1 (H P): the N right-hand sides at once, lane j substitutes through column j on its own (in LDS: T is
    // free by now) -- per entry the same operations in the same order as ldlt_solve_lds / the oracle (forward j
    // ascending, |d| <= DBL_MIN -> 0, backward j descending), without N rounds of cross-lane broadcasts.  (With fewer
    // right-hand sides than rows the cross-lane form is quicker: (3, 10) 1.07 vs 1.36 ms per 262 144 filters.)
    if (N < M) {
      for (int j = 0; j < N; ++j) {
        const double v = ldlt_solve_lds(M, W, perm, xch, lane < M ? HP[lane + j * M] : 0.0, lane);
        if (lane < M) Xs[lane + j * M] = v;
      }
    } else if (lane < N) {
      double *x = T + lane * M;
      for (int i = 0; i < M; ++i) x[i] = HP[perm[i] + lane * M];
      for (int jj = 0; jj < M - 1; ++jj) {
        const double xj = x[jj];
        for (int i = jj + 1; i < M; ++i) x[i] = fma(-W[tri(i, jj)], xj, x[i]);
      }
      for (int i = 0; i < M; ++i) {
        const double d = W[tri(i, i)];
        x[i]           = (fabs(d) > DBL_MIN) ? x[i] / d : 0.0;
      }
      for (int jj = M - 1; jj > 0; --jj) {
        const double xj = x[jj];
        for (int i = 0; i < jj; ++i) x[i] = fma(-W[tri(jj, i)], xj, x[i]);
      }
      for (int i = 0; i < M; ++i) Xs[perm[i] + lane * M] = x[i];
    }
    wave_sync();
    if (lane < N) {  // delta = K r, K = X'   (:137)
      double s = 0.0;
      for (int aa = 0; aa < M; ++aa) s = fma(Xs[aa + lane * M], a.r[b * M + aa], s);
      a.delta[b * N + lane] = s;
    }
    if (a.info != nullptr && lane == 0) a.info[b] = ok ? 0 : 1;
    for (int e = lane; e < NN; e += kWave) {  // I - K H
      const int i = e % N, k = e / N;
      double s = 0.0;
      for (int aa = 0; aa < M; ++aa) s = fma(Xs[aa + i * M], H[aa + k * M], s);
      IK[e] = ((i == k) ? 1.0 : 0.0) - s;
    }
    wave_sync();
    double *Pn = A;  // P = symU((I - K H) P)   (:138)
    for (int e = lane; e < NN; e += kWave) {
      const int i = e % N, j = e / N;
      if (i <= j) {
        double s = 0.0;
        for (int k = 0; k < N; ++k) s = fma(IK[i + k * N], P[k + j * N], s);
        Pn[i + j * N] = s;
        Pn[j + i * N] = s;
      }
    }
    wave_sync();
    for (int e = lane; e < NN; e += kWave) gP[e] = Pn[e];
    return;
  }
  for (int e = lane; e < NN; e += kWave) gP[e] = P[e];
}

static hipError_t ekf_generic_launch(const EkfArgs &a, int dof, int ny, int mode, bool update, hipStream_t stream)
{
  if (a.batch > 0x7FFFFFFFll) return hipErrorInvalidValue;
  constexpr int D = kEkfMaxDim;
  const int buf    = std::max(dof * dof, dof * ny);
  const size_t lds = (size_t)(6 * buf + D * D + D * (D + 1) / 2 + 2 * D + D / 2 + 1 + dof * dof) * sizeof(double);
  hipLaunchKernelGGL(ekf_generic_kernel, dim3((unsigned)a.batch), dim3(kWave), lds, stream, a, dof, ny, mode, update ? 1 : 0);
  return hipGetLastError();
}

static bool ekf_fast(int dof, int ny, bool update)
{
  const bool nok = dof == 2 || dof == 3 || dof == 4 || dof == 6 || dof == 7;  // (7: the last size whose fused step fits the registers)
  if (dof == 8 && !update) return true;  // predict alone still fits at dof 8 (0.11 ms for 262 144 filters; generic: 0.2); at 9 and 10 it spills and loses
  if (update && ((dof == 6 && ny == 6) || (dof == 4 && ny == 4))) return true;  // full-state measurements
  return update ? (nok && ny >= 1 && ny <= 3) : nok;
}

hipError_t ekf_rk4_launch(const EkfArgs &a, int dof, hipStream_t stream)
{
  const dim3 grid((unsigned)((a.batch + kWave - 1) / kWave)), block(kWave);
  switch (dof) {
    case 2: hipLaunchKernelGGL((ekf_rk4_kernel<2>), grid, block, 0, stream, a); break;
    case 3: hipLaunchKernelGGL((ekf_rk4_kernel<3>), grid, block, 0, stream, a); break;
    case 4: hipLaunchKernelGGL((ekf_rk4_kernel<4>), grid, block, 0, stream, a); break;
    case 6: hipLaunchKernelGGL((ekf_rk4_kernel<6>), grid, block, 0, stream, a); break;
    default: return ekf_generic_launch(a, dof, 1, 2, false, stream);
  }
  return hipGetLastError();
}

template<int N, int M>
static hipError_t launch_nm(const EkfArgs &a, bool predict, bool update, hipStream_t stream)
{
  const dim3 grid((unsigned)((a.batch + kWave - 1) / kWave)), block(kWave);
  if constexpr (N == 6 && M <= 3) {  // (the pair exchange of the kernel is what an N = 6 tile needs; M N + M M <= N N)
    // the persistent form of the fused step (see the kernel): as many waves as the device holds at once -- its LDS admits four
    // per compute unit -- when there are tiles for several rounds of them; SFB_EKF_PERSISTENT=0: the one-tile-per-wave kernel
    const char *pk = sfb::knob("SFB_EKF_PERSISTENT");
    if (predict && update && !(pk && atoi(pk) == 0)) {
      static std::mutex mu;
      static std::map<int, int> resident;  // device -> waves of this kernel it holds
      int dev = 0, waves = 0;
      if (hipGetDevice(&dev) == hipSuccess) {
        std::lock_guard<std::mutex> lk(mu);
        const auto it = resident.find(dev);
        if (it != resident.end()) waves = it->second;
        else {
          int per_cu = 0, cus = 0;
          if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ekf_fused_persistent_kernel<N, M>, kWave, 0) == hipSuccess &&
              hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
            waves = per_cu * cus;
          else (void)hipGetLastError();
          resident[dev] = waves;
        }
      } else (void)hipGetLastError();
      const int64_t ntiles = (a.batch + kWave - 1) / kWave;
      if (waves > 0 && ntiles >= 4 * (int64_t)waves) {
        hipLaunchKernelGGL((ekf_fused_persistent_kernel<N, M>), dim3((unsigned)waves), block, 0, stream, a, ntiles);
        return hipGetLastError();
      }
    }
  }
  if (predict && update) hipLaunchKernelGGL((ekf_kernel<N, M, true, true>), grid, block, 0, stream, a);
  else if (predict) hipLaunchKernelGGL((ekf_kernel<N, M, true, false>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((ekf_kernel<N, M, false, true>), grid, block, 0, stream, a);
  return hipGetLastError();
}

bool ekf_supported(int dof, int ny, bool update)
{
  return dof >= 1 && dof <= kEkfMaxDim && (!update || (ny >= 1 && ny <= kEkfMaxDim));
}

hipError_t ekf_launch(const EkfArgs &a, int dof, int ny, bool predict, bool update, hipStream_t stream)
{
  if (update && dof >= 8 && dof <= 10 && ny >= 1 && ny <= 3) {
    // predict (if any) by the generic kernel, then the per-lane update: two launches on the stream, P through HBM in
    // between -- 2.6 x faster than the generic kernel's fused step at (9, 3), same bits
    if (predict) {
      const hipError_t e = ekf_launch(a, dof, 1, true, false, stream);
      if (e != hipSuccess) return e;
    }
    const dim3 grid((unsigned)((a.batch + kWave - 1) / kWave)), block(kWave);
#define SFB_EKF_WIDE(N, M) \
  if (dof == N && ny == M) hipLaunchKernelGGL((ekf_update_wide_kernel<N, M>), grid, block, 0, stream, a);
    SFB_EKF_WIDE(8, 1) SFB_EKF_WIDE(8, 2) SFB_EKF_WIDE(8, 3)
    SFB_EKF_WIDE(9, 1) SFB_EKF_WIDE(9, 2) SFB_EKF_WIDE(9, 3) SFB_EKF_WIDE(10, 1) SFB_EKF_WIDE(10, 2) SFB_EKF_WIDE(10, 3)
#undef SFB_EKF_WIDE
    return hipGetLastError();
  }
  if (!update) ny = 1;
  if (!ekf_fast(dof, ny, update)) return ekf_generic_launch(a, dof, ny, predict ? 1 : 0, update, stream);
#define SFB_EKF_CASE(N, M) \
  if (dof == N && ny == M) return launch_nm<N, M>(a, predict, update, stream);
  SFB_EKF_CASE(2, 1) SFB_EKF_CASE(2, 2) SFB_EKF_CASE(2, 3)
  SFB_EKF_CASE(3, 1) SFB_EKF_CASE(3, 2) SFB_EKF_CASE(3, 3)
  SFB_EKF_CASE(4, 1) SFB_EKF_CASE(4, 2) SFB_EKF_CASE(4, 3)
  SFB_EKF_CASE(6, 1) SFB_EKF_CASE(6, 2) SFB_EKF_CASE(6, 3) SFB_EKF_CASE(6, 6) SFB_EKF_CASE(4, 4)
  SFB_EKF_CASE(7, 1) SFB_EKF_CASE(7, 2) SFB_EKF_CASE(7, 3) SFB_EKF_CASE(8, 1)
#undef SFB_EKF_CASE
  return hipErrorInvalidValue;
}

}  // namespace sfb
