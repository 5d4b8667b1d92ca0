// Per-lane building blocks of the batched EKF covariance kernels (one filter per lane, every small matrix in registers)
// and the coalesced tile I/O around them.  Shared by the library's kernels (smooth_feedback_amd/csrc/ekf.hip) and by
// kernels compiled in the caller's translation unit (ekf_device.hpp: a swarm whose dynamics / measurement functors are
// device-callable runs linearisation, covariance step, state step and g (+) delta in ONE launch around these).
//
// Arithmetic == oracle/ekf_oracle.c operation for operation (k-ascending fma chains, Eigen 3.4's pivoted LDL'):
//   ekf_lane_predict   ekf.hpp:84-89 + the Euler step of :96     P <- P + dt * symU(A P + P A' + Q)
//   ekf_lane_update    ekf.hpp:119-138   S = triU(H symU(P) H' + R); K = (ldlt(symU(S)).solve(H P))'; delta = K r;
//                                        P <- symU((I - K H) P)
// HIP only (device code).
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdint>

namespace sfb {
namespace ekf_lane {

constexpr int kLanes = 64;

typedef double vd2 __attribute__((ext_vector_type(2)));

// ---- coalesced tile I/O: 64 items x W doubles, contiguous in HBM, one item per lane in LDS ----
template<int W>
__device__ __forceinline__ void tile_load(const double *__restrict__ g, int64_t item0, int64_t nitems, double *lds,
                                          int lane)
{
  constexpr int WP = W | 1;  // odd stride: conflict-free per-lane walks
  const int total  = (int)(nitems - item0 < kLanes ? nitems - item0 : kLanes) * W;
  const double *src = g + item0 * W;
  // all global loads first (independent, one wait), then the LDS scatter
  if constexpr (W % 2 == 0) {
    constexpr int W2 = W / 2;  // 16-byte loads; a pair never straddles two items
    // (non-temporal: every byte of the batch is touched exactly once per launch)
    const vd2 *src2 = reinterpret_cast<const vd2 *>(src);
    vd2 v[W2];
#pragma unroll
    for (int c = 0; c < W2; ++c) {
      const int idx = c * kLanes + lane;
      v[c]          = (2 * idx < total) ? __builtin_nontemporal_load(&src2[idx]) : vd2{0.0, 0.0};
    }
#pragma unroll
    for (int c = 0; c < W2; ++c) {
      const int idx = 2 * (c * kLanes + lane);
      const int o   = (idx / W) * WP + (idx % W);
      lds[o]        = v[c].x;
      lds[o + 1]    = v[c].y;
    }
  } else {
    double v[W];
#pragma unroll
    for (int c = 0; c < W; ++c) {
      const int idx = c * kLanes + lane;
      v[c]          = (idx < total) ? __builtin_nontemporal_load(&src[idx]) : 0.0;
    }
#pragma unroll
    for (int c = 0; c < W; ++c) {
      const int idx = c * kLanes + lane;
      lds[(idx / W) * WP + (idx % W)] = v[c];
    }
  }
}
template<int W>
__device__ __forceinline__ void tile_store(double *__restrict__ g, int64_t item0, int64_t nitems, const double *lds,
                                           int lane)
{
  constexpr int WP = W | 1;
  const int total  = (int)(nitems - item0 < kLanes ? nitems - item0 : kLanes) * W;
  double *dst      = g + item0 * W;
  if constexpr (W % 2 == 0) {
    constexpr int W2 = W / 2;
    vd2 *dst2        = reinterpret_cast<vd2 *>(dst);
#pragma unroll
    for (int c = 0; c < W2; ++c) {
      const int i2  = c * kLanes + lane;
      const int idx = 2 * i2;
      const int o   = (idx / W) * WP + (idx % W);
      if (idx < total) __builtin_nontemporal_store(vd2{lds[o], lds[o + 1]}, &dst2[i2]);
    }
  } else {
#pragma unroll
    for (int c = 0; c < W; ++c) {
      const int idx = c * kLanes + lane;
      if (idx < total) __builtin_nontemporal_store(lds[(idx / W) * WP + (idx % W)], &dst[idx]);
    }
  }
}

// ---- pivoted LDL' of a tiny symmetric matrix, fully unrolled (static register indices) ----
// W lower (row-major W[i][j], j <= i).  Same algorithm/ordering as oracle_ldlt_factor.
template<int M>
struct SmallLdlt {
  double W[M][M];
  int tr[M];
  bool ok;

  __device__ __forceinline__ void swap_rc(int kk, int p)  // symmetric swap kk <-> p (p > kk), static loops
  {
#pragma unroll
    for (int pc = 1; pc < M; ++pc) {
      if (pc == p) {
#pragma unroll
        for (int kc = 0; kc < M - 1; ++kc) {
          if (kc == kk && kc < pc) {
#pragma unroll
            for (int t = 0; t < M; ++t)
              if (t < kc) { const double a = W[kc][t]; W[kc][t] = W[pc][t]; W[pc][t] = a; }
#pragma unroll
            for (int i = 0; i < M; ++i)
              if (i > pc) { const double a = W[i][kc]; W[i][kc] = W[i][pc]; W[i][pc] = a; }
            { const double a = W[kc][kc]; W[kc][kc] = W[pc][pc]; W[pc][pc] = a; }
#pragma unroll
            for (int i = 0; i < M; ++i)
              if (i > kc && i < pc) { const double a = W[i][kc]; W[i][kc] = W[pc][i]; W[pc][i] = a; }
          }
        }
      }
    }
  }

  __device__ __forceinline__ void factor()
  {
    ok = true;
    if constexpr (M == 1) {
      tr[0] = 0;
      return;
    }
    bool found_zero = false, finished = false;
    double temp[M];
#pragma unroll
    for (int kk = 0; kk < M; ++kk) {
      if (!finished) {
        int p       = kk;
        double best = fabs(W[kk][kk]);
#pragma unroll
        for (int i = kk + 1; i < M; ++i) {
          const double a = fabs(W[i][i]);
          if (a > best) { best = a; p = i; }
        }
        tr[kk] = p;
        if (p != kk) swap_rc(kk, p);
        if (kk > 0) {
#pragma unroll
          for (int j = 0; j < kk; ++j) temp[j] = W[j][j] * W[kk][j];
          double s = 0.0;
#pragma unroll
          for (int j = 0; j < kk; ++j) s = fma(W[kk][j], temp[j], s);
          W[kk][kk] -= s;
#pragma unroll
          for (int i = kk + 1; i < M; ++i) {
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < kk; ++j) t = fma(W[i][j], temp[j], t);
            W[i][kk] -= t;
          }
        }
        const double akk = W[kk][kk];
        const bool valid = fabs(akk) > 0.0;
        if (kk == 0 && !valid) {
#pragma unroll
          for (int j = 0; j < M; ++j) {
            tr[j] = j;
#pragma unroll
            for (int i = j + 1; i < M; ++i) ok = ok && (W[i][j] == 0.0);
          }
          finished = true;
        } else {
          if (valid) {
#pragma unroll
            for (int i = kk + 1; i < M; ++i) W[i][kk] /= akk;
          } else {
#pragma unroll
            for (int i = kk + 1; i < M; ++i) ok = ok && (W[i][kk] == 0.0);
          }
          if (found_zero && valid) ok = false;
          else if (!valid) found_zero = true;
        }
      }
    }
  }

  __device__ __forceinline__ void solve(double (&b)[M]) const  // P b, L^-1, D^-1 (|d|<=DBL_MIN -> 0), L^-T, P^T
  {
#pragma unroll
    for (int i = 0; i < M; ++i) {
#pragma unroll
      for (int pc = 0; pc < M; ++pc)
        if (pc > i && tr[i] == pc) { const double a = b[i]; b[i] = b[pc]; b[pc] = a; }
    }
#pragma unroll
    for (int i = 0; i < M; ++i) {
      double s = b[i];
#pragma unroll
      for (int j = 0; j < i; ++j) s = fma(-W[i][j], b[j], s);
      b[i] = s;
    }
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const double d = W[i][i];
      b[i]           = (fabs(d) > DBL_MIN) ? b[i] / d : 0.0;
    }
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {
      double s = b[i];
#pragma unroll
      for (int j = M - 1; j > i; --j) s = fma(-W[j][i], b[j], s);
      b[i] = s;
    }
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {
#pragma unroll
      for (int pc = 0; pc < M; ++pc)
        if (pc > i && tr[i] == pc) { const double a = b[i]; b[i] = b[pc]; b[pc] = a; }
    }
  }
};


// P <- P + dt * symU(A P + P A' + Q): all matrices column-major N x N in registers; q(i, j) supplies Q (upper part used)
template<int N, class QF>
__device__ __forceinline__ void ekf_lane_predict(double (&P)[N * N], const double (&A)[N * N], QF &&q, const double dt)
{
  double Pn[N * N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
#pragma unroll
    for (int i = 0; i <= j; ++i) {
      double m1 = 0.0, m2 = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k) m1 = fma(A[i + k * N], P[k + j * N], m1);
#pragma unroll
      for (int k = 0; k < N; ++k) m2 = fma(P[i + k * N], A[j + k * N], m2);
      const double s = (m1 + m2) + q(i, j);  // ekf.hpp:88, upper triangle mirrored
      Pn[i + j * N]  = P[i + j * N] + dt * s;
      if (i != j) Pn[j + i * N] = P[j + i * N] + dt * s;
    }
  }
#pragma unroll
  for (int e = 0; e < N * N; ++e) P[e] = Pn[e];
}

// Kalman update of one filter: H (M x N), R via rr(a, b) for a <= b (upper part), innovation rv; writes delta (N) and the
// new covariance into P; returns whether the LDL' of the innovation covariance succeeded (the reference does not check)
template<int N, int M, class RF>
__device__ __forceinline__ bool ekf_lane_update(double (&P)[N * N], const double (&H)[M * N], RF &&rr, const double (&rv)[M],
                                                double (&delta)[N])
{
  constexpr int MN = M * N, NN = N * N;
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
#pragma unroll
  for (int b = 0; b < M; ++b) {
#pragma unroll
    for (int aa = 0; aa < M; ++aa) {
      if (aa <= b) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < N; ++k) s = fma(T[aa + k * M], H[b + k * M], s);
        F.W[b][aa] = s + rr(aa, b);
      } else {
        F.W[b][aa] = 0.0;
      }
    }
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
#pragma unroll
  for (int i = 0; i < N; ++i) {  // delta = K r, K = X'   (:137)
    double s = 0.0;
#pragma unroll
    for (int aa = 0; aa < M; ++aa) s = fma(X[aa + i * M], rv[aa], s);
    delta[i] = s;
  }
  double IK[NN];  // P = symU((I - K H) P)   (:138)
#pragma unroll
  for (int k = 0; k < N; ++k) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double s = 0.0;
#pragma unroll
      for (int aa = 0; aa < M; ++aa) s = fma(X[aa + i * M], H[aa + k * M], s);
      IK[i + k * N] = ((i == k) ? 1.0 : 0.0) - s;
    }
  }
  double Pn[NN];
#pragma unroll
  for (int j = 0; j < N; ++j) {
#pragma unroll
    for (int i = 0; i <= j; ++i) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k) s = fma(IK[i + k * N], P[k + j * N], s);
      Pn[i + j * N] = s;
      Pn[j + i * N] = s;
    }
  }
#pragma unroll
  for (int e = 0; e < NN; ++e) P[e] = Pn[e];
  return F.ok;
}

}  // namespace ekf_lane
}  // namespace sfb
