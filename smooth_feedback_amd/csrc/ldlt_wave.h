// Pivoted LDL' of a small symmetric matrix held in LDS, one wavefront, lane i owns row i (k <= 64).
//
// Arithmetic follows Eigen 3.4's LDLT<.,Upper> (the factorisation smooth_feedback uses at
// qp_solver.hpp:259,428 and :187-188): left-looking, largest-|diagonal| symmetric pivoting on the
// NOT-yet-updated trailing diagonal, dot-product-then-subtract updates, true divisions.
// Accumulation order: s = 0; j ascending; s = fma(L(i,j), temp(j), s)  -- identical to
// oracle/qp_oracle.c so that both produce the same bits.
#pragma once
#include <cfloat>

#include "wave_util.h"

namespace sfb {

__device__ __forceinline__ int tri(const int i, const int j) { return ((i * (i + 1)) >> 1) + j; }

// W: PACKED lower triangle, W(i,j) (j <= i) at tri(i,j) = i(i+1)/2 + j.  Triangular numbers are a
// complete residue system mod 2^k, so a column walk (lane i reads W(i,j)) is bank-conflict-free, and
// the factor of a k=30 problem takes 3.7 KB of LDS instead of 7.4 KB (more waves per CU).
// perm[k] (LDS): composed row permutation, (P b)[i] = b[perm[i]].  temp[k] (LDS) scratch.
// Returns 1 on success, 0 on failure (Eigen info()==NumericalIssue).  Wave-uniform.
__device__ inline int ldlt_factor_lds(const int k, double *W, int *perm, double *temp, const int lane)
{
  if (lane < k) perm[lane] = lane;
  wave_lds_fence();
  if (k <= 1) return 1;

  int ret = 1, found_zero_pivot = 0;
  const bool inmat = lane < k;

  for (int kk = 0; kk < k; ++kk) {
    // pivot: first index of the largest |diag| among rows kk..k-1
    const double dg = inmat ? W[tri(lane, lane)] : 0.0;
    const bool cand = inmat && lane >= kk;
    const double a  = cand ? fabs(dg) : -1.0;
    const double mx = wave_max(a);
    unsigned long long bal = wave_ballot(cand && a == mx);
    // Eigen's maxCoeff visitor starts from the first candidate and replaces it only by a strictly greater value: NaNs
    // further down are skipped (fmax / the equality test above do the same), but a NaN AT kk stays the "maximum" --
    // the pivot is then kk itself (and, being invalid, fails the factorisation unless the column below it is zero)
    const double dkk = lane_bcast(dg, kk);
    const int p      = (bal && !(dkk != dkk)) ? (int)__builtin_ctzll(bal) : kk;

    if (p != kk) {
      if (lane == 0) {
        const int t = perm[kk];
        perm[kk]    = perm[p];
        perm[p]     = t;
      }
      if (lane < kk) {  // row(kk).head(kk) <-> row(p).head(kk)
        const double a1   = W[tri(kk, lane)];
        const double a2   = W[tri(p, lane)];
        W[tri(kk, lane)] = a2;
        W[tri(p, lane)]  = a1;
      } else if (lane == kk) {  // diagonal entries
        const double a1 = W[tri(kk, kk)];
        const double a2 = W[tri(p, p)];
        W[tri(kk, kk)] = a2;
        W[tri(p, p)]   = a1;
      } else if (lane < p) {  // kk < i < p : W(i,kk) <-> W(p,i)
        const double a1   = W[tri(lane, kk)];
        const double a2   = W[tri(p, lane)];
        W[tri(lane, kk)] = a2;
        W[tri(p, lane)]  = a1;
      } else if (lane > p && inmat) {  // col(kk).tail <-> col(p).tail
        const double a1   = W[tri(lane, kk)];
        const double a2   = W[tri(lane, p)];
        W[tri(lane, kk)] = a2;
        W[tri(lane, p)]  = a1;
      }
      wave_lds_fence();
    }

    // temp(j) = D(j) * L(kk,j), j < kk
    if (lane < kk) temp[lane] = W[tri(lane, lane)] * W[tri(kk, lane)];
    wave_lds_fence();

    double val = 0.0;
    if (cand) {
      double s = 0.0;
      for (int j = 0; j < kk; ++j) s = fma(W[tri(lane, j)], temp[j], s);
      val = W[tri(lane, kk)];
      if (kk > 0) val -= s;
    }
    const double akk = lane_bcast(val, kk);
    const bool valid = fabs(akk) > 0.0;

    if (kk == 0 && !valid) {
      // whole diagonal is zero: success iff the strictly lower triangle is zero (perm = identity)
      bool nz = false;
      if (inmat)
        for (int j = 0; j < lane; ++j) nz = nz || !(W[tri(lane, j)] == 0.0);
      return wave_ballot(nz) ? 0 : 1;
    }

    if (lane == kk) W[tri(kk, kk)] = val;
    if (cand && lane > kk) {
      if (valid) val = val / akk;
      W[tri(lane, kk)] = val;
    }
    if (!valid) {
      if (wave_ballot(cand && lane > kk && !(val == 0.0))) ret = 0;
    }
    if (found_zero_pivot && valid) {
      ret = 0;
    } else if (!valid) {
      found_zero_pivot = 1;
    }
    wave_lds_fence();
  }
  return ret;
}

// Generic (runtime k) solve with the factor in LDS:  P b -> L^-1 -> D^-1 -> L^-T -> P^T.
// `b` is lane i's entry of the right-hand side in ORIGINAL order; returns lane i's entry of the
// solution in original order.  xch[k] (LDS) scratch.  Same accumulation order as the oracle:
// forward j ascending, backward j descending.
__device__ inline double ldlt_solve_lds(const int k, const double *W, const int *perm, double *xch,
                                        double b, const int lane)
{
  const bool inmat = lane < k;
  if (inmat) xch[lane] = b;
  wave_lds_fence();
  double x = inmat ? xch[perm[lane]] : 0.0;
  wave_lds_fence();
  for (int j = 0; j < k - 1; ++j) {
    const double xj = lane_bcast(x, j);
    if (inmat && lane > j) x = fma(-W[tri(lane, j)], xj, x);
  }
  if (inmat) {
    const double d = W[tri(lane, lane)];
    x              = (fabs(d) > DBL_MIN) ? x / d : 0.0;
  }
  for (int j = k - 1; j > 0; --j) {
    const double xj = lane_bcast(x, j);
    if (lane < j) x = fma(-W[tri(j, lane)], xj, x);
  }
  if (inmat) xch[perm[lane]] = x;
  wave_lds_fence();
  const double r = inmat ? xch[lane] : 0.0;
  wave_lds_fence();
  return r;
}

}  // namespace sfb
