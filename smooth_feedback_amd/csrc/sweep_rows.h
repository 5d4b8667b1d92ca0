// Triangular sweeps (L D L')^-1 of ONE QP per wavefront over a PACKED, LDS-resident factor, k <= 128, in 16 x 16 BLOCKS
// with the pivot broadcast fused into the FP64 FMA (v_fmac_f64_dpp row_newbcast) -- the engine of the on-chip dense
// kernels for 32 < n+m <= 128 (qp_dense_big.hip packed engine; reference qp_solver.hpp:462 ldlt.solveInPlace(p),
// :192-195 in polish_qp; Eigen LDLT::_solve_impl = oracle/qp_oracle.c oracle_ldlt_solve).
//
// Why blocks: a sweep is k dependent steps [x_j final -> every later row subtracts L(i, j) x_j].  With the pivot going
// through v_readlane -> SGPR -> v_fma_f64 a step costs a lone wave 49-69 cycles; a DPP row_newbcast FMA reads the pivot
// from a neighbour lane of the same 16-lane row in the FMA itself: 18 cycles per dependent step.  DPP only reaches inside
// a row, so the wave is used as FOUR rows of 16 lanes:
//   - lane = 16 r + cc carries rows lane and lane + 64 of the permuted system, i.e. 16-row blocks r and r + 4 (same
//     mapping as the engine this replaces);
//   - the pivot block's segment of the vector is REPLICATED in all four rows (tp): the in-block chain
//     tp(cc) -= L(16 jb + cc, 16 jb + J) tp(J), cc > J, runs in every row at once (one instruction, same cost), so right
//     after step J - 1 every row has x_J at hand and the updates of the other blocks ("riders": one v_fmac_f64_dpp per
//     group of blocks, row_mask selecting the rows whose block lies beyond the pivot block) ride in the chain's wait
//     states instead of following it;
//   - between blocks the next pivot segment travels from its owner row to all rows by v_permlane16_swap /
//     v_permlane32_swap (no LDS round trip).
// Per row the operations and their order are the oracle's: row i subtracts L(i, j) x_j for j ascending (forward),
// L(j, i) x_j for j descending (backward), each one fma(-L, x, acc).  Lanes a step does not concern are switched off
// through EXEC (chain) or row_mask (riders): no 0 * pivot products, non-finite data included.
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <type_traits>
#include <utility>

#include "wave_util.h"

namespace sfb {
namespace rows {

using lds_d = __attribute__((address_space(3))) double;

template<int I> using ic = std::integral_constant<int, I>;
template<int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
  [&]<int... I>(std::integer_sequence<int, I...>) { (f(ic<I>{}), ...); }(std::make_integer_sequence<int, N>{});
}

struct Pair { double lo, hi; };  // rows lane and lane + 64 of a vector in the permuted order of the factorisation

// lanes cc >= J of every 16-lane row
__host__ __device__ constexpr unsigned long long mask_ge(const int J)
{
  const unsigned long long m16 = (0xFFFFull << J) & 0xFFFFull;
  return m16 | (m16 << 16) | (m16 << 32) | (m16 << 48);
}
// rows (bit r = lane row r) whose block 4 q + r lies beyond / before the pivot block JB
template<int NB>
__host__ __device__ constexpr int fwd_rows(const int JB, const int q)
{
  int rm = 0;
  for (int r = 0; r < 4; ++r)
    if (4 * q + r > JB && 4 * q + r < NB) rm |= 1 << r;
  return rm;
}
__host__ __device__ constexpr int bwd_rows(const int JB, const int q)
{
  int rm = 0;
  for (int r = 0; r < 4; ++r)
    if (4 * q + r < JB) rm |= 1 << r;
  return rm;
}

// value of lane row RP in every row (v_permlane16_swap + v_permlane32_swap; wait states inside)
template<int RP>
__device__ __forceinline__ double row_to_all(double t)
{
  cross_lane_fence(t);  // the producer may be a hand-written VALU op the hazard recogniser does not see
  double ev, od, lo2, hi2;
  row_swap16(t, ev, od);  // ev = [r0 r0 r2 r2], od = [r1 r1 r3 r3]
  double e = (RP & 1) ? od : ev;
  half_swap32(e, lo2, hi2);  // lo2 = [lo lo], hi2 = [hi hi]
  return (RP & 2) ? hi2 : lo2;
}

// Chain step of the forward sweep: tp(cc) = fma(-l, tp(J), tp(cc)) on the lanes cc >= J of every row (m = mask_ge(J)).
// The pivot lane J has to stay enabled -- a DPP read of a lane that EXEC switches off disables the write of its readers --
// and meets the diagonal slot of the packed triangle, which the factorisation leaves at -0.0: fma(+0.0, x, x) == x for
// every finite x, either zero included (the convention of the k <= 32 kernel, qp_dense4.hip struct Factor).
template<int J>
__device__ __forceinline__ void chain_ge(double &tp, const double l, const unsigned long long m)
{
  if constexpr (J == 0)
    asm volatile("v_fmac_f64_dpp %0, %0, -%1 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "+v"(tp) : "v"(l));
  else
    asm volatile("s_mov_b64 exec, %2\n\t"
                 "v_fmac_f64_dpp %0, %0, -%1 row_newbcast:%3 row_mask:0xf bank_mask:0xf\n\t"
                 "s_mov_b64 exec, -1"
                 : "+v"(tp)
                 : "v"(l), "s"(m), "n"(J));
}
// chain step of the backward sweep: the lanes cc <= J (m = mask_ge(J + 1): its complement)
template<int J>
__device__ __forceinline__ void chain_le(double &tp, const double l, const unsigned long long m)
{
  if constexpr (J == 15)
    asm volatile("v_fmac_f64_dpp %0, %0, -%1 row_newbcast:15 row_mask:0xf bank_mask:0xf" : "+v"(tp) : "v"(l));
  else
    asm volatile("s_not_b64 exec, %2\n\t"
                 "v_fmac_f64_dpp %0, %0, -%1 row_newbcast:%3 row_mask:0xf bank_mask:0xf\n\t"
                 "s_mov_b64 exec, -1"
                 : "+v"(tp)
                 : "v"(l), "s"(m), "n"(J)
                 : "scc");
}
// rider: t(cc) = fma(-l, tp(J), t(cc)) on the rows of RM.  It reads lane J of tp, which the chain step issued just
// before it leaves as it is (see above), so it needs no wait states of its own.
template<int J, int RM>
__device__ __forceinline__ void rider(double &t, const double &tp, const double l)
{
  asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:%4 bank_mask:0xf" : "+v"(t) : "v"(tp), "v"(l), "n"(J), "n"(RM));
}

// (L D L')^-1 applied to the PERMUTED vector t held in registers (rows lane, lane + 64): L^-1, D^-1, L^-T -- the middle
// of oracle_ldlt_solve.  T: the packed lower triangle in LDS, entry (i, j), j < i, at T[i (i + 1) / 2 + j], its diagonal
// slots at -0.0 (chain_ge); Dg: the diagonal D.  NB = ceil(K / 16) blocks, compile time; everything is unrolled over compile-time block and step numbers
// (pivot lane, LDS offsets, lane and row masks are immediates).  The operands of a half block (8 steps) are fetched
// from LDS while the previous half block runs.
template<int NB>
__device__ __attribute__((noinline)) Pair row_sweeps(const int K_, const double *T_, const double *Dg_, Pair t, const int lane)
{
  static_assert(NB >= 1 && NB <= 8, "k <= 128");
  constexpr int NQ = NB > 4 ? 2 : 1;
  const int K      = __builtin_amdgcn_readfirstlane(K_);  // (an outlined function receives its arguments in VGPRs)
  const lds_d *const T  = (const lds_d *)T_;
  const lds_d *const Dg = (const lds_d *)Dg_;
  const int cc = lane & 15, r = lane >> 4;
  const bool vlo = lane < K, vhi = NQ > 1 && lane + kWave < K;
  const int ilo = lane, ihi = lane + kWave;
  double tlo = vlo ? t.lo : 0.0, thi = vhi ? t.hi : 0.0;
  // rider operands: forward L(row, col) = prow[col], backward L(row, col of this lane) = pcol[tri(row, 0)].  A row that
  // does not exist reads row 0's region (valid LDS; its value is never used: no pivot, no store).
  const lds_d *const plo = T + (vlo ? (ilo * (ilo + 1)) >> 1 : 0);
  const lds_d *const phi = T + (vhi ? (ihi * (ihi + 1)) >> 1 : 0);
  const lds_d *const cl0 = T + (vlo ? ilo : cc);
  const lds_d *const cl1 = T + (vhi ? ihi : cc);
  // chain operands: forward L(16 jb + cc, 16 jb + J) = pch[jb][J]; backward L(16 jb + J, 16 jb + cc) = pcc[tri(16 jb + J, 0) + 16 jb]
  const lds_d *pch[NB];
  static_for<NB>([&]<int JB>(ic<JB>) {
    const int row = 16 * JB + cc;
    pch[JB]       = T + ((row < K) ? ((row * (row + 1)) >> 1) + 16 * JB : 0);
  });
  const lds_d *const pcc = T + cc;
  // the lane masks of the chain steps, resident in SGPR pairs for the whole call
  unsigned long long mk[16];  // mk[J] = lanes cc >= J, J = 1 .. 15
  mk[0] = ~0ull;
  static_for<15>([&]<int J>(ic<J>) {
    mk[J + 1] = mask_ge(J + 1);
    asm volatile("" : "+s"(mk[J + 1]));
  });
  const unsigned owner = (unsigned)r;
  double tp = 0.0;
  double ch[2][8], f0[2][8], f1[2][8];

  // ---------------- forward: blocks 0 .. NB-1, steps J = 0 .. 15 (stage S = 2 JB + half) ----------------
  auto fload = [&]<int S>(ic<S>) {
    constexpr int JB = S >> 1, C = S & 1;
    static_for<8>([&]<int U>(ic<U>) {
      constexpr int J = 8 * C + U;
      if constexpr (J < 15) ch[C][U] = pch[JB][J];
      if constexpr (fwd_rows<NB>(JB, 0) != 0) f0[C][U] = plo[16 * JB + J];
      if constexpr (NQ > 1 && fwd_rows<NB>(JB, 1) != 0) f1[C][U] = phi[16 * JB + J];
    });
  };
  auto frun = [&]<int S>(ic<S>) {
    constexpr int JB = S >> 1, C = S & 1, RP = JB & 3, QP = JB >> 2;
    constexpr int RM0 = fwd_rows<NB>(JB, 0), RM1 = NQ > 1 ? fwd_rows<NB>(JB, 1) : 0;
    if constexpr (C == 0) tp = row_to_all<RP>(QP ? thi : tlo);
    if (JB < NB - 1 || 16 * JB + 8 * C < K) {  // (the last block beyond K: rows that do not exist)
      static_for<8>([&]<int U>(ic<U>) {
        constexpr int J = 8 * C + U;
        if constexpr (J < 15) chain_ge<J>(tp, ch[C][U], mk[J]);
        if constexpr (RM0 != 0) rider<J, RM0>(tlo, tp, f0[C][U]);
        if constexpr (RM1 != 0) rider<J, RM1>(thi, tp, f1[C][U]);
      });
    }
    if constexpr (C == 1) {  // the finished segment back to its owner row
      if constexpr (QP) thi = (owner == (unsigned)RP) ? tp : thi;
      else tlo = (owner == (unsigned)RP) ? tp : tlo;
    }
  };
  fload(ic<0>{});
  static_for<2 * NB>([&]<int S>(ic<S>) {
    if constexpr (S + 1 < 2 * NB) fload(ic<S + 1>{});
    frun(ic<S>{});
  });

  // ---------------- backward operands of the first stage, then D^-1 (|d| <= DBL_MIN -> 0, true division) ----------------
  auto bload = [&]<int S>(ic<S>) {  // stage S: block JB = NB-1 - S/2, steps J = 15 - 8 C - U
    constexpr int JB = NB - 1 - (S >> 1), C = S & 1;
    static_for<8>([&]<int U>(ic<U>) {
      constexpr int J = 15 - 8 * C - U, RW = 16 * JB + J, RO = (RW * (RW + 1)) / 2;
      if constexpr (J >= 1) ch[C][U] = pcc[RO + 16 * JB];
      if constexpr (bwd_rows(JB, 0) != 0) f0[C][U] = cl0[RO];
      if constexpr (NQ > 1 && bwd_rows(JB, 1) != 0) f1[C][U] = cl1[RO];
    });
  };
  bload(ic<0>{});
  {
    const double dlo = Dg[vlo ? ilo : 0], dhi = Dg[vhi ? ihi : 0];
    tlo = (fabs(dlo) > DBL_MIN) ? tlo / dlo : 0.0;
    thi = (fabs(dhi) > DBL_MIN) ? thi / dhi : 0.0;
    if (!vlo) tlo = 0.0;
    if (!vhi) thi = 0.0;
  }

  // ---------------- backward: blocks NB-1 .. 0, steps J = 15 .. 0 ----------------
  auto brun = [&]<int S>(ic<S>) {
    constexpr int JB = NB - 1 - (S >> 1), C = S & 1, RP = JB & 3, QP = JB >> 2;
    constexpr int RM0 = bwd_rows(JB, 0), RM1 = NQ > 1 ? bwd_rows(JB, 1) : 0;
    if constexpr (C == 0) tp = row_to_all<RP>(QP ? thi : tlo);
    static_for<8>([&]<int U>(ic<U>) {
      constexpr int J = 15 - 8 * C - U;
      if (JB < NB - 1 || 16 * JB + J < K) {  // (the last block: pivots beyond K do not exist)
        if constexpr (J >= 1) chain_le<J>(tp, ch[C][U], mk[J < 15 ? J + 1 : 0]);
        if constexpr (RM0 != 0) rider<J, RM0>(tlo, tp, f0[C][U]);
        if constexpr (RM1 != 0) rider<J, RM1>(thi, tp, f1[C][U]);
      }
    });
    if constexpr (C == 1) {
      if constexpr (QP) thi = (owner == (unsigned)RP) ? tp : thi;
      else tlo = (owner == (unsigned)RP) ? tp : tlo;
    }
  };
  static_for<2 * NB>([&]<int S>(ic<S>) {
    if constexpr (S + 1 < 2 * NB) bload(ic<S + 1>{});
    brun(ic<S>{});
  });
  return Pair{tlo, thi};
}

// NB = ceil(K / 16) is wave-uniform at run time: one instance per block count
__device__ __forceinline__ Pair row_sweeps_any(const int K, const double *T, const double *Dg, const Pair t, const int lane)
{
  switch ((__builtin_amdgcn_readfirstlane(K) + 15) >> 4) {
    case 0:
    case 1: return row_sweeps<1>(K, T, Dg, t, lane);
    case 2: return row_sweeps<2>(K, T, Dg, t, lane);
    case 3: return row_sweeps<3>(K, T, Dg, t, lane);
    case 4: return row_sweeps<4>(K, T, Dg, t, lane);
    case 5: return row_sweeps<5>(K, T, Dg, t, lane);
    case 6: return row_sweeps<6>(K, T, Dg, t, lane);
    case 7: return row_sweeps<7>(K, T, Dg, t, lane);
    default: return row_sweeps<8>(K, T, Dg, t, lane);
  }
}

}  // namespace rows
}  // namespace sfb
