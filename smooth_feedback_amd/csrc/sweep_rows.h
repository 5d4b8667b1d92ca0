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

// One operand of a sweep step from LDS as ONE ds_read_b64.  The load is volatile so that the compiler does not pair
// neighbours into ds_read2_b64: that instruction occupies the LDS for 8 cycles per wave (two passes of four 16-lane
// groups) against 2 for ds_read_b64 -- and with every resident wave sweeping, the LDS pipe of the CU is what the
// batch rate hangs on (measured: (16, 32) at 95 % of the ds_read2_b64 bound).
#ifndef SFB_ROWS_LD_PAIRED
__device__ __forceinline__ double ld1(const __attribute__((address_space(3))) double *p)
{
  return *(const volatile __attribute__((address_space(3))) double *)p;
}
#else
__device__ __forceinline__ double ld1(const __attribute__((address_space(3))) double *p) { return *p; }
#endif

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

// One half block (8 steps) as ONE asm statement: [lane mask; chain step; full mask; riders] x 8 -- nothing of the
// compiler's between the steps (it pads consecutive asm statements with s_nop and waits for every LDS operand
// separately: a lone wave pays ~5.5 cycles for every instruction of any kind), one s_waitcnt in front for all operands.
// Forward: step U is J = 8 C + U, lanes cc >= J (s_mov_b64 exec, mask_ge(J)); backward: J = 15 - 8 C - U, lanes
// cc <= J (s_not_b64 exec, mask_ge(J + 1); 0 for J = 15).  The last step of the second half has no chain operation
// (forward J = 15, backward J = 0: only the pivot lane would be left).  RM0 / RM1: row masks of the riders on tlo / thi, 0 = none.
#define SFB_RS_CHF(U) "s_mov_b64 exec, %[m" #U "]\n\tv_fmac_f64_dpp %[tp], %[tp], -%[c" #U "] row_newbcast:%[j" #U "] row_mask:0xf bank_mask:0xf\n\ts_mov_b64 exec, -1\n\t"
#define SFB_RS_CHB(U) "s_not_b64 exec, %[m" #U "]\n\tv_fmac_f64_dpp %[tp], %[tp], -%[c" #U "] row_newbcast:%[j" #U "] row_mask:0xf bank_mask:0xf\n\ts_mov_b64 exec, -1\n\t"
#define SFB_RS_NOC(U) ""
#define SFB_RS_R0(U) "v_fmac_f64_dpp %[t0], %[tp], -%[a" #U "] row_newbcast:%[j" #U "] row_mask:%[r0] bank_mask:0xf\n\t"
#define SFB_RS_R1(U) "v_fmac_f64_dpp %[t1], %[tp], -%[b" #U "] row_newbcast:%[j" #U "] row_mask:%[r1] bank_mask:0xf\n\t"
#define SFB_RS_NOR(U) ""
#define SFB_RS_STAGE(CH, CH7, RA, RB)                                                                                      \
  CH(0) RA(0) RB(0) CH(1) RA(1) RB(1) CH(2) RA(2) RB(2) CH(3) RA(3) RB(3) CH(4) RA(4) RB(4) CH(5) RA(5) RB(5) CH(6) RA(6) \
      RB(6) CH7(7) RA(7) RB(7)
#define SFB_RS_V8(p, a) [p##0] "v"(a[0]), [p##1] "v"(a[1]), [p##2] "v"(a[2]), [p##3] "v"(a[3]), [p##4] "v"(a[4]), [p##5] "v"(a[5]), [p##6] "v"(a[6]), [p##7] "v"(a[7])
#define SFB_RS_S8(m) [m0] "s"(m[0]), [m1] "s"(m[1]), [m2] "s"(m[2]), [m3] "s"(m[3]), [m4] "s"(m[4]), [m5] "s"(m[5]), [m6] "s"(m[6]), [m7] "s"(m[7])
#define SFB_RS_J8(J0, D) [j0] "n"(J0), [j1] "n"(J0 + D), [j2] "n"(J0 + 2 * D), [j3] "n"(J0 + 3 * D), [j4] "n"(J0 + 4 * D), [j5] "n"(J0 + 5 * D), [j6] "n"(J0 + 6 * D), [j7] "n"(J0 + 7 * D)
template<bool FWD, int C, int RM0, int RM1>
__device__ __forceinline__ void stage8(double &tp, double &tlo, double &thi, const double (&ch)[8], const double (&f0)[8],
                                       const double (&f1)[8], const unsigned long long (&m)[8])
{
  constexpr int J0 = FWD ? 8 * C : 15 - 8 * C, D = FWD ? 1 : -1;
  const double c[8] = {ch[0], ch[1], ch[2], ch[3], ch[4], ch[5], ch[6], C == 1 ? ch[6] : ch[7]};  // (no chain operand at the last step of C = 1)
  const double(&A)[8] = RM0 != 0 ? f0 : c;  // (absent riders: operands that cost no register)
  const double(&B)[8] = RM1 != 0 ? f1 : c;
#define SFB_RS_EMIT(CH, CH7, RA, RB)                                                                                            \
  asm volatile(SFB_RS_STAGE(CH, CH7, RA, RB)                                                                                    \
               : [tp] "+v"(tp), [t0] "+v"(tlo), [t1] "+v"(thi)                                                                  \
               : SFB_RS_V8(c, c), SFB_RS_V8(a, A), SFB_RS_V8(b, B), SFB_RS_S8(m), SFB_RS_J8(J0, D), [r0] "n"(RM0), [r1] "n"(RM1) \
               : "scc")
#define SFB_RS_PICK(CH, CH7)                                                                \
  if constexpr (RM0 != 0 && RM1 != 0) SFB_RS_EMIT(CH, CH7, SFB_RS_R0, SFB_RS_R1);           \
  else if constexpr (RM0 != 0) SFB_RS_EMIT(CH, CH7, SFB_RS_R0, SFB_RS_NOR);                 \
  else if constexpr (RM1 != 0) SFB_RS_EMIT(CH, CH7, SFB_RS_NOR, SFB_RS_R1);                 \
  else SFB_RS_EMIT(CH, CH7, SFB_RS_NOR, SFB_RS_NOR)
  if constexpr (FWD && C == 0) { SFB_RS_PICK(SFB_RS_CHF, SFB_RS_CHF); }
  else if constexpr (FWD) { SFB_RS_PICK(SFB_RS_CHF, SFB_RS_NOC); }
  else if constexpr (C == 0) { SFB_RS_PICK(SFB_RS_CHB, SFB_RS_CHB); }
  else { SFB_RS_PICK(SFB_RS_CHB, SFB_RS_NOC); }
#undef SFB_RS_PICK
#undef SFB_RS_EMIT
}

// (L D L')^-1 applied to the PERMUTED vector t held in registers (rows lane, lane + 64): L^-1, D^-1, L^-T -- the middle
// of oracle_ldlt_solve.  T: the packed lower triangle in LDS, entry (i, j), j < i, at T[i (i + 1) / 2 + j], its diagonal
// slots at -0.0 (chain_ge); Dg: the diagonal D.  NB = ceil(K / 16) blocks, compile time; everything is unrolled over compile-time block and step numbers
// (pivot lane, LDS offsets, lane and row masks are immediates).  The operands of a half block (8 steps) are fetched
// from LDS while the previous half block runs.  An instance serves every K <= 16 NB (half blocks beyond K are skipped
// by one wave-uniform test each): the polish system of a QP goes through the instance of its ADMM loop.
// The lane masks of the chain steps as opaque SGPR pairs (a 64-bit mask is no literal of s_mov_b64: the compiler would
// otherwise build it from two 32-bit moves at every use).  A caller with a loop around the sweeps makes them ONCE in
// front of it -- made inside, they are re-built and the registers they displace spilled and reloaded every iteration.
struct Masks {
  unsigned long long mk[16];  // mk[J] = lanes cc >= J of every 16-lane row, J = 1 .. 15; mk[0] = all
  unsigned long long zero;
};
__device__ __forceinline__ Masks make_masks()
{
  Masks M;
  M.mk[0] = ~0ull;
  M.zero  = 0ull;
  asm volatile("" : "+s"(M.mk[0]), "+s"(M.zero));
  static_for<15>([&]<int J>(ic<J>) {
    M.mk[J + 1] = mask_ge(J + 1);
    asm volatile("" : "+s"(M.mk[J + 1]));
  });
  return M;
}

template<int NB, bool ANYK = true>  // ANYK: any K <= 16 NB; otherwise 16 (NB - 1) < K <= 16 NB (only the last block can be partial: no tests on the others)
__device__ __forceinline__ Pair row_sweeps_inl(const int K_, const double *T_, const double *Dg_, Pair t, const int lane, const Masks &M)
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
  const unsigned long long(&mk)[16] = M.mk;  // the lane masks of the chain steps, resident in SGPR pairs
  const unsigned long long mzero    = M.zero;
  const unsigned owner = (unsigned)r;
  double tp = 0.0;
  double ch[2][8], f0[2][8], f1[2][8];

  // ---------------- forward: blocks 0 .. NB-1, steps J = 0 .. 15 (stage S = 2 JB + half) ----------------
  auto fload = [&]<int S>(ic<S>) {
    constexpr int JB = S >> 1, C = S & 1;
    static_for<8>([&]<int U>(ic<U>) {
      constexpr int J = 8 * C + U;
      if constexpr (J < 15) ch[C][U] = ld1(pch[JB] + J);
      if constexpr (fwd_rows<NB>(JB, 0) != 0) f0[C][U] = ld1(plo + 16 * JB + J);
      if constexpr (NQ > 1 && fwd_rows<NB>(JB, 1) != 0) f1[C][U] = ld1(phi + 16 * JB + J);
    });
  };
  auto frun = [&]<int S>(ic<S>) {
    constexpr int JB = S >> 1, C = S & 1, RP = JB & 3, QP = JB >> 2;
    constexpr int RM0 = fwd_rows<NB>(JB, 0), RM1 = NQ > 1 ? fwd_rows<NB>(JB, 1) : 0;
    if constexpr (ANYK)
      if (16 * JB >= K) return;  // (blocks beyond K: rows that do not exist)
    if constexpr (C == 0) tp = row_to_all<RP>(QP ? thi : tlo);
    if ((!ANYK && JB < NB - 1) || 16 * JB + 8 * C < K) {  // (a half block beyond K: pivots there only reach rows that do not exist)
      const unsigned long long m8[8] = {mk[8 * C], mk[8 * C + 1], mk[8 * C + 2], mk[8 * C + 3], mk[8 * C + 4], mk[8 * C + 5], mk[8 * C + 6], mk[8 * C + 7]};
      stage8<true, C, RM0, RM1>(tp, tlo, thi, ch[C], f0[C], f1[C], m8);
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
      if constexpr (J >= 1) ch[C][U] = ld1(pcc + RO + 16 * JB);
      if constexpr (bwd_rows(JB, 0) != 0) f0[C][U] = ld1(cl0 + RO);
      if constexpr (NQ > 1 && bwd_rows(JB, 1) != 0) f1[C][U] = ld1(cl1 + RO);
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
    if constexpr (ANYK)
      if (16 * JB >= K) return;
    if constexpr (C == 0) tp = row_to_all<RP>(QP ? thi : tlo);
    if ((!ANYK && JB < NB - 1) || 16 * JB + 15 - 8 * C < K) {  // every pivot of this half block exists
      // step U is J = 15 - 8 C - U: s_not_b64 of mask_ge(J + 1) (of nothing at J = 15: every lane)
      const unsigned long long m8[8] = {C == 0 ? mzero : mk[8], mk[15 - 8 * C], mk[14 - 8 * C], mk[13 - 8 * C], mk[12 - 8 * C], mk[11 - 8 * C], mk[10 - 8 * C], mk[9 - 8 * C]};
      stage8<false, C, RM0, RM1>(tp, tlo, thi, ch[C], f0[C], f1[C], m8);
    } else if ((!ANYK && JB < NB - 1) || 16 * JB + 8 - 8 * C < K) {  // the block that holds row K - 1 without being full: pivots beyond K do not exist, step by step
      static_for<8>([&]<int U>(ic<U>) {
        constexpr int J = 15 - 8 * C - U;
        if (16 * JB + J < K) {
          if constexpr (J >= 1) chain_le<J>(tp, ch[C][U], mk[J < 15 ? J + 1 : 0]);
          if constexpr (RM0 != 0) rider<J, RM0>(tlo, tp, f0[C][U]);
          if constexpr (RM1 != 0) rider<J, RM1>(thi, tp, f1[C][U]);
        }
      });
    }
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

// ---------------------------------------------------------------------------------------------------------------
// REGISTERS-ONLY engine, K <= 64 (NB <= 4 blocks, one row per lane): the whole factor of the QP lives in VGPRs for
// the duration of the ADMM loop and a sweep touches no LDS at all.
//
// The block engine above keeps the pivot block's chain replicated in all four 16-lane rows so that the riders can read
// x_J while the chain is still running -- which makes every lane need the chain operands of EVERY block (16 NB doubles
// if they were kept in registers).  Here the two are separated per pivot block JB:
//   1. chain: the in-block substitution runs in row JB only, in place on the lanes' own values
//      (v_fmac_f64_dpp t, t, -ch[J] row_newbcast:J row_mask:1<<JB, lanes cc >= J / cc <= J through EXEC) -- a lane needs the
//      operands of ITS OWN block only: ch[J] = Lsym(cc, J), the symmetric 16 x 16 diagonal block (row form for the
//      forward sweep where cc > J, column form for the backward sweep where cc < J, the factorisation's -0.0 at cc == J:
//      the triangles are complementary, one array serves both sweeps like in qp_dense4.hip);
//   2. the finished segment goes to all rows (row_to_all: v_permlane16_swap / v_permlane32_swap, no LDS);
//   3. riders: every row beyond (forward) / before (backward) the pivot block takes the block's 16 updates,
//      v_fmac_f64_dpp t, xb, -ro[.][J] row_newbcast:J, J ascending / descending -- 16 (NB - 1) operands per lane, the same
//      array for both sweeps (rows beyond block o hold L(row, 16 o + J), rows up to block o hold L(16 (o + 1) + J, row)).
// 16 NB doubles per lane in all (96 VGPRs at K <= 48, 128 at K <= 64) against 16 (2 NB - 1) for a replicated chain.
// Per row the operations and their order are the oracle's, exactly as in the block engine: row i subtracts L(i, j) x_j
// for j ascending (riders of the earlier blocks in block order, then its own block's chain), L(j, i) x_j for j
// descending on the way back; lanes a step does not concern are switched off (EXEC / row_mask): no 0 * pivot products.
// A lone wave pays the chain's dependent latency un-hidden (18-20 cycles per step) but issues 4 instructions per step
// instead of 6-8 and waits for no operand; with 2-3 waves per SIMD the VALU is the only shared resource left.
template<int NB>
struct RegFactor {
  static_assert(NB >= 2 && NB <= 4, "32 < K <= 64");
  double ch[16];
  double ro[NB - 1][16];
  double d;
};

// F <- the packed factor T (LDS), for the lane's row = lane.  16 (NB - 1) < K <= 16 NB.  Entries of rows / columns that do not
// exist are exact zeros.
template<int NB>
__device__ __forceinline__ void reg_factor_load(RegFactor<NB> &F, const int K_, const double *T_, const double *Dg_, const int lane)
{
  const int K           = __builtin_amdgcn_readfirstlane(K_);
  const lds_d *const T  = (const lds_d *)T_;
  const lds_d *const Dg = (const lds_d *)Dg_;
  // (the addresses below depend on the lane and K only: inside a persistent loop over QPs the compiler would hoist all
  // 16 NB of them out of the loop and keep them in scratch memory -- the opaque copy of the lane number ties them to the call)
  int ln = lane;
  asm volatile("" : "+v"(ln));
  const int cc = ln & 15, r = ln >> 4, i = ln;
  const bool vi = i < K;
  // Addresses as [per-lane base] + [J x per-lane stride] + [compile-time offset]; an entry that does not exist reads T[0]
  // (valid LDS) and is replaced by an exact zero.  Diagonal block of row block r, entry (a, b) = (max, min)(cc, J) at
  // tri(16 r + a) + 16 r + b:  J < cc: tri(i) + 16 r + J;  J >= cc: tri(16 r + J) + i = [tri(16 r) + i] + J 16 r + tri(J).
  const int rowA = vi ? ((i * (i + 1)) >> 1) : 0;            // tri(i)
  const int baseA = rowA + 16 * r;                            // + J
  const int r16   = 16 * r;
  const int baseB = ((r16 * (r16 + 1)) >> 1) + i;             // + J r16 + tri(J)
  static_for<16>([&]<int J>(ic<J>) {
    const bool lower = J < cc;  // the lane's own row holds the entry
    const bool ok    = lower ? vi : (r16 + J < K);
    const int idx    = lower ? baseA + J : baseB + J * r16 + (J * (J + 1)) / 2;
    const double v   = T[ok ? idx : 0];
    F.ch[J] = ok ? v : 0.0;
  });
  static_for<NB - 1>([&]<int O>(ic<O>) {
    const bool fwd = r > O;  // rows beyond block O: forward operands L(i, 16 O + J); rows up to it: backward operands L(16 (O + 1) + J, i)
    static_for<16>([&]<int J>(ic<J>) {
      constexpr int p = 16 * (O + 1) + J;
      const bool ok   = vi && (fwd || O + 1 < NB - 1 || p < K);  // (only the last block can be partial)
      const int idx   = fwd ? rowA + (16 * O + J) : i + (p * (p + 1)) / 2;
      const double v  = T[ok ? idx : 0];
      F.ro[O][J] = ok ? v : 0.0;
    });
  });
  F.d = Dg[vi ? i : 0];
}

// chain steps of one half block (8 steps; the last step of the second half has no operation) as ONE asm statement.  The
// lane mask of a step is the same 32-bit pattern in both halves of EXEC (four 16-lane rows): two s_mov_b32 with a literal --
// no SGPRs for the masks -- which are also the two wait states the DPP read of t needs after the VALU write before it;
// nothing rides between the chain steps here, so EXEC stays narrowed from step to step and is restored once at the end.
__host__ __device__ constexpr unsigned mask32_ge(const int J) { return (unsigned)(mask_ge(J) & 0xFFFFFFFFull); }
#define SFB_RG_CH(U) "s_mov_b32 exec_lo, %[m" #U "]\n\ts_mov_b32 exec_hi, %[m" #U "]\n\tv_fmac_f64_dpp %[t], %[t], -%[c" #U "] row_newbcast:%[j" #U "] row_mask:%[rm] bank_mask:0xf\n\t"
#define SFB_RG_RID(U) "v_fmac_f64_dpp %[t], %[x], -%[c" #U "] row_newbcast:%[j" #U "] row_mask:%[rm] bank_mask:0xf\n\t"
#define SFB_RG_M8(M) [m0] "n"(M(0)), [m1] "n"(M(1)), [m2] "n"(M(2)), [m3] "n"(M(3)), [m4] "n"(M(4)), [m5] "n"(M(5)), [m6] "n"(M(6)), [m7] "n"(M(7))
template<bool FWD, int C, int RP>
__device__ __forceinline__ void reg_chain8(double &t, const double (&ch)[16])
{
  constexpr int J0 = FWD ? 8 * C : 15 - 8 * C, D = FWD ? 1 : -1;
  const double c[8] = {ch[J0], ch[J0 + D], ch[J0 + 2 * D], ch[J0 + 3 * D], ch[J0 + 4 * D], ch[J0 + 5 * D], ch[J0 + 6 * D], ch[C == 1 ? J0 + 6 * D : J0 + 7 * D]};
  // forward step J: lanes cc >= J; backward step J: lanes cc <= J
#define SFB_RG_MASK(U) (FWD ? mask32_ge(J0 + (U) * D) : (~mask32_ge(J0 + (U) * D + 1) & 0xFFFFFFFFu))
  if constexpr (C == 0)
    asm volatile(SFB_RG_CH(0) SFB_RG_CH(1) SFB_RG_CH(2) SFB_RG_CH(3) SFB_RG_CH(4) SFB_RG_CH(5) SFB_RG_CH(6) SFB_RG_CH(7) "s_mov_b64 exec, -1"
                 : [t] "+v"(t)
                 : SFB_RS_V8(c, c), SFB_RG_M8(SFB_RG_MASK), SFB_RS_J8(J0, D), [rm] "n"(1 << RP));
  else
    asm volatile(SFB_RG_CH(0) SFB_RG_CH(1) SFB_RG_CH(2) SFB_RG_CH(3) SFB_RG_CH(4) SFB_RG_CH(5) SFB_RG_CH(6) "s_mov_b64 exec, -1"
                 : [t] "+v"(t)
                 : SFB_RS_V8(c, c), SFB_RG_M8(SFB_RG_MASK), SFB_RS_J8(J0, D), [rm] "n"(1 << RP));
#undef SFB_RG_MASK
}
// the riders of one half block: t(cc) = fma(-ro[J], xb(J), t(cc)) on the rows of RM, J = J0, J0 + D, ...
template<bool FWD, int C, int RM>
__device__ __forceinline__ void reg_riders8(double &t, const double &xb, const double (&ro)[16])
{
  constexpr int J0 = FWD ? 8 * C : 15 - 8 * C, D = FWD ? 1 : -1;
  const double c[8] = {ro[J0], ro[J0 + D], ro[J0 + 2 * D], ro[J0 + 3 * D], ro[J0 + 4 * D], ro[J0 + 5 * D], ro[J0 + 6 * D], ro[J0 + 7 * D]};
  asm volatile("s_nop 1\n\t" SFB_RG_RID(0) SFB_RG_RID(1) SFB_RG_RID(2) SFB_RG_RID(3) SFB_RG_RID(4) SFB_RG_RID(5) SFB_RG_RID(6) SFB_RG_RID(7)
               : [t] "+v"(t)
               : [x] "v"(xb), SFB_RS_V8(c, c), SFB_RS_J8(J0, D), [rm] "n"(RM));
}
// single steps (the half block that holds row K - 1 without being full, backward sweep)
template<int J, int RP>
__device__ __forceinline__ void reg_chain_le(double &t, const double l)
{
  asm volatile("s_mov_b32 exec_lo, %2\n\ts_mov_b32 exec_hi, %2\n\t"
               "v_fmac_f64_dpp %0, %0, -%1 row_newbcast:%3 row_mask:%4 bank_mask:0xf\n\t"
               "s_mov_b64 exec, -1"
               : "+v"(t)
               : "v"(l), "n"(~mask32_ge(J + 1) & 0xFFFFFFFFu), "n"(J), "n"(1 << RP));
}

// (L D L')^-1 applied to the permuted vector t (row = lane) from the factor in registers.  16 (NB - 1) < K <= 16 NB.
template<int NB>
__device__ __forceinline__ double row_sweeps_reg(const int K_, const RegFactor<NB> &F, const double t, const int lane)
{
  const int K   = __builtin_amdgcn_readfirstlane(K_);
  const bool vi = lane < K;
  double tl     = vi ? t : 0.0;
  // ---------------- forward: blocks 0 .. NB-1 ----------------
  static_for<NB>([&]<int JB>(ic<JB>) {
    constexpr bool LAST = JB == NB - 1;
    static_for<2>([&]<int C>(ic<C>) {
      if (!LAST || 16 * JB + 8 * C < K) reg_chain8<true, C, JB>(tl, F.ch);  // (a half block beyond K: pivots there only reach rows that do not exist)
    });
    if constexpr (!LAST) {
      constexpr int RM = fwd_rows<NB>(JB, 0);
      const double xb  = row_to_all<JB>(tl);
      reg_riders8<true, 0, RM>(tl, xb, F.ro[JB]);
      reg_riders8<true, 1, RM>(tl, xb, F.ro[JB]);
    }
  });
  // ---------------- D^-1 (|d| <= DBL_MIN -> 0, true division) ----------------
  tl = (fabs(F.d) > DBL_MIN) ? tl / F.d : 0.0;
  if (!vi) tl = 0.0;
  // ---------------- backward: blocks NB-1 .. 0 ----------------
  static_for<NB>([&]<int S>(ic<S>) {
    constexpr int JB    = NB - 1 - S;
    constexpr bool LAST = JB == NB - 1;
    static_for<2>([&]<int C>(ic<C>) {
      if (!LAST || 16 * JB + 15 - 8 * C < K) {  // every pivot of this half block exists
        reg_chain8<false, C, JB>(tl, F.ch);
      } else if (16 * JB + 8 - 8 * C < K) {  // the half block that holds row K - 1 without being full: step by step
        static_for<8>([&]<int U>(ic<U>) {
          constexpr int J = 15 - 8 * C - U;
          if constexpr (J >= 1)
            if (16 * JB + J < K) reg_chain_le<J, JB>(tl, F.ch[J]);
        });
      }
    });
    if constexpr (JB > 0) {
      constexpr int RM = bwd_rows(JB, 0);
      const double xb  = row_to_all<JB>(tl);
      static_for<2>([&]<int C>(ic<C>) {
        if (!LAST || 16 * JB + 15 - 8 * C < K) {
          reg_riders8<false, C, RM>(tl, xb, F.ro[JB - 1]);
        } else if (16 * JB + 8 - 8 * C < K) {
          static_for<8>([&]<int U>(ic<U>) {
            constexpr int J = 15 - 8 * C - U;
            if (16 * JB + J < K) {
              double tt = tl;
              asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:%4 bank_mask:0xf" : "+v"(tt) : "v"(xb), "v"(F.ro[JB - 1][J]), "n"(J), "n"(RM));
              tl = tt;
            }
          });
        }
      });
    }
  });
  return tl;
}

// outlined instance (call sites off the hot loop); any K <= 16 NB
template<int NB, bool ANYK = true>
__device__ __attribute__((noinline)) Pair row_sweeps(const int K_, const double *T_, const double *Dg_, Pair t, const int lane)
{
  return row_sweeps_inl<NB, ANYK>(K_, T_, Dg_, t, lane, make_masks());
}

// NB = ceil(K / 16) is wave-uniform at run time: one instance per block count
__device__ __forceinline__ Pair row_sweeps_any(const int K, const double *T, const double *Dg, const Pair t, const int lane)
{
  switch ((__builtin_amdgcn_readfirstlane(K) + 15) >> 4) {
    case 0:
    case 1: return row_sweeps<1, false>(K, T, Dg, t, lane);
    case 2: return row_sweeps<2, false>(K, T, Dg, t, lane);
    case 3: return row_sweeps<3, false>(K, T, Dg, t, lane);
    case 4: return row_sweeps<4, false>(K, T, Dg, t, lane);
    case 5: return row_sweeps<5, false>(K, T, Dg, t, lane);
    case 6: return row_sweeps<6, false>(K, T, Dg, t, lane);
    case 7: return row_sweeps<7, false>(K, T, Dg, t, lane);
    default: return row_sweeps<8, false>(K, T, Dg, t, lane);
  }
}

}  // namespace rows
}  // namespace sfb
