// Batched dense ADMM QP solver for gfx950, k = n+m <= 32: FOUR QPs PER WAVEFRONT, one per 16-lane row.
//
// Same algorithm and the same bits as qp_dense.hip (reference qp_solver.hpp:343-568); what changes is
// the mapping of the ADMM loop onto the wave:
//   - a DPP row_newbcast FP64 fmac costs the same VALU time whether 16 or 64 lanes do useful work,
//     and the one-QP-per-wave kernel keeps only one 16-lane row busy per sweep step.  Here every row
//     carries its own QP ("slot"): lane cc of a row holds system rows cc and 16+cc, so a whole
//     triangular sweep stays inside the row (no row swaps) and one instruction stream advances four
//     QPs in lockstep.  The L10 block update is interleaved with the L00 chain (step J of one is the
//     wait state of the other);
//   - iteration counts differ per QP, so slots are refilled from a device-side queue (atomic ticket):
//     a persistent grid of waves, each slot takes the next problem when its own finishes.  New
//     problems start only on iterations that are multiples of stop_check_iter, which keeps the
//     stopping checks of the four slots on the same iteration; the check itself runs row-parallel;
//   - problem setup (scaling, KKT, pivoted LDL'), polish and reporting are the wave-wide routines of
//     qp_dense_common.h working on the slot's LDS area; the other three slots just wait (setup is a
//     few iterations' worth of time against hundreds to thousands of iterations per problem).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <mutex>

#include "../../include/sfb.h"
#include "qp_dense_common.h"

namespace sfb {

namespace {

constexpr int kSlots = 4;

// max over the 16 lanes of each row, every lane of the row gets it
__device__ __forceinline__ double row_max16(double v)
{
  v = fmax(v, dpp_mov<0x128>(v));
  v = fmax(v, dpp_mov<0x124>(v));
  v = fmax(v, dpp_mov<0x122>(v));
  v = fmax(v, dpp_mov<0x121>(v));
  return v;
}

struct Lds4 {
  int shared_doubles;  // W, temp, rho, perm, LU (setup / polish scratch, one problem at a time)
  int slot_doubles;    // stride of a slot area
};

__host__ __device__ inline Lds4 lds4_layout(int n, int m)
{
  const int k = n + m;
  Lds4 L;
  int sh = (k * (k + 1)) / 2 + k + m + (k + m + 1) / 2;  // W, temp, rho, (perm, LU as ints)
  sh     = (sh + 1) & ~1;
  L.shared_doubles = sh;
  int sl = n * n + m * n + 4 * n + 7 * m + 2;  // P A | q sx xv dxus | l u sy yv zus dyus (+1 m spare) | cval
  // rows of a wave read the same offsets of different slots: stride = 16 mod 32 doubles puts
  // neighbouring slots on complementary LDS banks
  sl = ((sl + 15) & ~31) + 16;
  L.slot_doubles = sl;
  return L;
}

// view of slot `slot` (shared scratch + the slot's own arrays)
__device__ __forceinline__ Lds slot_view(double *base, const Lds4 &L, const int n, const int m, const int slot,
                                         double *&cval)
{
  const int k = n + m;
  Lds s;
  double *p = base;
  s.W    = p; p += (k * (k + 1)) >> 1;
  s.temp = p; p += k;
  s.rho  = p; p += m;
  s.perm = reinterpret_cast<int *>(p);
  s.LU   = s.perm + k;
  p      = base + L.shared_doubles + slot * L.slot_doubles;
  s.P    = p; p += n * n;
  s.A    = p; p += m * n;
  s.q    = p; p += n;
  s.l    = p; p += m;
  s.u    = p; p += m;
  s.sx   = p; p += n;
  s.sy   = p; p += m;
  s.xv   = p; p += n;
  s.yv   = p; p += m;
  s.zus  = p; p += m;
  s.dxus = p; p += n;   // dxus | dyus contiguous: qp_polish uses them as one k-vector
  s.dyus = p; p += m;
  cval   = p;
  return s;
}

// QPSolver::check_stopping (qp_solver.hpp:574-644), ROW-PARALLEL: every 16-lane row checks its own
// slot; lane cc handles entries cc and 16+cc.  Same values as qp_check_stopping (the norms are
// maxima, every sum keeps its sequential order).  Returns a status or -1, uniform per row.
template<int NB>
__device__ inline int check_rows(const Lds &s, const DenseKernelParams &kp, const int n, const int m, const int cc)
{
  const double inf = INFINITY;
  int res          = -1;
  bool ln[NB], lm[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    ln[b] = cc + 16 * b < n;
    lm[b] = cc + 16 * b < m;
  }

  // OPTIMALITY :584-594
  {
    double Ax[NB], zi[NB];
    double a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int i = cc + 16 * b;
      Ax[b]       = lm[b] ? row_A(s, n, m, i, s.xv) : 0.0;
      zi[b]       = lm[b] ? s.zus[i] : 0.0;
      a1          = fmax(a1, fabs(Ax[b]));
      a2          = fmax(a2, lm[b] ? fabs(Ax[b] - zi[b]) : 0.0);
      a3          = fmax(a3, fabs(zi[b]));
    }
    const double Ax_norm = row_max16(a1), r_norm = row_max16(a2), z_norm = row_max16(a3);
    if (r_norm <= kp.eps_abs + kp.eps_rel * fmax(Ax_norm, z_norm)) {
      double p1 = 0.0, p2 = 0.0, p3 = 0.0, p4 = 0.0;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int i      = cc + 16 * b;
        const double Px  = ln[b] ? row_P(s, n, i, s.xv) : 0.0;
        const double Aty = ln[b] ? row_At(s, n, m, i, s.yv) : 0.0;
        const double qi  = ln[b] ? s.q[i] : 0.0;
        const double rs  = ln[b] ? Px + (qi + Aty) : 0.0;
        p1 = fmax(p1, fabs(Px));
        p2 = fmax(p2, fabs(qi));
        p3 = fmax(p3, fabs(Aty));
        p4 = fmax(p4, fabs(rs));
      }
      const double dual_scale = fmax(fmax(row_max16(p1), row_max16(p2)), row_max16(p3));
      if (row_max16(p4) <= kp.eps_abs + kp.eps_rel * dual_scale) res = SFB_QP_OPTIMAL;
    }
  }

  // PRIMAL INFEASIBILITY :598-621
  if (res < 0) {
    double a1 = 0.0, a2 = 0.0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int i = cc + 16 * b;
      a1          = fmax(a1, ln[b] ? fabs(row_At(s, n, m, i, s.dyus)) : 0.0);
      a2          = fmax(a2, lm[b] ? fabs(s.dyus[i]) : 0.0);
    }
    const double Aty_norm = row_max16(a1), Edy_norm = row_max16(a2);
    const double thr      = kp.eps_pinf * Edy_norm;
    double acc            = 0.0;  // sequential with early exit, every lane of the row runs it
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
    const double mxv = (Aty_norm < acc) ? acc : Aty_norm;  // std::max(a,b) = (a<b)?b:a
    if (mxv < thr) res = SFB_QP_PRIMAL_INFEASIBLE;
  }

  // DUAL INFEASIBILITY :625-641
  if (res < 0) {
    double a1 = 0.0, a2 = 0.0;
    double Adx[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int i = cc + 16 * b;
      Adx[b]      = lm[b] ? row_A(s, n, m, i, s.dxus) : 0.0;
      a1          = fmax(a1, ln[b] ? fabs(s.dxus[i]) : 0.0);
      a2          = fmax(a2, ln[b] ? fabs(row_P(s, n, i, s.dxus)) : 0.0);
    }
    const double dx_norm = row_max16(a1), Pdx_n = row_max16(a2);
    double qdx           = 0.0;
    for (int j = 0; j < n; ++j) qdx = fma(s.q[j], s.dxus[j], qdx);
    const double thr = kp.eps_dinf * dx_norm;
    const bool ok    = (Pdx_n <= thr) && (qdx <= thr);
    double viol      = 0.0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (lm[b]) {
        const int i     = cc + 16 * b;
        const double ui = s.u[i], li = s.l[i];
        bool rowok;
        if (ui == inf) {
          rowok = Adx[b] >= -thr;
        } else if (li == -inf) {
          rowok = Adx[b] <= thr;
        } else {
          rowok = fabs(Adx[b]) < thr;
        }
        if (!rowok) viol = 1.0;
      }
    }
    if (ok && row_max16(viol) == 0.0) res = SFB_QP_DUAL_INFEASIBLE;
  }
  return res;
}

// Negated 16x16 blocks of the unit-lower factor L = [L00 0; L10 L11] as lane cc of a row needs them:
//   D0[j] = -L00(cc, j)  for j < cc  (forward, row form)   |  -L00(j, cc) for j > cc (backward, column form)
//   D1[j] = the same for L11;   F10[j] = -L10(cc, j);   B10[j] = -L10(j, cc)
// The two triangles of a diagonal block share one register array.  A sweep step is one
// v_fmac_f64_dpp over the whole row, so the lanes that hold the OTHER triangle's entry for this j are
// switched off through EXEC for that instruction (forward step J: lanes cc >= J, backward: cc <= J;
// the pivot lane stays on because DPP cannot read a disabled lane, its own entry D[J] is 0).  The two
// s_mov_b32 that set EXEC are also the two wait states a DPP read needs after the VALU write of its
// source, so the masking costs no issue slots over the plain `s_nop 1` form.
template<int NB>
struct Factor {
  double D0[16];
  double D1[NB > 1 ? 16 : 1], F10[NB > 1 ? 16 : 1], B10[NB > 1 ? 16 : 1];
};

constexpr unsigned mask_ge(int J)  // lanes cc >= J of every row
{
  const unsigned m16 = (0xFFFFu << J) & 0xFFFFu;
  return m16 | (m16 << 16);
}
constexpr unsigned mask_le(int J)  // lanes cc <= J of every row
{
  const unsigned m16 = (1u << (J + 1)) - 1u;
  return m16 | (m16 << 16);
}

#define SFB_SELF(J, OPD, OPM)                                                                       \
  "s_mov_b32 exec_lo, %" #OPM "\n\ts_mov_b32 exec_hi, %" #OPM "\n\tv_fmac_f64_dpp %0, %0, %" #OPD \
  " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"

// t(cc) -= sum_{j<cc} L(cc,j) t(j), j ascending: the in-row forward chain of a diagonal block
__device__ __forceinline__ void chain_fwd(double &t, const double (&D)[16])
{
  asm volatile(SFB_SELF(0, 1, 16) SFB_SELF(1, 2, 17) SFB_SELF(2, 3, 18) SFB_SELF(3, 4, 19) SFB_SELF(4, 5, 20)
               SFB_SELF(5, 6, 21) SFB_SELF(6, 7, 22) SFB_SELF(7, 8, 23) SFB_SELF(8, 9, 24) SFB_SELF(9, 10, 25)
               SFB_SELF(10, 11, 26) SFB_SELF(11, 12, 27) SFB_SELF(12, 13, 28) SFB_SELF(13, 14, 29)
               SFB_SELF(14, 15, 30) "s_mov_b64 exec, -1"
               : "+v"(t)
               : "v"(D[0]), "v"(D[1]), "v"(D[2]), "v"(D[3]), "v"(D[4]), "v"(D[5]), "v"(D[6]), "v"(D[7]), "v"(D[8]),
                 "v"(D[9]), "v"(D[10]), "v"(D[11]), "v"(D[12]), "v"(D[13]), "v"(D[14]), "n"(mask_ge(0)),
                 "n"(mask_ge(1)), "n"(mask_ge(2)), "n"(mask_ge(3)), "n"(mask_ge(4)), "n"(mask_ge(5)),
                 "n"(mask_ge(6)), "n"(mask_ge(7)), "n"(mask_ge(8)), "n"(mask_ge(9)), "n"(mask_ge(10)),
                 "n"(mask_ge(11)), "n"(mask_ge(12)), "n"(mask_ge(13)), "n"(mask_ge(14)));
}
// t(cc) -= sum_{j>cc} L(j,cc) t(j), j descending: the in-row backward chain
__device__ __forceinline__ void chain_bwd(double &t, const double (&D)[16])
{
  asm volatile(SFB_SELF(15, 1, 16) SFB_SELF(14, 2, 17) SFB_SELF(13, 3, 18) SFB_SELF(12, 4, 19) SFB_SELF(11, 5, 20)
               SFB_SELF(10, 6, 21) SFB_SELF(9, 7, 22) SFB_SELF(8, 8, 23) SFB_SELF(7, 9, 24) SFB_SELF(6, 10, 25)
               SFB_SELF(5, 11, 26) SFB_SELF(4, 12, 27) SFB_SELF(3, 13, 28) SFB_SELF(2, 14, 29)
               SFB_SELF(1, 15, 30) "s_mov_b64 exec, -1"
               : "+v"(t)
               : "v"(D[15]), "v"(D[14]), "v"(D[13]), "v"(D[12]), "v"(D[11]), "v"(D[10]), "v"(D[9]), "v"(D[8]),
                 "v"(D[7]), "v"(D[6]), "v"(D[5]), "v"(D[4]), "v"(D[3]), "v"(D[2]), "v"(D[1]), "n"(mask_le(15)),
                 "n"(mask_le(14)), "n"(mask_le(13)), "n"(mask_le(12)), "n"(mask_le(11)), "n"(mask_le(10)),
                 "n"(mask_le(9)), "n"(mask_le(8)), "n"(mask_le(7)), "n"(mask_le(6)), "n"(mask_le(5)),
                 "n"(mask_le(4)), "n"(mask_le(3)), "n"(mask_le(2)), "n"(mask_le(1)));
}
#undef SFB_SELF

// Step J of the block-0 forward chain with the t1 -= L10(:,J) t0(J) update riding on it: t0(J) is
// final once step J-1 is done, and each of the two fmacs sits in the other's wait states.
template<int J, bool FIRST>
__device__ __forceinline__ void pair_fwd(double &t0, double &t1, const double f10, const double d0)
{
  if constexpr (J < 15) {
    asm volatile("s_nop %6\n\t"
                 "v_fmac_f64_dpp %1, %0, %2 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
                 "s_mov_b32 exec_lo, %4\n\ts_mov_b32 exec_hi, %4\n\t"
                 "v_fmac_f64_dpp %0, %0, %3 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
                 "s_mov_b64 exec, -1"
                 : "+v"(t0), "+v"(t1)
                 : "v"(f10), "v"(d0), "n"(mask_ge(J)), "n"(J), "n"(FIRST ? 1 : 0));
  } else {
    asm volatile("s_nop 0\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:15 row_mask:0xf bank_mask:0xf"
                 : "+v"(t1)
                 : "v"(t0), "v"(f10));
  }
}
// backward twin: t0 -= L10(J,:)' t1(J), then step J of the block-1 backward chain
template<int J, bool FIRST>
__device__ __forceinline__ void pair_bwd(double &t0, double &t1, const double b10, const double d1)
{
  if constexpr (J >= 1) {
    asm volatile("s_nop %6\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
                 "s_mov_b32 exec_lo, %4\n\ts_mov_b32 exec_hi, %4\n\t"
                 "v_fmac_f64_dpp %1, %1, %3 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
                 "s_mov_b64 exec, -1"
                 : "+v"(t0), "+v"(t1)
                 : "v"(b10), "v"(d1), "n"(mask_le(J)), "n"(J), "n"(FIRST ? 1 : 0));
  } else {
    asm volatile("s_nop 0\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf"
                 : "+v"(t0)
                 : "v"(t1), "v"(b10));
  }
}

template<int J>
struct Pairs {
  static __device__ __forceinline__ void fwd(double &t0, double &t1, const double (&F10)[16], const double (&D0)[16])
  {
    if constexpr (J < 16) {
      pair_fwd<J, J == 0>(t0, t1, F10[J], D0[J]);
      Pairs<J + 1>::fwd(t0, t1, F10, D0);
    }
  }
  static __device__ __forceinline__ void bwd(double &t0, double &t1, const double (&B10)[16], const double (&D1)[16])
  {
    if constexpr (J >= 0) {
      pair_bwd<J, J == 15>(t0, t1, B10[J], D1[J]);
      Pairs<J - 1>::bwd(t0, t1, B10, D1);
    }
  }
};

// K^-1 t without the D^-1 in the middle: forward / backward unit-triangular sweeps of the whole row
template<int NB>
__device__ __forceinline__ void sweep_fwd(double (&t)[NB], const Factor<NB> &f)
{
  if constexpr (NB == 1) {
    chain_fwd(t[0], f.D0);
  } else {
    Pairs<0>::fwd(t[0], t[1], f.F10, f.D0);
    chain_fwd(t[1], f.D1);
  }
}
template<int NB>
__device__ __forceinline__ void sweep_bwd(double (&t)[NB], const Factor<NB> &f)
{
  if constexpr (NB == 1) {
    chain_bwd(t[0], f.D0);
  } else {
    Pairs<15>::bwd(t[0], t[1], f.B10, f.D1);
    chain_bwd(t[0], f.D0);
  }
}

}  // namespace

template<int NB>
__global__ void __launch_bounds__(64, 2) qp_dense4_kernel(const DenseKernelParams kp, const QpBatch g,
                                                          unsigned *__restrict__ queue, const unsigned batch)
{
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x, row = lane >> 4, cc = lane & 15;
  const int n = kp.n, m = kp.m, k = n + m;
  const Lds4 L = lds4_layout(n, m);

  // ---- per-lane state: block b of this lane is system row 16 b + cc of the row's slot ----
  Factor<NB> f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    f.D0[j] = 0.0;
    if constexpr (NB > 1) {
      f.D1[j]  = 0.0;
      f.F10[j] = 0.0;
      f.B10[j] = 0.0;
    }
  }
  int role[NB], idx[NB];  // 0 none / 1 primal variable / 2 constraint; index in original order
  double ws[NB], zs[NB], rinv[NB], rho[NB], lo[NB], hi[NB], dg[NB];  // ws: x or y;  lo: q (scaled) on x lanes
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    role[b] = 0; idx[b] = 0;
    ws[b] = 0.0; zs[b] = 0.0; rinv[b] = 1.0; rho[b] = 1.0; lo[b] = 0.0; hi[b] = 0.0; dg[b] = 1.0;
  }

  // ---- per-slot bookkeeping (wave-uniform) ----
  int qb[kSlots];
  uint32_t it[kSlots];
#pragma unroll
  for (int s = 0; s < kSlots; ++s) {
    qb[s] = -1;
    it[s] = 0;
  }
  const uint32_t sci   = kp.stop_check_iter;
  const uint32_t maxit = kp.max_iter;
  const bool aligned   = sci >= 2;  // iter % sci == 1 never holds for sci <= 1 (:465)
  uint32_t phase       = 0;         // iteration index of every running slot, mod sci
  bool qempty          = false;

  // scaled iterate of slot s -> its LDS area, then polish / un-scale / report
  auto finish_slot = [&](const int s, const int code, const uint32_t iters) {
    double *cval;
    const Lds S = slot_view(smem, L, n, m, s, cval);
    if (row == s) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (role[b] == 1) S.xv[idx[b]] = ws[b];
        if (role[b] == 2) S.yv[idx[b]] = ws[b];
        role[b] = 0;
        ws[b] = 0.0; zs[b] = 0.0; lo[b] = 0.0;
      }
    }
    wave_sync();
    const double c = cval[0];
    qp_finish(S, kp, n, m, c, (size_t)qb[s], g, lane, code, iters);
  };

  for (;;) {
    // ---- refill empty slots ----
    if (!aligned || phase == 0) {
#pragma unroll
      for (int s = 0; s < kSlots; ++s) {
        while (qb[s] < 0 && !qempty) {
          unsigned nb = 0;
          if (lane == 0) nb = atomicAdd(queue, 1u);
          nb = __builtin_amdgcn_readfirstlane(nb);
          if (nb >= batch) {
            qempty = true;
            break;
          }
          double *cval;
          const Lds S = slot_view(smem, L, n, m, s, cval);
          double c;
          const int rc = qp_setup(S, kp, n, m, (size_t)nb, g, lane, c);
          if (lane == 0) cval[0] = c;
          const bool warm = g.wx != nullptr;
          if (warm) {
            if (lane < n) S.xv[lane] = g.wx[(size_t)nb * n + lane];
            if (lane < m) S.yv[lane] = g.wy[(size_t)nb * m + lane];
          }
          wave_sync();
          if (row == s) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
              const int i      = 16 * b + cc;
              const bool inmat = i < k;
              const int v      = inmat ? S.perm[i] : 0;
              const bool isx   = inmat && v < n;
              const bool isc   = inmat && v >= n;
              const int xi = isx ? v : 0, ci = isc ? v - n : 0;
              role[b] = isx ? 1 : (isc ? 2 : 0);
              idx[b]  = isx ? xi : ci;
              const double sxv = isx ? S.sx[xi] : 1.0;
              const double syv = isc ? S.sy[ci] : 1.0;
              rho[b]  = isc ? S.rho[ci] : 1.0;
              rinv[b] = 1.0 / rho[b];                                             // rho_.cwiseInverse()
              lo[b]   = isx ? c * sxv * S.q[xi] : (isc ? syv * S.l[ci] : 0.0);   // :450 / :473
              hi[b]   = isc ? syv * S.u[ci] : 0.0;                                // :474
              dg[b]   = inmat ? S.W[tri(i, i)] : 1.0;
              ws[b]   = 0.0;
              zs[b]   = 0.0;
              if (warm) {  // :436-445
                if (isx) ws[b] = (1.0 / sxv) * S.xv[xi];
                if (isc) {
                  ws[b]    = c * ((1.0 / syv) * S.yv[ci]);
                  double t = 0.0;
                  for (int j = 0; j < n; ++j) t = fma(syv * S.A[ci + j * m], S.xv[j], t);
                  zs[b] = t;
                }
              }
            }
            // factor blocks (zero outside the factor: padded steps are exact no-ops)
            const int i0 = cc, i1 = 16 + cc;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
              double d0 = 0.0;
              if (j < i0 && i0 < k) d0 = -S.W[tri(i0, j)];
              if (j > i0 && j < k) d0 = -S.W[tri(j, i0)];
              f.D0[j] = d0;
              if constexpr (NB > 1) {
                double d1 = 0.0;
                if (j < cc && i1 < k) d1 = -S.W[tri(i1, 16 + j)];
                if (j > cc && 16 + j < k) d1 = -S.W[tri(16 + j, i1)];
                f.D1[j]  = d1;
                f.F10[j] = (i1 < k) ? -S.W[tri(i1, j)] : 0.0;
                f.B10[j] = (16 + j < k && i0 < k) ? -S.W[tri(16 + j, i0)] : 0.0;
              }
            }
          }
          wave_sync();
          qb[s] = (int)nb;
          it[s] = 0;
          if (rc >= 0 || maxit == 0) {  // ends before the first iteration
            finish_slot(s, rc, 0);
            qb[s] = -1;
          }
        }
      }
    }
    bool any = false, want = false;
    uint32_t rem = 0xFFFFFFFFu;  // iterations until the first running slot reaches max_iter
#pragma unroll
    for (int s = 0; s < kSlots; ++s) {
      if (qb[s] >= 0) {
        any = true;
        rem = min(rem, maxit - it[s]);
      } else {
        want = true;
      }
    }
    if (!any) {
      if (qempty) break;
      phase = 0;  // nothing is running: any iteration can be iteration 0
      continue;
    }
    want = want && !qempty;

    // ---- plain iterations up to the next event: a stopping check (phase == 1), a slot reaching
    //      max_iter, or a refill point.  The tight loop keeps only factor + iterate state live. ----
    uint32_t nplain = rem - 1;  // the iteration in which a slot reaches max_iter is an event iteration
    bool event      = true;
    if (aligned) {
      const uint32_t d_check = (phase <= 1) ? 1 - phase : sci + 1 - phase;
      nplain                 = min(nplain, d_check);
      const uint32_t d_fill  = sci - phase;  // phase 0: this refill point is behind us
      if (want && d_fill <= nplain) {
        nplain = d_fill;
        event  = false;
      }
    } else if (want) {
      nplain = 0;
    }

    // loop constants as fresh values: their live ranges start here, so the register allocator
    // keeps them (and not values of the setup / check code) in registers across the tight loop
    double sigma = kp.sigma, alpha = kp.alpha, alpha_comp = kp.alpha_comp;
    asm volatile("" : "+s"(sigma), "+s"(alpha), "+s"(alpha_comp));
    bool isc[NB], dok[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      isc[b] = role[b] == 2;
      dok[b] = fabs(dg[b]) > DBL_MIN;
    }

    // one ADMM iteration of all four slots :447-477, branch-free: lanes that carry no constraint
    // (primal variables and padding, where ws = lo = 0) take the "x" expressions
    auto iterate = [&](double (&wold)[NB]) {
      double t[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {  // right-hand side :450-451
        const double tx = sigma * ws[b] - lo[b];
        const double tc = zs[b] - rinv[b] * ws[b];
        t[b]            = isc[b] ? tc : tx;
      }
      // K^-1 t :462
      sweep_fwd<NB>(t, f);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const double tq = t[b] / dg[b];
        t[b]            = dok[b] ? tq : 0.0;
      }
      sweep_bwd<NB>(t, f);
#pragma unroll
      for (int b = 0; b < NB; ++b) {  // :470-477
        wold[b]           = ws[b];
        const double base = alpha * t[b] + alpha_comp * ws[b];
        double zn         = alpha * (rinv[b] * t[b]) + alpha_comp * (rinv[b] * ws[b]) + zs[b];
        zn                = (zn < lo[b]) ? lo[b] : zn;  // cwiseMax(sy*l)
        zn                = (hi[b] < zn) ? hi[b] : zn;  // cwiseMin(sy*u)
        const double yn   = base + rho[b] * zs[b] - rho[b] * zn;
        ws[b]             = isc[b] ? yn : base;
        zs[b]             = isc[b] ? zn : zs[b];
      }
    };

    double wold[NB];
    for (uint32_t r = 0; r < nplain; ++r) iterate(wold);
#pragma unroll
    for (int s = 0; s < kSlots; ++s)
      if (qb[s] >= 0) it[s] += nplain;
    if (aligned) phase = (phase + nplain) % sci;
    if (!event) continue;

    // ---- event iteration: stopping check and/or a slot at max_iter ----
    iterate(wold);
    const bool chk = aligned && phase == 1;  // :465
    int res        = -1;
    if (chk) {  // :479-509
      double *cval;
      const Lds S    = slot_view(smem, L, n, m, row, cval);
      const double c = cval[0];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (role[b] == 1) {
          const double sxv = S.sx[idx[b]];
          S.xv[idx[b]]     = sxv * ws[b];
          S.dxus[idx[b]]   = sxv * (ws[b] - wold[b]);
        }
        if (role[b] == 2) {
          const double syv = S.sy[idx[b]];
          S.yv[idx[b]]     = syv * ws[b] / c;
          S.zus[idx[b]]    = (1.0 / syv) * zs[b];
          S.dyus[idx[b]]   = syv * (ws[b] - wold[b]) / c;
        }
      }
      wave_sync();
      res = check_rows<NB>(S, kp, n, m, cc);
      wave_sync();
    }
#pragma unroll
    for (int s = 0; s < kSlots; ++s) {
      if (qb[s] >= 0) {
        ++it[s];
        const int code = chk ? __builtin_amdgcn_readlane(res, 16 * s) : -1;
        if (code >= 0 || it[s] == maxit) {
          finish_slot(s, code, it[s]);
          qb[s] = -1;
        }
      }
    }
    if (aligned) phase = (phase + 1 == sci) ? 0 : phase + 1;
  }
}

size_t qp_dense4_lds_bytes(int n, int m)
{
  const Lds4 L = lds4_layout(n, m);
  return ((size_t)L.shared_doubles + (size_t)kSlots * L.slot_doubles) * sizeof(double);
}

// device-side ticket counters, one per launch in flight (zeroed on the launch's stream)
static unsigned *ticket_pool(int &index)
{
  constexpr int kPool = 256, kMaxDev = 64;
  static unsigned *pool[kMaxDev] = {};
  static unsigned next[kMaxDev]  = {};
  static std::mutex mtx;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return nullptr;
  std::lock_guard<std::mutex> lock(mtx);
  if (!pool[dev]) {
    if (hipMalloc(&pool[dev], kPool * 64) != hipSuccess) return nullptr;  // one counter per 64-byte line
  }
  index = (int)(__atomic_fetch_add(&next[dev], 1u, __ATOMIC_RELAXED) % kPool);
  return pool[dev];
}

hipError_t qp_dense4_launch(const DenseKernelParams &kp, int64_t batch, const QpBatch &g, hipStream_t stream)
{
  static int cus[64] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev >= 0 && dev < 64 && cus[dev] == 0) {
    e = hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
  }
  const int ncu = (dev >= 0 && dev < 64 && cus[dev] > 0) ? cus[dev] : 256;
  int ti = 0;
  unsigned *pool = ticket_pool(ti);
  if (!pool) return hipErrorOutOfMemory;
  unsigned *queue = pool + 16 * ti;
  e = hipMemsetAsync(queue, 0, sizeof(unsigned), stream);
  if (e != hipSuccess) return e;

  const int k      = kp.n + kp.m;
  const size_t lds = qp_dense4_lds_bytes(kp.n, kp.m);
  int per_cu       = (int)((160u * 1024u) / lds);
  static const int occ_cap = getenv("SFB_QP4_WAVES_PER_CU") ? atoi(getenv("SFB_QP4_WAVES_PER_CU")) : 8;  // A/B only
  if (per_cu > occ_cap) per_cu = occ_cap;  // 2 waves per SIMD (VGPR budget of the kernel)
  if (per_cu < 1) per_cu = 1;
  int64_t max_waves = (int64_t)ncu * per_cu;
  if (const char *cap = getenv("SFB_QP4_MAX_WAVES")) {  // tests: small grids exercise the slot refill path
    const int64_t c = atoll(cap);
    if (c >= 1 && c < max_waves) max_waves = c;
  }
  const dim3 grid((unsigned)(batch < max_waves ? batch : max_waves)), block(kWave);
  if (k <= 16) {
    hipLaunchKernelGGL((qp_dense4_kernel<1>), grid, block, lds, stream, kp, g, queue, (unsigned)batch);
  } else {
    hipLaunchKernelGGL((qp_dense4_kernel<2>), grid, block, lds, stream, kp, g, queue, (unsigned)batch);
  }
  return hipGetLastError();
}

}  // namespace sfb
