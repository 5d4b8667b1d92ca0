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
//   - iteration counts differ per QP (2 ... max_iter), so the grid is persistent and slots are refilled from
//     a device-side queue.  The queue is TIME-SLICED: a QP that has held its slot for a slice of iterations
//     while others are waiting goes back to the end of the queue (its iterate and iteration count are in
//     its record anyway) -- round-robin among the long runners, so that they finish together at the end of
//     the launch instead of the last-started one running alone (greedy list scheduling left a third of the
//     benchmark batch's time to that tail).  Problems start / resume only on iterations that are multiples
//     of stop_check_iter, which keeps the stopping checks of the four slots on the same iteration; the
//     check itself runs row-parallel;
//   - the wave-wide parts of a solve do not run in this kernel at all.  A launch is three kernels:
//       setup    one wavefront per QP (massively parallel): scaling, KKT, pivoted LDL' (qp_dense_common.h),
//                then every lane writes the registers of "its" lane of the iterate kernel (factor blocks,
//                scaled bounds, rho, initial iterate) to the QP's record in a per-launch workspace;
//       iterate  four QPs per wavefront, persistent; a refill is ~80 coalesced loads per lane of one
//                row, a finished slot scatters its iterate into the record;
//       finish   one wavefront per QP: polish, un-scale, objective, report.
//     Setup and finish thus keep the parallelism of the one-QP-per-wave kernel (they dominate batches of
//     easy problems), and the lockstep loop never waits for them.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <mutex>

#include "knobs.h"
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

// LDS of the iterate kernel: one area per slot with what the stopping checks need
__host__ __device__ inline int slot4_doubles(int n, int m)
{
  int sl = n * n + m * n + 4 * n + 7 * m + 2;  // P A | q sx xv dxus | l u sy yv zus dyus (+1 m spare) | cval
  // rows of a wave read the same offsets of different slots: stride = 16 mod 32 doubles puts
  // neighbouring slots on complementary LDS banks
  return ((sl + 15) & ~31) + 16;
}
__device__ __forceinline__ Lds slot_view(double *base, const int n, const int m, const int slot, double *&cval)
{
  Lds s;
  double *p = base + slot * slot4_doubles(n, m);
  s.W = nullptr; s.temp = nullptr; s.rho = nullptr; s.perm = nullptr; s.LU = nullptr;  // setup / finish kernels only
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
  s.dxus = p; p += n;
  s.dyus = p; p += m;
  cval   = p;
  return s;
}

// Record of one QP in the launch workspace (doubles).  Lane registers of the iterate kernel are stored
// field-major, [reg][16 lanes], so that a row reads them with coalesced loads:
//   regs 0..15 D0 | (NB = 2: 16..31 D1, 32..47 F10, 48..63 B10) | per block b: ws zs rinv rho lo hi dg role+4 idx
struct Rec4 {
  int nreg, reg_state, off_sx, off_sy, off_c, off_xs, off_ys, off_code, off_iter, size;
};
template<int NB>
__host__ __device__ inline Rec4 rec4_layout(int n, int m)
{
  Rec4 r;
  r.reg_state = (NB > 1) ? 64 : 16;
  r.nreg      = r.reg_state + 8 * NB;
  int o       = 16 * r.nreg;
  r.off_sx = o; o += n;
  r.off_sy = o; o += m;
  r.off_c  = o; o += 1;
  r.off_xs = o; o += n;   // scaled iterate, original order: initial (setup), final (iterate)
  r.off_ys = o; o += m;
  r.off_code = o; o += 1;  // -1: still to be iterated / ran into max_iter
  r.off_iter = o; o += 1;
  r.size = (o + 1) & ~1;
  return r;
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

  // PRIMAL INFEASIBILITY :598-621.  The verdict is max(|A'dy|, certificate sum) < thr; the expensive A'dy is
  // only formed when the certificate sum does not already rule the verdict out (same result, NaN included).
  if (res < 0) {
    double a2 = 0.0;
#pragma unroll
    for (int b = 0; b < NB; ++b) a2 = fmax(a2, lm[b] ? fabs(s.dyus[cc + 16 * b]) : 0.0);
    const double Edy_norm = row_max16(a2);
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
    if (!(acc >= thr)) {
      double a1 = 0.0;
#pragma unroll
      for (int b = 0; b < NB; ++b) a1 = fmax(a1, ln[b] ? fabs(row_At(s, n, m, cc + 16 * b, s.dyus)) : 0.0);
      const double Aty_norm = row_max16(a1);
      const double mxv      = (Aty_norm < acc) ? acc : Aty_norm;  // std::max(a,b) = (a<b)?b:a
      if (mxv < thr) res = SFB_QP_PRIMAL_INFEASIBLE;
    }
  }

  // DUAL INFEASIBILITY :625-641: |P dx| <= thr, q'dx <= thr and the row conditions on A dx; evaluated cheapest
  // first, the later ones only while the verdict is still open.
  if (res < 0) {
    double a1 = 0.0, a2 = 0.0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int i = cc + 16 * b;
      a1          = fmax(a1, ln[b] ? fabs(s.dxus[i]) : 0.0);
      a2          = fmax(a2, ln[b] ? fabs(row_P(s, n, i, s.dxus)) : 0.0);
    }
    const double dx_norm = row_max16(a1), Pdx_n = row_max16(a2);
    const double thr     = kp.eps_dinf * dx_norm;
    if (Pdx_n <= thr) {
      double qdx = 0.0;
      for (int j = 0; j < n; ++j) qdx = fma(s.q[j], s.dxus[j], qdx);
      if (qdx <= thr) {
        double viol = 0.0;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          if (lm[b]) {
            const int i      = cc + 16 * b;
            const double Adx = row_A(s, n, m, i, s.dxus), ui = s.u[i], li = s.l[i];
            bool rowok;
            if (ui == inf) {
              rowok = Adx >= -thr;
            } else if (li == -inf) {
              rowok = Adx <= thr;
            } else {
              rowok = fabs(Adx) < thr;
            }
            if (!rowok) viol = 1.0;
          }
        }
        if (row_max16(viol) == 0.0) res = SFB_QP_DUAL_INFEASIBLE;
      }
    }
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

// ---- kernel 1: setup, one wavefront per QP ----
template<int NB>
__global__ void __launch_bounds__(64) qp_dense4_setup_kernel(const DenseKernelParams kp, const QpBatch g,
                                                             double *__restrict__ wsp)
{
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  const int n = kp.n, m = kp.m, k = n + m;
  const size_t b = blockIdx.x;
  Lds S          = carve(smem, n, m, k);
  const Rec4 R   = rec4_layout<NB>(n, m);
  double *rec    = wsp + b * (size_t)R.size;
  double c;
  const int rc    = qp_setup(S, kp, n, m, b, g, lane, c);
  const bool warm = g.wx != nullptr;
  if (warm) {
    if (lane < n) S.xv[lane] = g.wx[b * n + lane];
    if (lane < m) S.yv[lane] = g.wy[b * m + lane];
  }
  wave_lds_fence();
  if (lane < 16 * NB) {  // lane = system row i = 16 blk + cc  ->  registers of lane cc, block blk
    const int blk = lane >> 4, cc = lane & 15, i = lane;
    const bool inmat = i < k;
    const int v      = inmat ? S.perm[i] : 0;
    const bool isx   = inmat && v < n;
    const bool isc   = inmat && v >= n;
    const int xi = isx ? v : 0, ci = isc ? v - n : 0;
    const double sxv = isx ? S.sx[xi] : 1.0;
    const double syv = isc ? S.sy[ci] : 1.0;
    const double rho = isc ? S.rho[ci] : 1.0;
    double ws = 0.0, zs = 0.0;
    if (warm) {  // :436-445
      if (isx) ws = (1.0 / sxv) * S.xv[xi];
      if (isc) {
        ws       = c * ((1.0 / syv) * S.yv[ci]);
        double t = 0.0;
        for (int j = 0; j < n; ++j) t = fma(syv * S.A[ci + j * m], S.xv[j], t);
        zs = t;
      }
    }
    double *lr = rec + cc;
    double *st = lr + (R.reg_state + 8 * blk) * 16;
    st[0 * 16] = ws;
    st[1 * 16] = zs;
    st[2 * 16] = 1.0 / rho;                                                     // rho_.cwiseInverse()
    st[3 * 16] = rho;
    st[4 * 16] = isx ? c * sxv * S.q[xi] : (isc ? syv * S.l[ci] : 0.0);        // :450 / :473
    st[5 * 16] = isc ? syv * S.u[ci] : 0.0;                                     // :474
    st[6 * 16] = inmat ? S.W[tri(i, i)] : 1.0;
    st[7 * 16] = (double)((isx ? 1 : (isc ? 2 : 0)) + 4 * (isx ? xi : ci));
    // factor blocks (zero outside the factor: padded steps are exact no-ops), see struct Factor
    const int i0 = cc, i1 = 16 + cc;
    for (int j = 0; j < 16; ++j) {
      if (blk == 0) {
        double d0 = 0.0;
        if (j < i0 && i0 < k) d0 = -S.W[tri(i0, j)];
        if (j > i0 && j < k) d0 = -S.W[tri(j, i0)];
        lr[j * 16] = d0;
        if constexpr (NB > 1) lr[(48 + j) * 16] = (16 + j < k && i0 < k) ? -S.W[tri(16 + j, i0)] : 0.0;  // B10
      } else {
        double d1 = 0.0;
        if (j < cc && i1 < k) d1 = -S.W[tri(i1, 16 + j)];
        if (j > cc && 16 + j < k) d1 = -S.W[tri(16 + j, i1)];
        lr[(16 + j) * 16] = d1;
        lr[(32 + j) * 16] = (i1 < k) ? -S.W[tri(i1, j)] : 0.0;  // F10
      }
    }
  }
  if (lane < n) {
    rec[R.off_sx + lane] = S.sx[lane];
    rec[R.off_xs + lane] = warm ? (1.0 / S.sx[lane]) * S.xv[lane] : 0.0;
  }
  if (lane < m) {
    rec[R.off_sy + lane] = S.sy[lane];
    rec[R.off_ys + lane] = warm ? c * ((1.0 / S.sy[lane]) * S.yv[lane]) : 0.0;
  }
  if (lane == 0) {
    rec[R.off_c]    = c;
    rec[R.off_code] = (double)rc;
    rec[R.off_iter] = 0.0;
  }
}

// ---- kernel 2: the ADMM loops, four QPs per wavefront, persistent ----
template<int NB>
__global__ void __launch_bounds__(64, 2) qp_dense4_iterate_kernel(const DenseKernelParams kp, const QpBatch g,
                                                                  double *__restrict__ wsp,
                                                                  unsigned *__restrict__ queue, const unsigned batch,
                                                                  const unsigned slice_checks)
{
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x, row = lane >> 4, cc = lane & 15;
  const int n = kp.n, m = kp.m;
  const Rec4 R = rec4_layout<NB>(n, m);

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
  // Queue (zeroed by the launcher).  FRESH QPs are handed out in index order by a ticket counter; SUSPENDED ones
  // go through a ring of ring_n = 2 * batch entries: push j writes (j+1) << 32 | id+1 into ring[j % ring_n] (after the previous user of that
  // entry has cleared it), pop ticket t -- also a plain atomicAdd, wait-free -- may be ahead of the pushes: the
  // slot then keeps its ticket and polls its ring entry at the following refill points.
  unsigned *const q_fresh = queue, *const q_rhead = queue + 16, *const q_tail = queue + 32, *const q_done = queue + 48;
  unsigned long long *const ring = reinterpret_cast<unsigned long long *>(queue + 64);
  // (twice the number of QPs: a push can then never land on the entry of a ticket its own wave still holds --
  //  entries are only cleared by the holder of their ticket, which with batch entries could make a wave wait for itself)
  const unsigned ring_n = 2u * batch;
  constexpr unsigned kNone = 0xFFFFFFFFu;
  uint32_t it0[kSlots];    // iteration count at which the slot took its QP
  unsigned pend[kSlots];   // ring ticket the (empty) slot is waiting for
#pragma unroll
  for (int s = 0; s < kSlots; ++s) {
    it0[s]  = 0;
    pend[s] = kNone;
  }
  bool fresh_left      = true;
  unsigned local_done  = 0;  // QPs this wave has finished since its last report to q_done
  // iterations a QP may hold a slot while others wait (a multiple of the check interval)
  const uint32_t slice = aligned ? slice_checks * sci : slice_checks * 25u;

  // scaled iterate of slot s -> its record (original order), status and iteration count
  auto finish_slot = [&](const int s, const int code, const uint32_t iters) {
    double *rec = wsp + (size_t)qb[s] * (size_t)R.size;
    if (row == s) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (role[b] == 1) rec[R.off_xs + idx[b]] = ws[b];
        if (role[b] == 2) rec[R.off_ys + idx[b]] = ws[b];
        role[b] = 0;
        ws[b] = 0.0; zs[b] = 0.0; lo[b] = 0.0;
      }
      if (cc == 0) {
        rec[R.off_code] = (double)code;
        rec[R.off_iter] = (double)iters;
      }
    }
    ++local_done;
  };
  // slot s gives its QP back to the queue: iterate and iteration count -> record, id -> ring
  auto suspend_slot = [&](const int s) {
    double *rec = wsp + (size_t)qb[s] * (size_t)R.size;
    if (row == s) {
      double *lr = rec + cc;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        double *st = lr + (R.reg_state + 8 * b) * 16;
        st[0 * 16] = ws[b];
        st[1 * 16] = zs[b];
        role[b] = 0;
        ws[b] = 0.0; zs[b] = 0.0; lo[b] = 0.0;
      }
      if (cc == 0) rec[R.off_iter] = (double)it[s];
    }
    __threadfence();  // the record is complete (device scope) before the id can be popped
    if (lane == 0) {
      const unsigned j           = atomicAdd(q_tail, 1u);
      unsigned long long *slot   = ring + (j % ring_n);
      const unsigned long long v = ((unsigned long long)(j + 1u) << 32) | ((unsigned)qb[s] + 1u);
      while (atomicCAS(slot, 0ull, v) != 0ull) __builtin_amdgcn_s_sleep(1);
    }
    qb[s] = -1;
  };
  for (;;) {
    // ---- refill point: hand long runners back if others are waiting, then take QPs for the empty slots,
    //      every row loads its own ----
    if (!aligned || phase == 0) {
      bool due = false;
#pragma unroll
      for (int s = 0; s < kSlots; ++s) due = due || (qb[s] >= 0 && it[s] - it0[s] >= slice);
      if (due) {
        unsigned waiting = 0;
        if (lane == 0) {
          const unsigned fr = __hip_atomic_load(q_fresh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned h  = __hip_atomic_load(q_rhead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned p  = __hip_atomic_load(q_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          waiting = (fr < batch || (int)(p - h) > 0) ? 1u : 0u;  // QPs that have no slot
        }
        if (__builtin_amdgcn_readfirstlane(waiting)) {
#pragma unroll
          for (int s = 0; s < kSlots; ++s)
            if (qb[s] >= 0 && it[s] - it0[s] >= slice) suspend_slot(s);
        } else {
#pragma unroll
          for (int s = 0; s < kSlots; ++s)
            if (qb[s] >= 0 && it[s] - it0[s] >= slice) it0[s] = it[s];  // nobody waits: a fresh slice
        }
      }
      for (;;) {
        int tk[kSlots];
        int mine = -1;  // QP of this lane's row, if the row is being refilled
        bool got_any = false, resumed = false;
        unsigned nfree = 0;
#pragma unroll
        for (int s = 0; s < kSlots; ++s) {
          tk[s] = -1;
          nfree += (qb[s] < 0 && pend[s] == kNone) ? 1u : 0u;
        }
        unsigned base = batch;
        if (fresh_left && nfree != 0) {
          if (lane == 0) base = atomicAdd(q_fresh, nfree);
          base = __builtin_amdgcn_readfirstlane(base);
        }
#pragma unroll
        for (int s = 0; s < kSlots; ++s) {
          if (qb[s] < 0) {
            if (pend[s] == kNone) {
              if (base < batch) {
                tk[s] = (int)base++;
              } else {  // no fresh QP left: queue up for a suspended one
                fresh_left = false;
                unsigned t = 0;
                if (lane == 0) t = atomicAdd(q_rhead, 1u);
                pend[s] = __builtin_amdgcn_readfirstlane(t);
              }
            }
            if (pend[s] != kNone) {
              unsigned long long e = 0;
              if (lane == 0) e = __hip_atomic_load(ring + (pend[s] % ring_n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)e);
              const unsigned hi32 = __builtin_amdgcn_readfirstlane((unsigned)(e >> 32));
              if (hi32 == pend[s] + 1u) {  // my entry has arrived
                if (lane == 0) __hip_atomic_store(ring + (pend[s] % ring_n), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                tk[s]   = (int)(lo32 - 1u);
                pend[s] = kNone;
                resumed = true;
              }
            }
            got_any = got_any || tk[s] >= 0;
          }
          if (row == s) mine = tk[s];
        }
        if (!got_any) break;
        if (resumed) __threadfence();  // records of resumed QPs were written by other waves
        bool accepted = false;
        uint32_t my_it = 0;
        if (mine >= 0) {
          const double *rec = wsp + (size_t)mine * (size_t)R.size;
          // a QP that ended in setup (pre-check, failed factorisation) is reported by the finish kernel
          accepted = rec[R.off_code] < 0.0;
          if (accepted) {
            my_it = (uint32_t)rec[R.off_iter];
            const double *lr = rec + cc;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
              f.D0[j] = lr[j * 16];
              if constexpr (NB > 1) {
                f.D1[j]  = lr[(16 + j) * 16];
                f.F10[j] = lr[(32 + j) * 16];
                f.B10[j] = lr[(48 + j) * 16];
              }
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
              const double *st = lr + (R.reg_state + 8 * b) * 16;
              ws[b]   = st[0 * 16];
              zs[b]   = st[1 * 16];
              rinv[b] = st[2 * 16];
              rho[b]  = st[3 * 16];
              lo[b]   = st[4 * 16];
              hi[b]   = st[5 * 16];
              dg[b]   = st[6 * 16];
              const int ri = (int)st[7 * 16];
              role[b] = ri & 3;
              idx[b]  = ri >> 2;
            }
            // what the stopping checks need, in the slot's LDS area (the 16 lanes of the row copy)
            double *cval;
            const Lds S     = slot_view(smem, n, m, row, cval);
            const size_t nb = (size_t)mine;
            const double *P = g.P + nb * (size_t)(n * n), *A = g.A + nb * (size_t)(m * n);
            for (int e = cc; e < n * n; e += 16) const_cast<double *>(S.P)[e] = P[e];
            for (int e = cc; e < m * n; e += 16) const_cast<double *>(S.A)[e] = A[e];
            for (int e = cc; e < n; e += 16) {
              S.q[e]  = g.q[nb * n + e];
              S.sx[e] = rec[R.off_sx + e];
            }
            for (int e = cc; e < m; e += 16) {
              S.l[e]  = g.l[nb * m + e];
              S.u[e]  = g.u[nb * m + e];
              S.sy[e] = rec[R.off_sy + e];
            }
            if (cc == 0) cval[0] = rec[R.off_c];
          }
        }
        wave_lds_fence();
        const unsigned long long acc = wave_ballot(accepted);
#pragma unroll
        for (int s = 0; s < kSlots; ++s) {
          if (tk[s] >= 0) {
            if ((acc >> (16 * s)) & 1ull) {
              qb[s]  = tk[s];
              it[s]  = (uint32_t)__builtin_amdgcn_readlane((int)my_it, 16 * s);
              it0[s] = it[s];
            } else {
              ++local_done;  // ended in the setup kernel
            }
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
      unsigned d = 0;
      if (lane == 0) {  // report (one atomic per idle period, not per QP), then look at the total
        if (local_done != 0) atomicAdd(q_done, local_done);
        d = __hip_atomic_load(q_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      local_done = 0;
      if (__builtin_amdgcn_readfirstlane(d) >= batch) break;  // every QP has left the loop
      __builtin_amdgcn_s_sleep(8);                             // others still hold QPs that may come back
      phase = 0;  // nothing is running: any iteration can be iteration 0
      continue;
    }
    // come back at the next refill point if a slot is empty or a slice runs out before the next event
#pragma unroll
    for (int s = 0; s < kSlots; ++s) want = want || (qb[s] >= 0 && it[s] - it0[s] + sci >= slice);

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
      const Lds S    = slot_view(smem, n, m, row, cval);
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
      wave_lds_fence();
      res = check_rows<NB>(S, kp, n, m, cc);
      wave_lds_fence();
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

// ---- kernel 3: polish, un-scale, report; one wavefront per QP ----
template<int NB>
__global__ void __launch_bounds__(64) qp_dense4_finish_kernel(const DenseKernelParams kp, const QpBatch g,
                                                              const double *__restrict__ wsp)
{
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  const int n = kp.n, m = kp.m, k = n + m;
  const size_t b    = blockIdx.x;
  const Lds S       = carve(smem, n, m, k);
  const Rec4 R      = rec4_layout<NB>(n, m);
  const double *rec = wsp + b * (size_t)R.size;
  {
    const double *P = g.P + b * (size_t)(n * n), *A = g.A + b * (size_t)(m * n);
    for (int e = lane; e < n * n; e += kWave) const_cast<double *>(S.P)[e] = P[e];
    for (int e = lane; e < m * n; e += kWave) const_cast<double *>(S.A)[e] = A[e];
    if (lane < n) {
      S.q[lane]  = g.q[b * n + lane];
      S.sx[lane] = rec[R.off_sx + lane];
      S.xv[lane] = rec[R.off_xs + lane];
    }
    if (lane < m) {
      S.l[lane]  = g.l[b * m + lane];
      S.u[lane]  = g.u[b * m + lane];
      S.sy[lane] = rec[R.off_sy + lane];
      S.yv[lane] = rec[R.off_ys + lane];
    }
  }
  const double c      = rec[R.off_c];
  const int code      = (int)rec[R.off_code];
  const uint32_t iter = (uint32_t)rec[R.off_iter];
  wave_lds_fence();
  qp_finish(S, kp, n, m, c, b, g, lane, code, iter);
}

size_t qp_dense4_lds_bytes(int n, int m) { return (size_t)kSlots * slot4_doubles(n, m) * sizeof(double); }

template<int NB>
static hipError_t launch4(const DenseKernelParams &kp, int64_t batch, const QpBatch &g, int ncu, hipStream_t stream, void *workspace)
{
  // per-launch workspace: QP records, then the queue (4 counters on their own cache lines + ring of `batch`
  // 8-byte entries); stream-ordered allocation (no device synchronisation in the steady state)
  const Rec4 R        = rec4_layout<NB>(kp.n, kp.m);
  const size_t rbytes = (size_t)batch * (size_t)R.size * sizeof(double);
  const size_t qbytes = 64 * sizeof(unsigned) + 2 * (size_t)batch * sizeof(unsigned long long);
  const size_t bytes  = rbytes + qbytes;
  double *wsp        = static_cast<double *>(workspace);  // the caller's (sfb_workspace), or per-launch below
  const bool owned   = workspace == nullptr;
  bool async_alloc   = true;
  hipError_t e       = hipSuccess;
  if (owned) {
    e = hipMallocAsync(reinterpret_cast<void **>(&wsp), bytes, stream);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      async_alloc = false;
      e           = hipMalloc(reinterpret_cast<void **>(&wsp), bytes);
      if (e != hipSuccess) return e;
    }
  }
  unsigned *queue = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(wsp) + rbytes);
  e               = hipMemsetAsync(queue, 0, qbytes, stream);
  if (e != hipSuccess) {
    if (!owned) return e;
    if (async_alloc) (void)hipFreeAsync(wsp, stream);
    else (void)hipFree(wsp);
    return e;
  }
  const dim3 block(kWave), full((unsigned)batch);
  const size_t lds1 = qp_dense_lds_bytes(kp.n, kp.m);
  hipLaunchKernelGGL((qp_dense4_setup_kernel<NB>), full, block, lds1, stream, kp, g, wsp);

  const size_t lds = qp_dense4_lds_bytes(kp.n, kp.m);
  int per_cu       = (int)((160u * 1024u) / lds);
  const int occ_cap = NB > 1 ? 8 : 16;
  if (per_cu > occ_cap) per_cu = occ_cap;  // waves per SIMD the VGPR budget of the kernel allows: 2 (NB = 2) / 4
  if (per_cu < 1) per_cu = 1;
  int64_t max_waves = (int64_t)ncu * per_cu;
  if (const char *cap = sfb::knob("SFB_QP4_MAX_WAVES")) {  // tests: small grids exercise the slot refill path
    const int64_t c = atoll(cap);
    if (c >= 1 && c < max_waves) max_waves = c;
  }
  const dim3 grid((unsigned)(batch < max_waves ? batch : max_waves));
  const unsigned slice_checks = 80;  // time slice in stopping-check intervals (2 000 iterations at the default 25)
  if (kp.max_iter != 0)  // nothing to iterate otherwise: the finish kernel reports the initial iterate
    hipLaunchKernelGGL((qp_dense4_iterate_kernel<NB>), grid, block, lds, stream, kp, g, wsp, queue, (unsigned)batch,
                       slice_checks);
  hipLaunchKernelGGL((qp_dense4_finish_kernel<NB>), full, block, lds1, stream, kp, g, wsp);
  e = hipGetLastError();
  if (!owned) return e;
  if (async_alloc) {
    const hipError_t e2 = hipFreeAsync(wsp, stream);
    if (e == hipSuccess) e = e2;
  } else {
    (void)hipStreamSynchronize(stream);
    (void)hipFree(wsp);
  }
  return e;
}

size_t qp_dense4_ws_bytes(int n, int m, int64_t batch)
{
  const size_t rsize = (n + m <= 16) ? (size_t)rec4_layout<1>(n, m).size : (size_t)rec4_layout<2>(n, m).size;
  return (size_t)batch * rsize * sizeof(double) + 64 * sizeof(unsigned) + 2 * (size_t)batch * sizeof(unsigned long long);
}

hipError_t qp_dense4_launch(const DenseKernelParams &kp, int64_t batch, const QpBatch &g, hipStream_t stream, void *workspace)
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
  return (kp.n + kp.m <= 16) ? launch4<1>(kp, batch, g, ncu, stream, workspace) : launch4<2>(kp, batch, g, ncu, stream, workspace);
}

}  // namespace sfb
