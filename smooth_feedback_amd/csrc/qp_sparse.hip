// Batched sparse ADMM QP solver for gfx950: all items share ONE sparsity pattern (e.g. a swarm of
// MPC problems produced by the same ocp_to_qp transcription), one wavefront per item.
//
// Replaces, per item, QPSolver<QuadraticProgramSparse<double>>::solve (reference
// qp_solver.hpp:343-568 sparse branches :379-397, :423-426, :452-460; scale :673-730;
// check_stopping :574-644; polish :92-204).  The symbolic work (elimination order, pattern of L)
// is done once on the host (sparse_plan.cpp) and shared; this kernel does everything numeric.
// Arithmetic follows oracle/qp_sparse_oracle.c operation for operation (bit-identical results):
//   - left-looking numeric LDL': column j gathers its source columns kk < j in ascending order;
//   - forward sweep column-oriented, D^-1 as reciprocal-multiply (:458), backward sweep pushing row
//     by row from the row-major copy of L;
//   - polish on the reduced system embedded in the full pattern.
//
// Data placement: the dense work/solution vector (k = n+m doubles) lives in LDS; the factor
// (2 x nnz(L) doubles: column-major for the forward sweep, row-major for the backward sweep) and
// the ADMM vectors live in a per-item HBM workspace and are streamed with coalesced accesses.  The
// factor does not fit on chip for MPC-sized problems (k ~ 1.5k), so every ADMM iteration streams
// it once in each direction: this kernel is HBM-bound by construction.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <map>
#include <mutex>

#include <algorithm>
#include <type_traits>
#include <utility>
#include <cfloat>
#include <cmath>

#include "knobs.h"
#include "../../include/sfb.h"
#include "qp_sparse_kernel.h"
#include "wave_util.h"

#ifndef SFB_SWEEP_ASM
#define SFB_SWEEP_ASM 1  // 0: the compiler-scheduled units for every form of the sweeps (A/B builds)
#endif
#ifndef SFB_SWEEP_DEPTH
#define SFB_SWEEP_DEPTH 8  // units (2 slots per lane each) in flight per sweep
#endif

namespace sfb {

namespace {

// A value every lane holds identically (loaded from the shared plan), as a scalar: loop bounds and branch conditions
// built from it then compile to scalar compares and branches instead of exec-masked ones.
__device__ __forceinline__ int uni(const int v) { return __builtin_amdgcn_readfirstlane(v); }

struct Ws {
  double *Lx, *LxF, *LxB, *D, *Dinv, *tv;  // Lx, D and one scratch double are contiguous (accumulators)
  double *sx, *qc, *xs, *xus, *dxus;
  double *sy, *rho, *rinv, *lo, *hi, *ys, *zs, *yus, *zus, *dyus, *act;
  double *hdr;  // 16 doubles: [0..5] state of a suspended item (time-sliced launches): c, iter, next_chk, t0;
                //             [6..9] what the ADMM factor in this workspace belongs to (reuse_factor, see kHdr*)
  double *Axc;  // pruned plans: the item's kept entries of A, compacted (qp_sparse_kernel.h)
};
// reuse_factor: the header of an item whose workspace holds a successful ADMM factorisation
constexpr int kHdrStamp = 6, kHdrC = 7, kHdrSigma = 8, kHdrScaling = 9;
// phased launches (see qp_sparse_launch): [10] status after the setup / ADMM phase (-1 = open), [11] 1.0 = the item was
// solved completely in the fallback pool during the setup phase (the later phases skip it), [12] timeline stamp
constexpr int kHdrCode = 10, kHdrComplete = 11;
[[maybe_unused]] constexpr int kHdrTl2 = 12;
// what one launch does with an item
enum { PH_SETUP = 1, PH_ADMM = 2, PH_FINISH = 3 };
// A launch performs the phases ph0 .. ph1 of every item it visits (packed into one kernel argument together with the
// pause point, see qp_sparse_launch): [SETUP, FINISH] = everything.
constexpr int phases_pack(int ph0, int ph1, unsigned pause_at = 0, int dq_mode = 0)
{
  return ph0 | (ph1 << 4) | (int)((pause_at & 0xFFFFFu) << 8) | (dq_mode << 28);  // dq_mode: ring of finished items, see the kernel
}
constexpr int PH_EVERYTHING = phases_pack(PH_SETUP, PH_FINISH);
constexpr unsigned long long kFactorStamp = 0x5FB0FAC7A11CE5EDull;

__device__ __forceinline__ Ws carve_ws(double *base, int n, int m, int nnzL, int funits, int bunits)
{
  const int k = n + m;
  Ws w;
  double *p = base;
  w.Lx = p; p += nnzL;
  w.D = p; p += k;
  p += 2;  // scratch accumulator of the padding slots, always-zero accumulator
  w.LxF = p; p += (size_t)(funits + kSweepPadDev) * 128;
  w.LxB = p; p += (size_t)(bunits + kSweepPadDev) * 128;
  w.Dinv = p; p += k;   w.tv = p; p += k;
  w.sx = p; p += n;     w.qc = p; p += n;     w.xs = p; p += n;   w.xus = p; p += n;  w.dxus = p; p += n;
  p += n;
  w.sy = p; p += m;     w.rho = p; p += m;    w.rinv = p; p += m; w.lo = p; p += m;   w.hi = p; p += m;
  w.ys = p; p += m;     w.zs = p; p += m;     w.yus = p; p += m;  w.zus = p; p += m;  w.dyus = p; p += m;
  w.act = p; p += m;
  w.hdr = p; p += 16;
  w.Axc = p;
  return w;
}
// the same workspace with the factor fields redirected to the second factor block (polish of a reuse_factor call)
__device__ __forceinline__ Ws polish_ws(const Ws &w, double *base, const size_t off, int n, int m, int nnzL, int funits, int bunits)
{
  const int k = n + m;
  Ws v = w;
  double *p = base + off;
  v.Lx = p; p += nnzL;
  v.D = p; p += k;
  p += 2;
  v.LxF = p; p += (size_t)(funits + kSweepPadDev) * 128;
  v.LxB = p; p += (size_t)(bunits + kSweepPadDev) * 128;
  v.Dinv = p;
  return v;
}

struct Item {
  const double *Px, *q, *Ax, *l, *u;
};

// KKT fill: every entry of the permuted lower pattern goes to its accumulator.  mode 0: ADMM matrix
// (qp_solver.hpp:382-395); mode 1: polish matrix H + diag(delta, -delta), inactive rows zeroed (:143-171).
// Batched: the descriptors {kind, idx, row, col} of U entries per lane are fetched together, then everything they
// point to (scaling factors, the P / A / rho value), then the values are formed and stored -- two memory round
// trips per U entries instead of four per entry.  [p_begin, p_end): entries of the descriptor array `desc4` /
// accumulator map `map` to fill; entries past p_end are fetched (the arrays are padded) but not stored.
typedef int vint4k __attribute__((ext_vector_type(4)));
template<int U>
__device__ __forceinline__ void kkt_fill(const int32_t *__restrict__ desc4, const int32_t *__restrict__ map, const int p_begin,
                                         const int p_end, const Item &it, const Ws &w, double *ACC, const int mode,
                                         const double c, const double sigma, const double delta, const int lane)
{
  const vint4k *desc = reinterpret_cast<const vint4k *>(desc4);
  // (the descriptors of batch b + 1 are requested before the entries batch b points to are gathered: one memory round trip per
  //  batch on the dependent path instead of two)
  vint4k dn[U];
  int mn[U];
  auto fetch = [&](const int p0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool on = p0 + u * kWave < p_end;
      dn[u] = desc[on ? p0 + u * kWave : p_begin];
      mn[u] = map[on ? p0 + u * kWave : p_begin];
    }
  };
  if (p_begin + lane < p_end) fetch(p_begin + lane);
  for (int p0 = p_begin + lane; p0 < p_end; p0 += kWave * U) {
    vint4k d[U];
    int mp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      d[u]  = dn[u];
      mp[u] = mn[u];
    }
    double s1[U], s2[U], val[U], ac[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool on  = p0 + u * kWave < p_end;
      const int kind = on ? d[u].x : K_SIGMA, idx = d[u].y, r = d[u].z, cc = d[u].w;
      d[u].x         = kind;
      const bool isP = kind == K_P, isA = kind == K_A;
      s1[u]  = (isP || isA) ? (isP ? w.sx : w.sy)[r] : 0.0;
      s2[u]  = (isP || isA) ? w.sx[cc] : 0.0;
      val[u] = isP ? it.Px[idx] : (isA ? it.Ax[idx] : (kind == K_RHO ? w.rho[idx] : 0.0));
      ac[u]  = (isA && mode != 0) ? w.act[r] : 1.0;
    }
    if (p0 + kWave * U < p_end) fetch(p0 + kWave * U);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kind = d[u].x, r = d[u].z, cc = d[u].w;
      double v;
      if (kind == K_P) {
        if (mode == 0) v = c * s1[u] * s2[u] * val[u];  // :385   c sx[r] sx[c] P
        else v = c * s2[u] * s1[u] * val[u];            // :145   c sx[c] sx[r] P
        if (r == cc) v += (mode == 0) ? sigma : delta;  // :389 / :170
      } else if (kind == K_A) {
        v = s1[u] * s2[u] * val[u];  // :392 / :154   sy[r] sx[c] A
        if (ac[u] == 0.0) v = 0.0;
      } else if (kind == K_SIGMA) {
        v = (mode == 0) ? sigma : 0.0 + delta;
      } else {
        v = (mode == 0) ? (-1.0 / val[u]) : 0.0 - delta;  // :395 / :171
      }
      if (p0 + u * kWave < p_end) ACC[mp[u]] = v;
    }
  }
}

// One block of DEPTH trailing accumulators per lane: v = fma(-L(a, j), L(b, j) D(j), v) over the columns j of the
// supernode in ascending order, the DEPTH chains advancing together.  PIPE: the LDS reads of column j + 1 are issued
// before the fmas of column j (2 DEPTH reads, then DEPTH dependent fmas per column otherwise); costs 2 DEPTH more
// registers.
template<int DEPTH, bool PIPE>
__device__ __forceinline__ void trailing_block(const double *pan, const double *mul, const int R, const int wd,
                                               const int (&tp)[DEPTH], const unsigned (&ab)[DEPTH], double (&v)[DEPTH],
                                               double *ACC)
{
  int ra[DEPTH], rb[DEPTH];
#pragma unroll
  for (int dd = 0; dd < DEPTH; ++dd) {
    ra[dd] = wd + (int)(ab[dd] & 0xFFFFu);
    rb[dd] = wd + (int)(ab[dd] >> 16);
  }
  if constexpr (PIPE) {
    double a[DEPTH], b[DEPTH];
#pragma unroll
    for (int dd = 0; dd < DEPTH; ++dd) {
      a[dd] = pan[ra[dd]];
      b[dd] = mul[rb[dd]];
    }
    for (int jj = 0; jj < wd; ++jj) {
      const int jn = (jj + 1 < wd) ? jj + 1 : jj;  // (the last column is read once more: no branch in the loop)
      double an[DEPTH], bn[DEPTH];
#pragma unroll
      for (int dd = 0; dd < DEPTH; ++dd) {
        an[dd] = pan[jn * R + ra[dd]];
        bn[dd] = mul[jn * R + rb[dd]];
      }
#pragma unroll
      for (int dd = 0; dd < DEPTH; ++dd) v[dd] = fma(-a[dd], b[dd], v[dd]);
#pragma unroll
      for (int dd = 0; dd < DEPTH; ++dd) {
        a[dd] = an[dd];
        b[dd] = bn[dd];
      }
    }
  } else {
    for (int jj = 0; jj < wd; ++jj) {
      double a[DEPTH], b[DEPTH];
#pragma unroll
      for (int dd = 0; dd < DEPTH; ++dd) {
        a[dd] = pan[jj * R + ra[dd]];
        b[dd] = mul[jj * R + rb[dd]];
      }
#pragma unroll
      for (int dd = 0; dd < DEPTH; ++dd) v[dd] = fma(-a[dd], b[dd], v[dd]);
    }
  }
#pragma unroll
  for (int dd = 0; dd < DEPTH; ++dd) ACC[tp[dd]] = v[dd];
}

constexpr int kPanelCols = 16;  // == the plan's widest supernode (sparse_plan.cpp kMaxWidth)

// Elimination of a panel of at most 64 rows in REGISTERS: lane r holds row r, one register per member column.
// Pivots and multipliers travel by v_readlane broadcasts; no LDS round trip and no fence on the dependent chain.
// Per entry: divide by D, multiply back, fma(-L(ra, j), L(rb, j) D(j), .) in ascending j.  Entries above the
// diagonal hold garbage and are never used.  Stages L(., j) in pan and L(., j) D(j) in mul for the trailing
// updates.  W = width class (wd <= W): the updates of the later columns run branch-free over all W registers --
// registers beyond wd belong to no column, collect garbage and are never stored -- so that the scheduler can
// overlap them with the next column's division chain (one scalar branch per COLUMN is all that is left).
// Returns false on a zero pivot.
template<int W>
__device__ __forceinline__ bool panel_eliminate_w(double (&reg)[kPanelCols], const int wd, const int R, double *pan,
                                                  double *mul, const int lane)
{
#pragma unroll
  for (int jj = 0; jj < W; ++jj) {
    if (jj < wd) {
      const double d = lane_bcast(reg[jj], jj);
      if (d == 0.0) return false;
      const double v  = reg[jj] / d;
      const double mv = v * d;
      if (lane > jj) reg[jj] = v;
      if (lane < R) {  // final column jj: L(r, j) below the diagonal, D on it; multipliers for the trailing update
        pan[jj * R + lane] = reg[jj];
        mul[jj * R + lane] = mv;
      }
#pragma unroll
      for (int rb = jj + 1; rb < W; ++rb) reg[rb] = fma(-v, lane_bcast(mv, rb), reg[rb]);
    }
  }
  return true;
}
__device__ __forceinline__ bool panel_eliminate(double (&reg)[kPanelCols], const int wd, const int R, double *pan, double *mul,
                                                const int lane)
{
  // (dispatching on narrower width classes for narrow supernodes was tried: the register allocator then spills)
  return panel_eliminate_w<kPanelCols>(reg, wd, R, pan, mul, lane);
}

// Schedule-ordered copies of the factor for the two sweeps (padding slots carry 0).  A streaming pass: writing
// the final values straight into the copies from the panels (scattered 8-byte writes) is quicker for a lone
// wave but costs the batch more HBM traffic than this gather + coalesced write -- measured.
// (branch-free, DEPTH gathers in flight per lane: padding slots read the always-zero accumulator; the copies'
//  lengths are multiples of 8 * 128 >= DEPTH * 64)
#ifndef SFB_COPY_DEPTH
#define SFB_COPY_DEPTH 16
#endif
#ifndef SFB_COPY_UB
#define SFB_COPY_UB 8
#endif
template<int DEPTH>
__device__ __forceinline__ void ldl_sweep_copies(const SparsePlanDev &pl, const Ws &w, const int lane)
{
  const int k = uni(pl.k), nnzL = uni(pl.nnzL);
  const double *ACC = w.Lx;
  auto gather_copy = [&](const int32_t *__restrict__ map, double *__restrict__ dst, const int total) {
    int srcn[DEPTH];  // (the map of block b + 1 is requested before the values of block b are gathered)
#pragma unroll
    for (int dd = 0; dd < DEPTH; ++dd) srcn[dd] = map[lane + dd * kWave];
    for (int q0 = lane; q0 < total; q0 += kWave * DEPTH) {
      int src[DEPTH];
      double v[DEPTH];
#pragma unroll
      for (int dd = 0; dd < DEPTH; ++dd) src[dd] = srcn[dd];
#pragma unroll
      for (int dd = 0; dd < DEPTH; ++dd) v[dd] = ACC[src[dd] >= 0 ? src[dd] : nnzL + k + 1];
      if (q0 + kWave * DEPTH < total) {
#pragma unroll
        for (int dd = 0; dd < DEPTH; ++dd) srcn[dd] = map[q0 + kWave * DEPTH + dd * kWave];
      }
#pragma unroll
      for (int dd = 0; dd < DEPTH; ++dd) __builtin_nontemporal_store(v[dd], &dst[q0 + dd * kWave]);  // read again only by the sweeps
    }
  };
  static_assert(kSweepPadDev * 2 >= DEPTH, "copy loop assumes whole blocks");
  wave_sync();
  gather_copy(pl.fmap, w.LxF, (uni(pl.funits) + kSweepPadDev) * 2 * kWave);
  gather_copy(pl.bmap, w.LxB, (uni(pl.bunits) + kSweepPadDev) * 2 * kWave);
  wave_sync();
}

// UNIT ENGINE of the numeric factorisation (sparse_plan.h; round 5).  Same accumulators, same arithmetic per accumulator
// and in the same order as the supernodal engine below (and the oracle's left-looking loop) -- executed like the sweeps: a
// static schedule of units of 128 independent slots on operands in LDS,
//   F  slot:  t[tgt] = fma(-t[a], t[b] * t[d], t[tgt])      L(a, j), L(b, j), D(j) of the source column j
//   DM slot:  t[tgt] = t[tgt] / t[d]                         the column becomes final
// streamed as 16 bytes per lane and unit from the shared plan (L2).  Per segment of columns: its accumulators (a closed
// segment = a subtree: filled on chip; an open one: fetched from the workspace, where the fill and the earlier segments
// left them) and the accumulators of later columns it updates are brought into LDS, the units run, everything goes back.
// A lone wave needed 1.06 ms for the headline plan's factorisation in the supernodal form (105 panels eliminated by
// v_readlane broadcasts, one IEEE division chain per column, two dependent global round trips per supernode, 150 k VALU
// instructions at 2-3 % lane utilisation); here it is ~1 600 units of ~60-100 ns.
// t = LDS of pl.lds_doubles doubles.  Returns 1 / 0 (zero pivot).
typedef int vint4u __attribute__((ext_vector_type(4)));
template<int DEPTH>
__device__ inline int ldl_numeric_units(const SparsePlanDev &pl, const Item &it, const Ws &w, double *t, const int mode,
                                        const double c, const double sigma, const double delta, const int lane,
                                        unsigned long long *fill_ticks = nullptr)
{
  // fill_ticks (TRACE instance: "Matrix filling" of the reference's summary, qp_solver.hpp:560): wall-clock ticks spent
  // bringing KKT entries and accumulators into place, as opposed to the units, the returns and the sweep copies
  unsigned long long fclk = fill_ticks ? wall_clock64() : 0ull;
  auto fill_lap = [&](const bool is_fill) {
    if (fill_ticks) {
      const unsigned long long now = wall_clock64();
      if (is_fill) *fill_ticks += now - fclk;
      fclk = now;
    }
  };
  const int nnzL = uni(pl.nnzL);
#ifdef SFB_PROF_LDL
  unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, pc = __builtin_amdgcn_s_memtime();
#define SFB_ULAP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); pt[i] += now_ - pc; pc = now_; }
#else
#define SFB_ULAP(i)
#endif
  double *ACC = w.Lx;  // [0, nnzL): L entries, [nnzL, nnzL+k): D, [nnzL+k]: sink, [nnzL+k+1]: always zero
  for (int z = 0, nz = uni(pl.nztop); z < nz; ++z) {  // accumulators of the open segments (the others never live in HBM)
    const int z0 = uni(pl.ztop[2 * z]), zn = uni(pl.ztop[2 * z + 1]);
    for (int p = lane; p < zn; p += kWave) ACC[z0 + p] = 0.0;
  }
  wave_sync();
  kkt_fill<4>(pl.KdescT, pl.KmapT, 0, uni(pl.nnzKT), it, w, ACC, mode, c, sigma, delta, lane);
  wave_sync();
  SFB_ULAP(0)
  fill_lap(true);
  const vint4u *__restrict__ stream = reinterpret_cast<const vint4u *>(pl.ustream);
  auto at = [&](const unsigned off) -> double & { return *reinterpret_cast<double *>(reinterpret_cast<char *>(t) + off); };
  bool zero_pivot = false;
  constexpr int UB = SFB_COPY_UB;  // loads in flight per lane in the copy loops
  for (int sg = 0, nseg = uni(pl.nseg); sg < nseg; ++sg) {
    const int32_t *sgp = pl.seg + 12 * sg;
    const int u0 = uni(sgp[0]), u1 = uni(sgp[1]), c0 = uni(sgp[2]), c1 = uni(sgp[3]), closed = uni(sgp[4]), nLs = uni(sgp[5]);
    const int accN = uni(sgp[6]), kp0 = uni(sgp[7]), kp1 = uni(sgp[8]), gL = uni(sgp[9]), nout = uni(sgp[10]);
    const int32_t *__restrict__ omap = pl.uomap + uni(sgp[11]);  // (padded by a batch: branch-free index loads)
    const int sink = accN + nout;
    // ---- working set -> LDS ----
    if (closed) {
      for (int e = lane; e < accN; e += kWave) t[e] = 0.0;
      wave_lds_fence();
      kkt_fill<4>(pl.Kdesc, pl.KmapL, kp0, kp1, it, w, t, mode, c, sigma, delta, lane);
    } else {
      for (int e0 = lane; e0 < accN; e0 += kWave * UB) {  // own L entries and diagonals: two contiguous ranges of the workspace
        double v[UB];
#pragma unroll
        for (int e = 0; e < UB; ++e) {
          const int x = e0 + e * kWave;
          v[e] = (x < accN) ? ACC[x < nLs ? gL + x : nnzL + c0 + (x - nLs)] : 0.0;
        }
#pragma unroll
        for (int e = 0; e < UB; ++e)
          if (e0 + e * kWave < accN) t[e0 + e * kWave] = v[e];
      }
    }
    for (int e0 = lane; e0 < nout; e0 += kWave * UB) {  // outside accumulators
      int src[UB];
      double v[UB];
#pragma unroll
      for (int e = 0; e < UB; ++e) src[e] = omap[e0 + e * kWave];
#pragma unroll
      for (int e = 0; e < UB; ++e) v[e] = ACC[src[e]];
#pragma unroll
      for (int e = 0; e < UB; ++e)
        if (e0 + e * kWave < nout) t[accN + e0 + e * kWave] = v[e];
    }
    if (lane < 3) t[sink + lane] = (lane == 2) ? 1.0 : 0.0;  // sink, zero, one: operands of the padding slots
    wave_lds_fence();
    SFB_ULAP(1)
    fill_lap(true);
    // ---- units ----
    // The stream is fetched in BLOCKS of DEPTH units, two register sets taking turns: all loads of block b + 1 are requested
    // when block b begins, so that they have a whole block's time (DEPTH dependent LDS round trips) to arrive -- the
    // compiler waits for everything in flight at the loop's head, which then costs nothing.  (Requested one per unit,
    // the same wait drained the queue every DEPTH units: 390 cycles per unit of a lone wave instead of ~200.  Hand-issued
    // loads with counted waits as in the sweeps do not survive the branch between the two kinds of units: the register
    // allocator copies loop-carried registers whose loads are in flight.)  A unit's kind travels in the stream itself
    // (second word of a slot: all ones = division), read from lane 0.
    {
      typedef const __attribute__((address_space(1))) vint4u *gstream_t;
      gstream_t sp = (gstream_t)(stream + (size_t)u0 * kWave + lane);
      auto fetch = [&](vint4u (&q)[DEPTH]) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) q[d] = sp[d * kWave];
        sp += DEPTH * kWave;
      };
      auto run = [&](const vint4u (&q)[DEPTH], const int ub) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          const unsigned x0 = (unsigned)q[d].x, y0 = (unsigned)q[d].y, x1 = (unsigned)q[d].z, y1 = (unsigned)q[d].w;
          const bool dm = __builtin_amdgcn_readfirstlane((int)y0) == -1;
          if (ub + d < u1) {
            if (dm) {
              const double T0 = at(x0 & 0xFFFFu), D0 = at(x0 >> 16), T1 = at(x1 & 0xFFFFu), D1 = at(x1 >> 16);
              at(x0 & 0xFFFFu) = T0 / D0;
              at(x1 & 0xFFFFu) = T1 / D1;
            } else {
              const double T0 = at(x0 & 0xFFFFu), A0 = at(x0 >> 16), B0 = at(y0 & 0xFFFFu), D0 = at(y0 >> 16);
              const double T1 = at(x1 & 0xFFFFu), A1 = at(x1 >> 16), B1 = at(y1 & 0xFFFFu), D1 = at(y1 >> 16);
              at(x0 & 0xFFFFu) = fma(-A0, B0 * D0, T0);
              at(x1 & 0xFFFFu) = fma(-A1, B1 * D1, T1);
            }
          }
        }
      };
      vint4u qa[DEPTH], qb[DEPTH];
      fetch(qa);
      for (int u = u0; u < u1; u += 2 * DEPTH) {  // (the stream is padded by two blocks: the fetches stay inside)
        fetch(qb);
        run(qa, u);
        fetch(qa);
        if (u + DEPTH < u1) run(qb, u + DEPTH);
      }
    }
    wave_lds_fence();
    SFB_ULAP(2)
    // ---- final L and D of the segment, the updated outside accumulators -> workspace; 1 / D (the sweeps multiply by the
    // reciprocal, qp_solver.hpp:458), in S numbering; a zero pivot anywhere = NumericalIssue ----
    for (int e = lane; e < nLs; e += kWave) ACC[gL + e] = t[e];
    for (int j = c0 + lane; j < c1; j += kWave) {
      const double d = t[nLs + (j - c0)];
      zero_pivot = zero_pivot || d == 0.0;
      ACC[nnzL + j]      = d;
      w.Dinv[pl.f2s[j]] = 1.0 / d;
    }
    for (int e0 = lane; e0 < nout; e0 += kWave * UB) {
      int dst[UB];
#pragma unroll
      for (int e = 0; e < UB; ++e) dst[e] = omap[e0 + e * kWave];
#pragma unroll
      for (int e = 0; e < UB; ++e)
        if (e0 + e * kWave < nout) ACC[dst[e]] = t[accN + e0 + e * kWave];
    }
    wave_sync();  // the workspace is up to date before a later segment fetches from it (and t is free again)
    SFB_ULAP(3)
    fill_lap(false);
  }
  if (wave_ballot(zero_pivot)) return 0;
  ldl_sweep_copies<SFB_COPY_DEPTH>(pl, w, lane);  // (gathers in flight per lane: a block of the copy is one memory round trip)
  SFB_ULAP(4)
#ifdef SFB_PROF_LDL
  if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))
    printf("[ldl units block %u mode %d] zero+open fill %llu  segments: working set in %llu  units %llu  out %llu | copies %llu  (cycles of s_memtime)\n",
           blockIdx.x, mode, pt[0], pt[1], pt[2], pt[3], pt[4]);
#endif
  return 1;
}

// Numeric LDL' on the shared pattern, RIGHT-LOOKING over RELAXED SUPERNODES with a static schedule
// (sparse_plan.h), in the factorisation numbering F (a postorder of the elimination tree).  Accumulators
// [L values | D | sink | zero]: every accumulator sees its sources in ascending column order,
// fma(-L(a,j), L(b,j) D(j), acc) -- the arithmetic of the oracle's left-looking loop with its sources in the same
// order, bit for bit (zero entries contribute exact zeros).  The columns come in SEGMENTS:
//  * an LDS SEGMENT is a subtree of the elimination tree whose accumulators fit the item's LDS (the work vector
//    is free during the factorisation).  Its KKT entries are filled into LDS, its supernodes gather their panels
//    from LDS and update each other in LDS: no HBM round trip on the dependent chain.  Final values of L and D
//    leave for the workspace straight from the registers (nobody reads them back before the sweep copies are
//    built), and only the updates to ancestors OUTSIDE the subtree are read-modify-writes in HBM;
//  * a TOP segment (columns whose subtrees do not fit: the separators of an MPC horizon) keeps its accumulators
//    in the item's HBM workspace: per supernode the panel and the first block of trailing accumulators are
//    fetched in one round trip, eliminated, and every trailing accumulator receives the group's updates with ONE
//    read-modify-write.
// Per supernode (R = w + |U| panel rows, U = union of the members' remaining structures): the panel (w x R, explicit
// zeros where a member has no entry) is eliminated in registers when R <= 64 (always inside LDS segments), else in
// LDS.  MPC pattern: 13 on-chip subtrees (the mesh intervals: 938 accumulators, 8 supernodes each) and 13
// supernodes of separators in HBM, 118 supernodes in all.
// t = LDS of pl.lds_doubles doubles.  Returns 1 / 0 (zero pivot).
template<int DEPTH>
__device__ inline int ldl_numeric_dev(const SparsePlanDev &pl, const Item &it, const Ws &w, double *t, const int mode,
                                      const double c, const double sigma, const double delta, const int lane,
                                      unsigned long long *fill_ticks = nullptr)
{
  if (uni(pl.units)) return ldl_numeric_units<DEPTH>(pl, it, w, t, mode, c, sigma, delta, lane, fill_ticks);
  unsigned long long fclk = fill_ticks ? wall_clock64() : 0ull;  // (see ldl_numeric_units)
  auto fill_lap = [&](const bool is_fill) {
    if (fill_ticks) {
      const unsigned long long now = wall_clock64();
      if (is_fill) *fill_ticks += now - fclk;
      fclk = now;
    }
  };
  const int k = uni(pl.k), nnzL = uni(pl.nnzL);
#ifdef SFB_PROF_LDL
  unsigned long long pt[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pc = __builtin_amdgcn_s_memtime();
#define SFB_LAP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); pt[i] += now_ - pc; pc = now_; }
#else
#define SFB_LAP(i)
#endif
  double *ACC   = w.Lx;  // [0, nnzL): L entries, [nnzL, nnzL+k): D, [nnzL+k]: sink, [nnzL+k+1]: always zero
  const int pad = nnzL + k;
  for (int z = 0, nz = uni(pl.nztop); z < nz; ++z) {  // accumulators of the top columns (the others never live in HBM)
    const int z0 = uni(pl.ztop[2 * z]), zn = uni(pl.ztop[2 * z + 1]);
    for (int p = lane; p < zn; p += kWave) ACC[z0 + p] = 0.0;
  }
  wave_sync();
  kkt_fill<4>(pl.KdescT, pl.KmapT, 0, uni(pl.nnzKT), it, w, ACC, mode, c, sigma, delta, lane);
  wave_sync();
  SFB_LAP(0)
  fill_lap(true);
  for (int sg = 0, nseg = uni(pl.nseg); sg < nseg; ++sg) {
    const int32_t *sgp = pl.seg + 12 * sg;
    const int sn0 = uni(sgp[0]), sn1 = uni(sgp[1]), on_chip = uni(sgp[4]);
    fill_lap(false);
    if (on_chip) {
      // ---------------- LDS segment ----------------
      const int accN = uni(sgp[6]), kp0 = uni(sgp[7]), kp1 = uni(sgp[8]), nLs = uni(sgp[5]);
      const int gL = uni(sgp[9]), gD = nnzL + uni(sgp[2]) - nLs;  // LDS offset -> accumulator: + gL (L entries), + gD (diagonal)
      const int zero = accN + 1;                  // LDS: [0, accN) accumulators, accN sink, accN + 1 always zero
      double *scr    = t + ((accN + 2 + 1) & ~1);  // panel scratch behind them
      for (int e = lane; e < accN + 2; e += kWave) t[e] = 0.0;
      wave_lds_fence();
      kkt_fill<4>(pl.Kdesc, pl.KmapL, kp0, kp1, it, w, t, mode, c, sigma, delta, lane);
      wave_lds_fence();
      SFB_LAP(1)
      fill_lap(true);
      // Supernodes of the segment.  Software pipeline over the supernodes: while one is being eliminated, the panel
      // map of the next is on its way from the L2, and the first block of accumulators OUTSIDE the subtree that
      // this supernode updates (they do not depend on its panel) is on its way from HBM.
      constexpr int DL = 4;  // trailing accumulators per lane and block inside an LDS segment
      int j0 = uni(pl.snptr[sn0]), wd = uni(pl.snptr[sn0 + 1]) - j0, R = uni(pl.snR[sn0]);
      int srcL[kPanelCols];
      {
        const int32_t *pmL = pl.pmapL + uni(pl.poff[sn0]);
#pragma unroll
        for (int jj = 0; jj < kPanelCols; ++jj) srcL[jj] = (jj < wd && lane < R) ? pmL[jj * R + lane] : zero;
      }
      for (int sn = sn0; sn < sn1; ++sn) {
        const int s0 = uni(pl.rptr[sn]), sp = uni(pl.rsplit[sn]), s1 = uni(pl.rptr[sn + 1]);
        double *pan = scr, *mul = scr + wd * R;
        // first block of outside accumulators
        int tpx[DL];
        unsigned abx[DL];
        double accx[DL];
#pragma unroll
        for (int dd = 0; dd < DL; ++dd) {  // the schedule arrays are padded: reading past s1 is safe
          tpx[dd] = pl.rtgt[(sp + dd) * kWave + lane];
          abx[dd] = (unsigned)pl.rab[(sp + dd) * kWave + lane];
          if (sp + dd >= s1) {
            tpx[dd] = pad;
            abx[dd] = 0u;
          }
        }
#pragma unroll
        for (int dd = 0; dd < DL; ++dd) accx[dd] = ACC[tpx[dd]];
        // panel of this supernode from LDS
        double reg[kPanelCols];
#pragma unroll
        for (int jj = 0; jj < kPanelCols; ++jj) reg[jj] = t[srcL[jj]];
        // panel map of the next one
        const int snn = (sn + 1 < sn1) ? sn + 1 : sn;
        const int j0n = uni(pl.snptr[snn]), wdn = uni(pl.snptr[snn + 1]) - j0n, Rn = uni(pl.snR[snn]);
        int srcN[kPanelCols];
        {
          const int32_t *pmN = pl.pmapL + uni(pl.poff[snn]);
#pragma unroll
          for (int jj = 0; jj < kPanelCols; ++jj) srcN[jj] = (jj < wdn && lane < Rn) ? pmN[jj * Rn + lane] : zero;
        }
        SFB_LAP(8)
        if (!panel_eliminate(reg, wd, R, pan, mul, lane)) return 0;
        SFB_LAP(9)
        // final D and L values -> workspace, straight from the registers (entries above the diagonal and explicit
        // zeros map to the sink / zero accumulators and are skipped); D also stays on chip for the 1 / D pass below
#pragma unroll
        for (int jj = 0; jj < kPanelCols; ++jj) {
          if (srcL[jj] < accN) ACC[srcL[jj] + (srcL[jj] < nLs ? gL : gD)] = reg[jj];
          if (jj < wd && lane == jj) t[srcL[jj]] = reg[jj];
        }
        wave_lds_fence();
        SFB_LAP(10)
        // trailing accumulators on chip: pairs (a >= b) of rows of U, local rows w + a, w + b
        for (int s = s0; s < sp; s += DL) {
          int tp[DL];
          unsigned ab[DL];
          double acc[DL];
#pragma unroll
          for (int dd = 0; dd < DL; ++dd) {  // the schedule arrays are padded: reading past sp is safe
            tp[dd] = pl.rtgt[(s + dd) * kWave + lane];
            ab[dd] = (unsigned)pl.rab[(s + dd) * kWave + lane];
            if (s + dd >= sp) {
              tp[dd] = accN;
              ab[dd] = 0u;
            }
          }
#pragma unroll
          for (int dd = 0; dd < DL; ++dd) acc[dd] = t[tp[dd]];
          trailing_block<DL, true>(pan, mul, R, wd, tp, ab, acc, t);
        }
        SFB_LAP(11)
        // ... and the ones of ancestors outside the subtree, in HBM
        trailing_block<DL, true>(pan, mul, R, wd, tpx, abx, accx, ACC);
        for (int s = sp + DL; s < s1; s += DL) {
          int tp[DL];
          unsigned ab[DL];
          double acc[DL];
#pragma unroll
          for (int dd = 0; dd < DL; ++dd) {
            tp[dd] = pl.rtgt[(s + dd) * kWave + lane];
            ab[dd] = (unsigned)pl.rab[(s + dd) * kWave + lane];
            if (s + dd >= s1) {
              tp[dd] = pad;
              ab[dd] = 0u;
            }
          }
#pragma unroll
          for (int dd = 0; dd < DL; ++dd) acc[dd] = ACC[tp[dd]];
          trailing_block<DL, true>(pan, mul, R, wd, tp, ab, acc, ACC);
        }
        wave_lds_fence();
        SFB_LAP(12)
        j0 = j0n; wd = wdn; R = Rn;
#pragma unroll
        for (int jj = 0; jj < kPanelCols; ++jj) srcL[jj] = srcN[jj];
      }
      {  // 1 / D of the segment's columns (the sweeps multiply by the reciprocal, qp_solver.hpp:458), in S numbering
        const int c0 = uni(sgp[2]), c1 = uni(sgp[3]);
        for (int j = c0 + lane; j < c1; j += kWave) w.Dinv[pl.f2s[j]] = 1.0 / t[nLs + (j - c0)];
      }
      SFB_LAP(2)
      wave_sync();  // the HBM updates of this segment are complete before a later segment gathers them
      continue;
    }
    // ---------------- top segment: accumulators in HBM ----------------
    for (int sn = sn0; sn < sn1; ++sn) {
    const int j0 = uni(pl.snptr[sn]), wd = uni(pl.snptr[sn + 1]) - j0;
    const int R  = uni(pl.snR[sn]);  // panel rows: the w columns themselves, then the union U of their structures
    double *pan  = t;                                // pan[jj * R + r], r >= jj: accumulators, then L(r, j0+jj)
    double *mul  = t + wd * R;                       // mul[jj * R + r] = L(r, j0+jj) * D(j0+jj)
    // The first DEPTH steps of trailing accumulators do not depend on this supernode's panel: their loads
    // are issued together with the panel's (one memory round trip instead of two per supernode).
    const int s0 = uni(pl.rsplit[sn]), s1 = uni(pl.rptr[sn + 1]);  // (no on-chip steps here: rsplit == rptr)
    int tp0[DEPTH];
    unsigned ab0[DEPTH];
    double acc0[DEPTH];
#pragma unroll
    for (int dd = 0; dd < DEPTH; ++dd) {  // the schedule arrays are padded: reading past s1 is safe
      tp0[dd] = pl.rtgt[(s0 + dd) * kWave + lane];
      ab0[dd] = (unsigned)pl.rab[(s0 + dd) * kWave + lane];
      if (s0 + dd >= s1) {
        tp0[dd] = pad;
        ab0[dd] = 0u;
      }
    }
#pragma unroll
    for (int dd = 0; dd < DEPTH; ++dd) acc0[dd] = ACC[tp0[dd]];
    const int32_t *pm = pl.pmap + uni(pl.poff[sn]);
    const int npan    = wd * R;
    if (R <= kWave) {
      // register form (the usual case): gather, eliminate, write the final values back with the gather indices
      double reg[kPanelCols];
      int src[kPanelCols];
#pragma unroll
      for (int jj = 0; jj < kPanelCols; ++jj)
        src[jj] = (jj < wd && lane < R) ? pm[jj * R + lane] : pad + 1;  // pad + 1: the always-zero accumulator
#pragma unroll
      for (int jj = 0; jj < kPanelCols; ++jj) reg[jj] = ACC[src[jj]];
      SFB_LAP(3)
      if (!panel_eliminate(reg, wd, R, pan, mul, lane)) return 0;
#pragma unroll
      for (int jj = 0; jj < kPanelCols; ++jj)
        if (src[jj] < pad) ACC[src[jj]] = reg[jj];
      wave_lds_fence();
    } else {
    // 1. panel accumulators -> LDS (gather through the panel map, DEPTH loads in flight per lane)
    for (int q0 = lane; q0 < npan; q0 += kWave * DEPTH) {
      int src[DEPTH];
      double v[DEPTH];
#pragma unroll
      for (int dd = 0; dd < DEPTH; ++dd) src[dd] = pm[q0 + dd * kWave];  // the map is padded
#pragma unroll
      for (int dd = 0; dd < DEPTH; ++dd) v[dd] = ACC[src[dd]];
#pragma unroll
      for (int dd = 0; dd < DEPTH; ++dd)
        if (q0 + dd * kWave < npan) pan[q0 + dd * kWave] = v[dd];
    }
    wave_lds_fence();
    SFB_LAP(3)
    // 2. eliminate the panel (LDS only)
    for (int jj = 0; jj < wd; ++jj) {
      const double d = pan[jj * R + jj];
      if (d == 0.0) return 0;
      for (int r = jj + 1 + lane; r < R; r += kWave) {
        const double v  = pan[jj * R + r] / d;
        pan[jj * R + r] = v;
        mul[jj * R + r] = v * d;
      }
      wave_lds_fence();
      for (int rb = jj + 1; rb < wd; ++rb)  // later panel columns, rows ra >= rb
        for (int ra = rb + lane; ra < R; ra += kWave)
          pan[rb * R + ra] = fma(-pan[jj * R + ra], mul[jj * R + rb], pan[rb * R + ra]);
      wave_lds_fence();
    }
    for (int q = lane; q < npan; q += kWave) {  // final D and L values of the panel -> workspace (fire and forget)
      const int dst = pm[q];
      if (dst < pad) ACC[dst] = pan[q];  // not the sink / zero accumulators
    }
    }
    SFB_LAP(4)
    for (int jj = lane; jj < wd; jj += kWave) w.Dinv[pl.f2s[j0 + jj]] = 1.0 / pan[jj * R + jj];
    // 3. trailing accumulators: pairs (a >= b) of rows of U, local rows w + a, w + b
    // (the DEPTH chains of a block advance together, column by column: 2 x DEPTH LDS reads in flight per lane
    //  instead of one dependent pair; padding slots compute on row 0 and land in the sink accumulator)
    trailing_block<DEPTH, false>(pan, mul, R, wd, tp0, ab0, acc0, ACC);
    for (int s = s0 + DEPTH; s < s1; s += DEPTH) {
      int tp[DEPTH];
      unsigned ab[DEPTH];
      double acc[DEPTH];
#pragma unroll
      for (int dd = 0; dd < DEPTH; ++dd) {
        tp[dd] = pl.rtgt[(s + dd) * kWave + lane];
        ab[dd] = (unsigned)pl.rab[(s + dd) * kWave + lane];
        if (s + dd >= s1) {
          tp[dd] = pad;
          ab[dd] = 0u;
        }
      }
#pragma unroll
      for (int dd = 0; dd < DEPTH; ++dd) acc[dd] = ACC[tp[dd]];
      trailing_block<DEPTH, false>(pan, mul, R, wd, tp, ab, acc, ACC);
    }
    SFB_LAP(5)
    wave_sync();
    SFB_LAP(6)
    }
  }
  ldl_sweep_copies<DEPTH>(pl, w, lane);
  SFB_LAP(7)
#ifdef SFB_PROF_LDL
  if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))
    printf("[ldl block %u mode %d] zero+top fill %llu  lds fill %llu  lds segments: panel-load %llu elim %llu finals+dinv %llu trailing-lds %llu trailing-hbm %llu rest %llu | top: panel-load %llu  panel-elim %llu  trailing %llu  sync %llu  copies %llu  (cycles)\n",
           blockIdx.x, mode, pt[0], pt[1], pt[8], pt[9], pt[10], pt[11], pt[12], pt[2], pt[3], pt[4], pt[5], pt[6], pt[7]);
#endif
  return 1;
}

// Stream loads of the sweeps are hand-issued: hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` at the head
// of a software-pipelined loop, which drains the whole prefetch queue every DEPTH steps and exposes one
// full memory latency per block (measured: 107 ns per step for a lone wave vs a 100-cycle LDS chain).
// With inline-asm loads the compiler tracks nothing and the counted waits below are exact: loads
// return in order, so before consuming the data of one step at most (DEPTH-1) steps' loads may remain
// in flight.
typedef double vdouble2 __attribute__((ext_vector_type(2)));  // native vectors: usable as asm register operands
typedef int vint2 __attribute__((ext_vector_type(2)));

// OFF: byte offset folded into the instruction (13-bit signed immediate), so that the units of one prefetch
// block share their address registers instead of paying a 64-bit add each
// NT (used while the launch is bandwidth-bound): the factor values are streamed once per sweep and not reused before
// thousands of other waves' streams have passed -- they should not push the schedule (index) arrays, which every
// wave re-reads, out of the L2.  When only a few waves are left their factors DO stay cached between iterations:
// plain loads then.
template<int OFF, bool NT = false>
__device__ __forceinline__ void stream_load(vdouble2 &v, const vdouble2 *p)
{
  static_assert(OFF >= 0 && OFF < 4096, "immediate offset of global_load");
  if constexpr (NT) asm volatile("global_load_dwordx4 %0, %1, off offset:%2 nt" : "=v"(v) : "v"(p), "n"(OFF));
  else asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(v) : "v"(p), "n"(OFF));
}
template<int OFF>
__device__ __forceinline__ void stream_load(vint2 &v, const vint2 *p)
{
  static_assert(OFF >= 0 && OFF < 4096, "immediate offset of global_load");
  asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=v"(v) : "v"(p), "n"(OFF));
}
// the same 16-byte load with the lanes outside `mask` switched off (they keep their register content and
// generate no memory request); exec is all ones around the sweeps (wave-uniform control flow)
typedef int sint8 __attribute__((ext_vector_type(8)));
// `shift` = 64 - (number of leading lanes that load): exec = all ones >> shift
template<int OFF>
__device__ __forceinline__ void stream_load_masked(vdouble2 &v, const vdouble2 *p, const int shift)
{
  static_assert(OFF >= 0 && OFF < 4096, "immediate offset of global_load");
  asm volatile("s_lshr_b64 exec, -1, %2\n\tglobal_load_dwordx4 %0, %1, off offset:%3 nt\n\ts_mov_b64 exec, -1"
               : "+v"(v)
               : "v"(p), "s"(shift), "n"(OFF)
               : "scc");  // s_lshr writes SCC
}
// lane-mask shifts of 8 consecutive units (32 bytes) by ONE scalar load; the wait is a separate statement placed
// behind the first unit's LDS round trip
__device__ __forceinline__ void mask_fetch(sint8 &m, const int32_t *tab, const int byte_off)
{
  asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(m) : "s"(tab), "s"(byte_off));
}
__device__ __forceinline__ void mask_wait(sint8 &m) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(m)); }
template<int N>
__device__ __forceinline__ void stream_wait(vdouble2 &a, vint2 &b)
{
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}

// One triangular sweep over a packed schedule (sparse_plan.h).  A UNIT is one step of 128 independent
// slots, two per lane,
//   t[tgt] = fma(-val, t[piv], t[tgt]),   (tgt, piv) = idx,
// fetched with one 16-byte and one 8-byte load; the LDS reads of both slots are issued together (one
// LDS round trip per unit on the dependent chain of a lone wave).  Branch-free; DEPTH units
// (= 2 x DEPTH slots per lane) are in flight ahead of their use so the ~1.7 us HBM latency of a lone
// wave is covered; t has k+1 entries, t[k] is the padding slot; `units` is a multiple of the prefetch block (8).
// CHAINED SWEEPS (cacheable form only): sweeps that follow each other share their prefetch registers so that the stream
// never drains between them.  NEXT: the last block fetches the first DEPTH units of the NEXT stream (idx_next / vals_next)
// instead of padding and leaves them in flight in lx / ix; PRE: no prologue, lx / ix hold this sweep's first DEPTH units
// already (issued by the sweep before).  A lone wave otherwise waits one full memory latency at the head of every sweep.
// (Between chained sweeps nothing else may touch memory: the counted waits assume the stream's loads only.)
// RES > 0 (the LAT forward sweep): the values of the first RES units are RESIDENT in the wave's AccVGPRs a[4 u .. 4 u + 3]
// (lat_resident_load) and are not streamed; see the static prefix in the hand-scheduled branch.
// The AccVGPR file is shared BY NUMBER between the compiler and the resident head of the factor stream (round 6): on gfx950 the
// compiler allocates long-lived values to AccVGPRs whenever the 256 architectural registers are short -- it does not know that the
// hand-written prefix keeps factor values there -- so a[0 .. kAgprFree) are left to it and the resident stream takes
// a[kAgprFree .. 255].  check_sweep_spills.py reads this constant and holds both sides to it.  Every 4 registers given away are
// one unit (1 KB per iteration) less resident: 64 free registers cost the headline 0.45 ms.  0: the kernel as it stands needs none.
constexpr int kAgprFree = 0;
template<int DEPTH, bool BYTEOFF, bool LEAN, bool PRE = false, bool NEXT = false, int RES = 0>
__device__ __forceinline__ void sweep_dev(const int32_t *__restrict__ idx, const int units, const double *vals, double *t,
                                 const int lane, const int32_t *__restrict__ mask32, const int full0, const int full1,
                                 vdouble2 (&lx)[DEPTH], vint2 (&ix)[DEPTH], const int32_t *__restrict__ idx_next = nullptr,
                                 const double *vals_next = nullptr, const int32_t *__restrict__ mask_next = nullptr)
{
  static_assert(!(PRE || NEXT) || (!LEAN && DEPTH == 8), "chained sweeps: cacheable form, prefetch distance 8");
  // lean == false (few waves left on the chip: latency matters, HBM traffic does not): every block issues plain loads
  static_assert(DEPTH <= kSweepPadDev && kSweepPadDev % DEPTH == 0, "schedule padding must cover the prefetch distance");
  static_assert(2 * DEPTH <= 62, "vmcnt is a 6-bit counter");
  static_assert(DEPTH == 8, "one scalar load fetches the lane-mask shifts of a block of 8 units");
  // per-lane stream pointers (VGPRs): one per 4 units of values (1 KB each) and per 8 units of indices (12-bit offsets)
  constexpr int NVP = DEPTH / 4, NIP = DEPTH / 8;
  const vdouble2 *vp[NVP];
  const vint2 *ip[NIP];
#pragma unroll
  for (int e = 0; e < NVP; ++e) vp[e] = reinterpret_cast<const vdouble2 *>(vals) + lane + e * 4 * kWave;
#pragma unroll
  for (int e = 0; e < NIP; ++e) ip[e] = reinterpret_cast<const vint2 *>(idx) + lane + e * 8 * kWave;
  if constexpr (!PRE) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) lx[d] = vdouble2{0.0, 0.0};  // masked lanes keep what the register holds
  }
  // Value loads of partially filled units are masked to the lanes that carry slots (sparse_plan.h): the padding
  // of the schedule then costs no HBM traffic.  Units in [full0, full1) are full: plain loads, no mask fetch.
  // MODE 0: plain cached loads (latency mode); 1: non-temporal loads (full units, bandwidth mode); 2: non-temporal
  // loads masked to the lanes that carry slots (partially filled units, bandwidth mode)
  auto issue = [&]<int D, int MODE>(std::integral_constant<int, D>, std::integral_constant<int, MODE>, const sint8 &mk) {
    if constexpr (MODE == 2) stream_load_masked<(D % 4) * kWave * 16>(lx[D], vp[D / 4], mk[D % 8]);
    else stream_load<(D % 4) * kWave * 16, MODE == 1>(lx[D], vp[D / 4]);
    stream_load<(D % 8) * kWave * 8>(ix[D], ip[D / 8]);
  };
  auto for_units = [&]<int... D>(std::integer_sequence<int, D...>, auto &&fn) { (fn(std::integral_constant<int, D>{}), ...); };
  auto advance = [&](const int by) {
#pragma unroll
    for (int e = 0; e < NVP; ++e) vp[e] += by * kWave;
#pragma unroll
    for (int e = 0; e < NIP; ++e) ip[e] += by * kWave;
  };
  sint8 mk;  // mask shifts of the units the current block issues
  {  // the table pointer as a scalar (it may live in a VGPR lane after register spilling)
    const unsigned long long a = reinterpret_cast<unsigned long long>(mask32);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    mask32 = reinterpret_cast<const int32_t *>(((unsigned long long)hi << 32) | lo);
  }
  if constexpr (LEAN) {
    mask_fetch(mk, mask32, 0);
    mask_wait(mk);
    for_units(std::make_integer_sequence<int, DEPTH>{},
              [&]<int D>(std::integral_constant<int, D> dd) { issue(dd, std::integral_constant<int, 2>{}, mk); });
  } else if constexpr (!PRE && RES > 0) {
    static_assert(RES >= DEPTH, "the prologue's units are all resident");
    for_units(std::make_integer_sequence<int, DEPTH>{},
              [&]<int D>(std::integral_constant<int, D>) { stream_load<(D % 8) * kWave * 8>(ix[D], ip[D / 8]); });
  } else if constexpr (!PRE) {
    for_units(std::make_integer_sequence<int, DEPTH>{},
              [&]<int D>(std::integral_constant<int, D> dd) { issue(dd, std::integral_constant<int, 0>{}, mk); });
  }
  advance(DEPTH);
  // (tgt, piv) of the two slots of a unit as byte offsets (BYTEOFF, plan.idx_scale == 8) or element indices
  struct Addr { unsigned t0, p0, t1, p1; };
  auto extract = [](const vint2 &v) { const unsigned a = (unsigned)v.x, b = (unsigned)v.y; return Addr{a & 0xFFFFu, a >> 16, b & 0xFFFFu, b >> 16}; };
  auto at = [&](unsigned v) -> double & {
    return BYTEOFF ? *reinterpret_cast<double *>(reinterpret_cast<char *>(t) + v) : t[v];
  };
  // HAND-SCHEDULED UNITS (the cacheable / latency form with byte offsets: the loop launch's lone waves).  The compiler
  // sinks the unpacking of the next unit's addresses and the issue of the stream loads onto the dependent chain
  // [reads -> fma -> writes -> reads]; here a unit is ONE asm statement in the order
  //   wait for unit u+1's stream data | unpack its four LDS addresses (in the shadow of unit u's reads) | wait for the reads |
  //   two FMAs | two writes | the four reads of unit u+1 right behind them (the LDS serves a wave in order) | the stream loads
  //   of unit u+8 (in the shadow of those reads).
  // The reads' results travel from one statement to the next in (a0, b0, a1, b1) while in flight, like the stream loads do in
  // lx / ix (check_sweep_spills.py replays both).  Same operations on the same operands: bit-identical.
  constexpr bool kAsmUnits = SFB_SWEEP_ASM != 0 && !LEAN && BYTEOFF && DEPTH == 8;
  if constexpr (kAsmUnits) {
#define SFB_SDWA_HI(d, s) "v_add_u32_sdwa " d ", %[tb], " s " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
#define SFB_SDWA_LO(d, s) "v_add_u32_sdwa " d ", %[tb], " s " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
    const unsigned tb = (unsigned)reinterpret_cast<unsigned long long>(t);  // LDS byte address of the work vector (low half of its flat address)
    if constexpr (RES > 0) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ix[0]) : "n"(DEPTH - 1));  // (one load per resident unit in flight)
    else stream_wait<2 * (DEPTH - 1)>(lx[0], ix[0]);
    unsigned ct0, cp0, ct1, cp1;  // LDS addresses of the current unit's targets / pivots
    double a0, b0, a1, b1;        // what its reads return (in flight between the statements)
    asm volatile(SFB_SDWA_HI("%[p0]", "%[ixx]") SFB_SDWA_LO("%[t0]", "%[ixx]") SFB_SDWA_HI("%[p1]", "%[ixy]") SFB_SDWA_LO("%[t1]", "%[ixy]")
                 "ds_read_b64 %[a0], %[p0]\n\tds_read_b64 %[b0], %[t0]\n\tds_read_b64 %[a1], %[p1]\n\tds_read_b64 %[b1], %[t1]"
                 : [p0] "=&v"(cp0), [t0] "=&v"(ct0), [p1] "=&v"(cp1), [t1] "=&v"(ct1), [a0] "=&v"(a0), [b0] "=&v"(b0), [a1] "=&v"(a1), [b1] "=&v"(b1)
                 : [tb] "v"(tb), [ixx] "v"(ix[0].x), [ixy] "v"(ix[0].y)
                 : "memory");  // (the units read and write t behind the compiler's back: no C-level LDS access moves across the sweep's ends)
    (void)cp0; (void)cp1;
    // The VALUE load of a unit is masked to the lanes that carry slots (round 5; the lean form always did): the second launch
    // holds the whole batch and runs at the fabric's read rate (5.9 of this box's 6.0-6.1 TB/s), 17 % of which was the padding
    // of partly filled units.  The lane count of unit u + 8 comes by a scalar load issued at the head of unit u's statement and
    // is used behind its `s_waitcnt lgkmcnt(0)` (scalar loads share that counter; the LDS reads return in order, so the counted
    // lgkmcnt(2) still covers the first two of them whatever the scalar load does).  Masked lanes keep what their register
    // holds: a factor value of an earlier unit, multiplied into the scratch entry t[k] only.
    const int32_t *mt = mask32 + DEPTH;  // lane-mask shifts of the units the current block requests
    // unit_g<D, W, TK>: the unit in register set D with the values (vx, vy); W = loads that may stay in flight when the NEXT
    // unit's data is needed; TK = what it requests for the unit eight ahead: 0 = masked values + indices, 1 = indices only
    // (that unit's values are resident)
    auto unit_g = [&]<int D, int W, int TK>(std::integral_constant<int, D>, std::integral_constant<int, W>, std::integral_constant<int, TK>,
                                            const double vx, const double vy) {
      constexpr int N = (D + 1) % DEPTH;
      unsigned nt0, np0, nt1, np1;
      if constexpr (TK == 0) {
        int msk;
        asm volatile("s_load_dword %[msk], %[mt], %[mo]\n\t"
                     "s_waitcnt vmcnt(%[w])\n\t"  // unit N's stream data (six younger units stay in flight)
                     SFB_SDWA_HI("%[np0]", "%[ixx]") SFB_SDWA_LO("%[nt0]", "%[ixx]") SFB_SDWA_HI("%[np1]", "%[ixy]") SFB_SDWA_LO("%[nt1]", "%[ixy]")
                     "s_waitcnt lgkmcnt(2)\n\t"
                     "v_fma_f64 %[a0], -%[vx], %[a0], %[b0]\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "v_fma_f64 %[a1], -%[vy], %[a1], %[b1]\n\t"
                     "ds_write_b64 %[ct0], %[a0]\n\t"
                     "ds_write_b64 %[ct1], %[a1]\n\t"
                     "ds_read_b64 %[a0], %[np0]\n\t"
                     "ds_read_b64 %[b0], %[nt0]\n\t"
                     "ds_read_b64 %[a1], %[np1]\n\t"
                     "ds_read_b64 %[b1], %[nt1]\n\t"
                     "s_lshr_b64 exec, -1, %[msk]\n\t"
                     "global_load_dwordx4 %[lxo], %[vp], off offset:%[vo]\n\t"
                     "s_mov_b64 exec, -1\n\t"
                     "global_load_dwordx2 %[ixo], %[ip], off offset:%[io]"
                     : [a0] "+v"(a0), [b0] "+v"(b0), [a1] "+v"(a1), [b1] "+v"(b1), [np0] "=&v"(np0), [nt0] "=&v"(nt0), [np1] "=&v"(np1),
                       [nt1] "=&v"(nt1), [lxo] "+v"(lx[D]), [ixo] "=v"(ix[D]), [msk] "=&s"(msk)
                     : [vx] "v"(vx), [vy] "v"(vy), [ct0] "v"(ct0), [ct1] "v"(ct1), [tb] "v"(tb), [ixx] "v"(ix[N].x), [ixy] "v"(ix[N].y),
                       [vp] "v"(vp[D / 4]), [ip] "v"(ip[D / 8]), [vo] "n"((D % 4) * kWave * 16), [io] "n"((D % 8) * kWave * 8), [mt] "s"(mt),
                       [mo] "n"(D * 4), [w] "n"(W)
                     : "scc");
        (void)msk;
      } else {
        asm volatile("s_waitcnt vmcnt(%[w])\n\t"
                     SFB_SDWA_HI("%[np0]", "%[ixx]") SFB_SDWA_LO("%[nt0]", "%[ixx]") SFB_SDWA_HI("%[np1]", "%[ixy]") SFB_SDWA_LO("%[nt1]", "%[ixy]")
                     "s_waitcnt lgkmcnt(2)\n\t"
                     "v_fma_f64 %[a0], -%[vx], %[a0], %[b0]\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "v_fma_f64 %[a1], -%[vy], %[a1], %[b1]\n\t"
                     "ds_write_b64 %[ct0], %[a0]\n\t"
                     "ds_write_b64 %[ct1], %[a1]\n\t"
                     "ds_read_b64 %[a0], %[np0]\n\t"
                     "ds_read_b64 %[b0], %[nt0]\n\t"
                     "ds_read_b64 %[a1], %[np1]\n\t"
                     "ds_read_b64 %[b1], %[nt1]\n\t"
                     "global_load_dwordx2 %[ixo], %[ip], off offset:%[io]"
                     : [a0] "+v"(a0), [b0] "+v"(b0), [a1] "+v"(a1), [b1] "+v"(b1), [np0] "=&v"(np0), [nt0] "=&v"(nt0), [np1] "=&v"(np1),
                       [nt1] "=&v"(nt1), [ixo] "=v"(ix[D])
                     : [vx] "v"(vx), [vy] "v"(vy), [ct0] "v"(ct0), [ct1] "v"(ct1), [tb] "v"(tb), [ixx] "v"(ix[N].x), [ixy] "v"(ix[N].y),
                       [ip] "v"(ip[D / 8]), [io] "n"((D % 8) * kWave * 8), [w] "n"(W));
      }
      ct0 = nt0;
      ct1 = nt1;
      (void)np0; (void)np1;
    };
    auto unit = [&]<int D>(std::integral_constant<int, D> dd) {
      unit_g(dd, std::integral_constant<int, 12>{}, std::integral_constant<int, 0>{}, lx[D].x, lx[D].y);
    };
    // RESIDENT PREFIX (RES > 0): units 0 .. RES + 7 with literal unit numbers.  Unit U < RES takes its two values per lane from the
    // AccVGPRs a[kAgprFree + 4 U .. + 3]; the loads in flight behind unit U + 1's are one per resident unit and two per streamed one among
    // the units U + 2 .. U + 7; unit U requests for unit U + 8 the indices only while that one is resident.  From unit RES + 8 on
    // the eight units in flight are all streamed: the loop below takes over with its counts unchanged.
    int u_first = 0;
    if constexpr (RES > 0) {
      auto for_prefix = [&]<int... U>(std::integer_sequence<int, U...>, auto &&fn) { (fn(std::integral_constant<int, U>{}), ...); };
      for_prefix(std::make_integer_sequence<int, RES + DEPTH>{}, [&]<int U>(std::integral_constant<int, U>) {
        constexpr int D = U % DEPTH;
        constexpr auto loads = [](int v) { return v < RES ? 1 : 2; };
        constexpr int W = loads(U + 2) + loads(U + 3) + loads(U + 4) + loads(U + 5) + loads(U + 6) + loads(U + 7);
        constexpr int TK = (U + DEPTH < RES) ? 1 : 0;
        double vx, vy;
        if constexpr (U < RES) {
          int l0, h0, l1, h1;
          asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\tv_accvgpr_read_b32 %2, a[%6]\n\tv_accvgpr_read_b32 %3, a[%7]"
                       : "=v"(l0), "=v"(h0), "=v"(l1), "=v"(h1)
                       : "n"(kAgprFree + 4 * U), "n"(kAgprFree + 4 * U + 1), "n"(kAgprFree + 4 * U + 2), "n"(kAgprFree + 4 * U + 3));
          vx = __hiloint2double(h0, l0);
          vy = __hiloint2double(h1, l1);
        } else {
          vx = lx[D].x;
          vy = lx[D].y;
        }
        unit_g(std::integral_constant<int, D>{}, std::integral_constant<int, W>{}, std::integral_constant<int, TK>{}, vx, vy);
        if constexpr (D == DEPTH - 1) {
          advance(DEPTH);
          mt += DEPTH;
        }
      });
      static_assert((RES + DEPTH) % DEPTH == 0, "whole blocks");
      u_first = RES + DEPTH;
    }
    for (int u0 = u_first; u0 + (NEXT ? DEPTH : 0) < units; u0 += DEPTH) {
      for_units(std::make_integer_sequence<int, DEPTH>{}, unit);
      advance(DEPTH);
      mt += DEPTH;
    }
    if constexpr (NEXT) {  // the last block (units >= DEPTH, the caller's condition): its loads are the next stream's first units
#pragma unroll
      for (int e = 0; e < NVP; ++e) vp[e] = reinterpret_cast<const vdouble2 *>(vals_next) + lane + e * 4 * kWave;
#pragma unroll
      for (int e = 0; e < NIP; ++e) ip[e] = reinterpret_cast<const vint2 *>(idx_next) + lane + e * 8 * kWave;
      {  // (the next stream's lane-mask shifts, as a scalar pointer like mask32 above)
        const unsigned long long a = reinterpret_cast<unsigned long long>(mask_next);
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
        mt = reinterpret_cast<const int32_t *>(((unsigned long long)hi << 32) | lo);
      }
      for_units(std::make_integer_sequence<int, DEPTH>{}, unit);
    }
    // the reads issued behind the last unit (a padding unit's scratch entry, or the next stream's first unit before its
    // D^-1 step: never used) are retired before their registers are
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1) : : "memory");
    if constexpr (NEXT) {
      wave_lds_fence();  // (no s_waitcnt vmcnt(0) here: the next sweep's first units stay in flight)
    } else {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) stream_wait<0>(lx[d], ix[d]);
      wave_sync();
    }
    return;
#undef SFB_SDWA_HI
#undef SFB_SDWA_LO
  }
  // Software pipeline over the units: the addresses of unit u+1 are unpacked (and its stream loads awaited) while
  // the LDS reads of unit u are in flight, so that only [reads -> fma -> writes] is left on the dependent chain.
  stream_wait<2 * (DEPTH - 1)>(lx[0], ix[0]);
  Addr cur = extract(ix[0]);
  // one block of DEPTH units: consume unit u0 + D, then issue the loads of unit u0 + D + DEPTH (their mask shifts are
  // fetched at the head of the block and awaited behind the first unit's LDS round trip)
  auto block = [&]<int MODE>(std::integral_constant<int, MODE> msk, const int u0) {
    constexpr bool MASKED = MODE == 2;
    sint8 mk;
    if constexpr (MASKED) mask_fetch(mk, mask32, __builtin_amdgcn_readfirstlane((u0 + DEPTH) * 4));
    for_units(std::make_integer_sequence<int, DEPTH>{}, [&]<int D>(std::integral_constant<int, D> dd) {
      constexpr int N = (D + 1) % DEPTH;  // the unit after this one (N == 0: unit 0 of the next block, issued at D == 0)
      const double a0 = at(cur.p0), b0 = at(cur.t0), a1 = at(cur.p1), b1 = at(cur.t1);
      stream_wait<2 * (DEPTH - 2)>(lx[N], ix[N]);  // six younger units are in flight behind unit N
      const Addr nxt = extract(ix[N]);
      at(cur.t0) = fma(-lx[D].x, a0, b0);
      at(cur.t1) = fma(-lx[D].y, a1, b1);
      if constexpr (D == 0 && MASKED) mask_wait(mk);  // behind the LDS round trip above
      issue(dd, msk, mk);  // unit u0 + D + DEPTH (always inside the padded arrays)
      cur = nxt;
    });
    advance(DEPTH);
  };
  // blocks whose TARGETS [u0 + DEPTH, u0 + 2 DEPTH) lie inside the full run issue unmasked loads
  const int r0 = max(0, min(units, full0 - DEPTH)), r1 = max(r0, min(units, full1 - DEPTH));
  int u0 = 0;
  if constexpr (LEAN) {
    for (; u0 < r0; u0 += DEPTH) block(std::integral_constant<int, 2>{}, u0);
    for (; u0 < r1; u0 += DEPTH) block(std::integral_constant<int, 1>{}, u0);
    for (; u0 < units; u0 += DEPTH) block(std::integral_constant<int, 2>{}, u0);
  } else {
    for (; u0 + (NEXT ? DEPTH : 0) < units; u0 += DEPTH) {
      block(std::integral_constant<int, 0>{}, u0);
    }
    if constexpr (NEXT) {  // the last block (units >= DEPTH, the caller's condition): its loads are the next stream's first units
#pragma unroll
      for (int e = 0; e < NVP; ++e) vp[e] = reinterpret_cast<const vdouble2 *>(vals_next) + lane + e * 4 * kWave;
#pragma unroll
      for (int e = 0; e < NIP; ++e) ip[e] = reinterpret_cast<const vint2 *>(idx_next) + lane + e * 8 * kWave;
      block(std::integral_constant<int, 0>{}, u0);
    }
  }
  if constexpr (NEXT) {
    wave_lds_fence();  // (no s_waitcnt vmcnt(0) here: the next sweep's first units stay in flight)
  } else {
    // the trailing prefetches (padding) are never consumed: retire them before their registers are reused
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) stream_wait<0>(lx[d], ix[d]);
    wave_sync();
  }
}

// t (LDS, permuted order) <- K^-1 t   (qp_solver.hpp:457-459)
// SD: prefetch distance of the cacheable (latency) form of the sweeps (8 units)
template<int SD>
__device__ inline void ldl_solve_dev(const SparsePlanDev &pl, const Ws &w, double *t, const int lane, const bool lean,
                                     const double *dinv_lds = nullptr)
{
  // dinv_lds (LAT form): 1 / D in LDS -- the global copy costs a lone wave one memory round trip per iteration
  const double *const Dinv = dinv_lds ? dinv_lds : w.Dinv;
  const int k = uni(pl.k);
  const bool bo = uni(pl.idx_scale) == 8;
  auto sweep = [&](const int32_t *idx, const int units, const double *vals, const int32_t *mask, const int f0, const int f1) {
    if (lean) {
      vdouble2 lx[SFB_SWEEP_DEPTH];
      vint2 ix[SFB_SWEEP_DEPTH];
      if (bo) sweep_dev<SFB_SWEEP_DEPTH, true, true>(idx, units, vals, t, lane, mask, f0, f1, lx, ix);
      else sweep_dev<SFB_SWEEP_DEPTH, false, true>(idx, units, vals, t, lane, mask, f0, f1, lx, ix);
    } else {
      vdouble2 lx[SD];
      vint2 ix[SD];
      if (bo) sweep_dev<SD, true, false>(idx, units, vals, t, lane, mask, f0, f1, lx, ix);
      else sweep_dev<SD, false, false>(idx, units, vals, t, lane, mask, f0, f1, lx, ix);
    }
  };
  sweep(pl.fidx, uni(pl.funits), w.LxF, pl.fmask, uni(pl.ffull0), uni(pl.ffull1));  // forward (column oriented order)
  for (int j0 = lane; j0 < k; j0 += kWave * 8) {  // D^-1 (:458), loads batched
    double dv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) dv[e] = (j0 + e * kWave < k) ? Dinv[j0 + e * kWave] : 0.0;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (j0 + e * kWave < k) t[j0 + e * kWave] = dv[e] * t[j0 + e * kWave];
  }
  wave_sync();
  sweep(pl.bidx, uni(pl.bunits), w.LxB, pl.bmask, uni(pl.bfull0), uni(pl.bfull1));  // backward (rows pushing, descending)
}

// The same for the LAT form of the loop (1 / D and every loop vector in LDS: between the sweeps, and between the backward
// sweep of one iteration and the forward sweep of the next, nothing touches memory): the sweeps are CHAINED -- the forward
// sweep's last block fetches the head of the backward stream, the backward sweep's last block the head of the forward
// stream for the next iteration when the caller says there is one without a stopping check in between (`next`); `pre`:
// lx / ix hold the forward stream's head from the previous call.  Requires byte offsets and funits, bunits >= 8.
// `resident`: the first kLatResident units of the forward stream sit in the wave's AccVGPRs (lat_resident_load).
// rows of 64 elements per vector of the iterate a LAT wave keeps in its registers (sp_solve_item): the LAT form takes plans with
// n, m <= 64 kLatRegRows (the launcher's condition; the headline plan has n = m = 740)
constexpr int kLatRegRows = 12;
// LDS of a LAT wave, in doubles: [work vector k + 2 | c q (n) | scaled bounds (2 m) | 1 / D (k) | positions of the x and of the
// y / z elements in the work vector as 16-bit byte offsets, PADDED to 64 kLatRegRows entries each with the scratch slot k -- the
// update phases run whole rows of 64 lanes without a predicate -- | class of rho per row (bytes, padded alike)]; between 1 / D and the
// positions: the table {1 / rho of the three classes, rho of the three classes} (a class byte is an offset into it)
__host__ __device__ constexpr size_t lat_lds_doubles(const int n, const int m)
{
  return (size_t)(((n + m + 2) + 1) & ~1) + n + 2 * (size_t)m + (n + m) + 8 + (2 * kLatRegRows * 64) / 4 + (kLatRegRows * 64) / 8 + 4;
}
constexpr int kLatResident = (256 - kAgprFree) / 4;  // kAgprFree == 0: all 256 AccVGPRs = 64 KB of the 232 KB a headline item streams per iteration
__device__ __forceinline__ bool lat_resident_ok(const SparsePlanDev &pl) { return uni(pl.funits) >= kLatResident + 16; }
__device__ __forceinline__ void lat_resident_load(const double *LxF, const int lane)
{
  asm volatile("" ::: "a255");  // (the kernel's AccVGPR count)
  const vdouble2 *p = reinterpret_cast<const vdouble2 *>(LxF) + lane;
  auto for_units = [&]<int... U>(std::integer_sequence<int, U...>, auto &&fn) { (fn(std::integral_constant<int, U>{}), ...); };
  for_units(std::make_integer_sequence<int, kLatResident / 4>{}, [&]<int G>(std::integral_constant<int, G>) {
    const vdouble2 *pg = p + G * 4 * kWave;  // four units (1 KB each) per address register
    asm volatile("global_load_dwordx4 a[%1:%2], %0, off\n\t"
                 "global_load_dwordx4 a[%3:%4], %0, off offset:1024\n\t"
                 "global_load_dwordx4 a[%5:%6], %0, off offset:2048\n\t"
                 "global_load_dwordx4 a[%7:%8], %0, off offset:3072"
                 :
                 : "v"(pg), "n"(kAgprFree + 16 * G), "n"(kAgprFree + 16 * G + 3), "n"(kAgprFree + 16 * G + 4), "n"(kAgprFree + 16 * G + 7),
                   "n"(kAgprFree + 16 * G + 8), "n"(kAgprFree + 16 * G + 11), "n"(kAgprFree + 16 * G + 12), "n"(kAgprFree + 16 * G + 15)
                 : "memory");
  });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void ldl_solve_lat(const SparsePlanDev &pl, const Ws &w, double *t, const int lane, const double *Dinv,
                                              vdouble2 (&lx)[8], vint2 (&ix)[8], const bool pre, const bool next, const bool resident = false)
{
  const int k = uni(pl.k);
  if (resident) sweep_dev<8, true, false, false, true, kLatResident>(pl.fidx, uni(pl.funits), w.LxF, t, lane, pl.fmask, 0, 0, lx, ix, pl.bidx, w.LxB, pl.bmask);
  else if (pre) sweep_dev<8, true, false, true, true>(pl.fidx, uni(pl.funits), w.LxF, t, lane, pl.fmask, 0, 0, lx, ix, pl.bidx, w.LxB, pl.bmask);
  else sweep_dev<8, true, false, false, true>(pl.fidx, uni(pl.funits), w.LxF, t, lane, pl.fmask, 0, 0, lx, ix, pl.bidx, w.LxB, pl.bmask);
  for (int j0 = lane; j0 < k; j0 += kWave * 8) {  // D^-1 (:458) from LDS
    double dv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) dv[e] = (j0 + e * kWave < k) ? Dinv[j0 + e * kWave] : 0.0;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (j0 + e * kWave < k) t[j0 + e * kWave] = dv[e] * t[j0 + e * kWave];
  }
  wave_lds_fence();
  if (next) sweep_dev<8, true, false, true, true>(pl.bidx, uni(pl.bunits), w.LxB, t, lane, pl.bmask, 0, 0, lx, ix, pl.fidx, w.LxF, pl.fmask);
  else sweep_dev<8, true, false, true, false>(pl.bidx, uni(pl.bunits), w.LxB, t, lane, pl.bmask, 0, 0, lx, ix);
}

// the same with the loads of UB rows of 64 elements issued together (one memory round trip per UB rows instead of one per row:
// the short vectors of a stopping check are 12 rows, and a loaded LAT wave waits ~2 us for each trip)
template<int UB>
__device__ __forceinline__ double lane_max_abs_b(const double *v, const int len, const int lane)
{
  double r = 0.0;
  for (int e0 = lane; e0 < len; e0 += kWave * UB) {
    double x[UB];
#pragma unroll
    for (int e = 0; e < UB; ++e) x[e] = (e0 + e * kWave < len) ? v[e0 + e * kWave] : 0.0;
#pragma unroll
    for (int e = 0; e < UB; ++e) r = fmax(r, fabs(x[e]));
  }
  return wave_max(r);
}
__device__ __forceinline__ double lane_max_abs(const double *v, int len, int lane)
{
  double r = 0.0;
  for (int e = lane; e < len; e += kWave) r = fmax(r, fabs(v[e]));
  return wave_max(r);
}

// rows of the sparse products, accumulation order of the oracle (storage order, fma).  The entries of a row
// are fetched in chunks of UB: indices / positions first (independent loads), then the values and vector
// entries they point to (independent again), then the fma chain -- two memory round trips per chunk instead
// of two per ENTRY (a stopping check of a lone wave cost five ADMM iterations before).
constexpr int kRowChunk = 8;
template<class PosF, class IdxF>
__device__ __forceinline__ double sp_row_dot(const int p0, const int p1, const double *vals, PosF pos, IdxF idx,
                                             const double *v)
{
  double s = 0.0;
  for (int q = p0; q < p1; q += kRowChunk) {
    int ps[kRowChunk], ix[kRowChunk];
    double a[kRowChunk], x[kRowChunk];
#pragma unroll
    for (int e = 0; e < kRowChunk; ++e) {
      const bool on = q + e < p1;
      ps[e] = on ? pos(q + e) : 0;
      ix[e] = on ? idx(q + e) : 0;
    }
#pragma unroll
    for (int e = 0; e < kRowChunk; ++e) {
      a[e] = vals[ps[e]];
      x[e] = v[ix[e]];
    }
#pragma unroll
    for (int e = 0; e < kRowChunk; ++e)
      if (q + e < p1) s = fma(a[e], x[e], s);
  }
  return s;
}
// The same for RB rows of a lane at once (rows i0, i0 + 64, ...), CE entries per row and memory round trip: a lone wave
// pays two round trips per CHUNK, and a lane's rows one after the other made a stopping check cost four ADMM iterations.
// Every row's fma chain is the sequential one (storage order).  s[r] = the product of row i0 + r * 64 (0 beyond nrows).
template<int RB, int CE, class PtrF, class PosF, class IdxF>
__device__ __forceinline__ void sp_rows_dot(double (&s)[RB], const int i0, const int nrows, PtrF ptr, const double *vals, PosF pos,
                                            IdxF idx, const double *v)
{
  int p[RB], pe[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const int i = i0 + r * kWave;
    const bool on = i < nrows;
    p[r]  = on ? ptr(i) : 0;
    pe[r] = on ? ptr(i + 1) : 0;
    s[r]  = 0.0;
  }
  for (;;) {
    bool more = false;
#pragma unroll
    for (int r = 0; r < RB; ++r) more = more || p[r] < pe[r];
    if (!more) break;
    int ps[RB][CE], ix[RB][CE];
    double a[RB][CE], x[RB][CE];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int e = 0; e < CE; ++e) {
        const bool on = p[r] + e < pe[r];
        ps[r][e] = on ? pos(p[r] + e) : 0;
        ix[r][e] = on ? idx(p[r] + e) : 0;
      }
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int e = 0; e < CE; ++e) {
        a[r][e] = vals[ps[r][e]];
        x[r][e] = v[ix[r][e]];
      }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
#pragma unroll
      for (int e = 0; e < CE; ++e)
        if (p[r] + e < pe[r]) s[r] = fma(a[r][e], x[r][e], s[r]);
      p[r] += CE;
    }
  }
}
template<int RB, int CE>
__device__ __forceinline__ void sp_rows_A(double (&s)[RB], const SparsePlanDev &pl, const Item &it, int i0, const double *v)
{
  sp_rows_dot<RB, CE>(s, i0, uni(pl.m), [&](int i) { return pl.Ap[i]; }, it.Ax, [](int p) { return p; },
                      [&](int p) { return pl.Aj[p]; }, v);
}
template<int RB, int CE>
__device__ __forceinline__ void sp_rows_P(double (&s)[RB], const SparsePlanDev &pl, const Item &it, int i0, const double *v)
{
  sp_rows_dot<RB, CE>(s, i0, uni(pl.n), [&](int i) { return pl.Prp[i]; }, it.Px, [&](int p) { return pl.Prpos[p]; },
                      [&](int p) { return pl.Prj[p]; }, v);
}
template<int RB, int CE>
__device__ __forceinline__ void sp_rows_At(double (&s)[RB], const SparsePlanDev &pl, const Item &it, int j0, const double *v)
{
  sp_rows_dot<RB, CE>(s, j0, uni(pl.n), [&](int j) { return pl.Acp[j]; }, it.Ax, [&](int p) { return pl.Acpos[p]; },
                      [&](int p) { return pl.Aci[p]; }, v);
}
__device__ __forceinline__ double sp_row_A(const SparsePlanDev &pl, const Item &it, int i, const double *v)
{
  return sp_row_dot(pl.Ap[i], pl.Ap[i + 1], it.Ax, [](int p) { return p; }, [&](int p) { return pl.Aj[p]; }, v);
}
__device__ __forceinline__ double sp_row_At(const SparsePlanDev &pl, const Item &it, int j, const double *v)
{
  return sp_row_dot(pl.Acp[j], pl.Acp[j + 1], it.Ax, [&](int p) { return pl.Acpos[p]; },
                    [&](int p) { return pl.Aci[p]; }, v);
}
__device__ __forceinline__ double sp_row_P(const SparsePlanDev &pl, const Item &it, int i, const double *v)
{
  return sp_row_dot(pl.Prp[i], pl.Prp[i + 1], it.Px, [&](int p) { return pl.Prpos[p]; },
                    [&](int p) { return pl.Prj[p]; }, v);
}

// QPSolver::check_stopping, qp_solver.hpp:574-644 (xus, yus, zus, dxus, dyus in the workspace).
// t: LDS scratch (the work vector is free between the update phase of one iteration and the right-hand side of
// the next).  The two order-dependent scalar sums of the test (:607-621, :633) are fed from LDS: their inputs
// are fetched by all lanes in parallel, chunk by chunk, and only the dependent add / fma chain is sequential.
// score (nullable): how far the item is from its tolerances at this check -- residual / tolerance of the primal test,
// or of the dual test when the primal one passes -- the launcher's predictor of the iterations that are left.
// RBX: rows per lane whose products are formed together (their loads share the memory round trips): 4 is what the
// register budget of three waves per SIMD leaves; the LAT form (one wave per SIMD pair of registers more, the loop's state in
// LDS) takes all rows of a lane at once -- a check cost a lone wave 46 us = two iterations, most of it these round trips.
template<int RBX = 4>
__device__ inline int sp_check_stopping(const SparsePlanDev &pl, const Item &it, const Ws &w,
                                        const DenseKernelParams &kp, double *t, const int lane, float *score = nullptr)
{
  const int n = uni(pl.n), m = uni(pl.m);
  const double inf = INFINITY;
  const int chunk  = (uni(pl.k) + 1) / 2;  // pairs of doubles that fit the work vector
  // x and y of the check as the caller left them in the work vector (t[0, n) and t[n, n + m): LDS gathers instead of global ones,
  // the same values); the later parts use t as scratch
  const double *const xl = t, *const yl = t + n;
  {  // OPTIMALITY
    double a = 0.0, r = 0.0, z = 0.0;
    constexpr int RB = RBX, CE = 2;  // (RB = 4: 8 entries in flight per lane)
    for (int i0 = lane; i0 < m; i0 += kWave * RB) {
      double Ax[RB], zi[RB];
#pragma unroll
      for (int rr = 0; rr < RB; ++rr) zi[rr] = (i0 + rr * kWave < m) ? w.zus[i0 + rr * kWave] : 0.0;
      sp_rows_A<RB, CE>(Ax, pl, it, i0, xl);
#pragma unroll
      for (int rr = 0; rr < RB; ++rr) {  // (rows beyond m contribute |0|: the norms are >= 0 anyway)
        a = fmax(a, fabs(Ax[rr]));
        r = fmax(r, fabs(Ax[rr] - zi[rr]));
        z = fmax(z, fabs(zi[rr]));
      }
    }
    const double Ax_norm = wave_max(a), r_norm = wave_max(r), z_norm = wave_max(z);
    if (score != nullptr && lane == 0) *score = (float)(r_norm / (kp.eps_abs + kp.eps_rel * fmax(Ax_norm, z_norm)));
    if (r_norm <= kp.eps_abs + kp.eps_rel * fmax(Ax_norm, z_norm)) {
      double pn = 0.0, qn = 0.0, an = 0.0, rn = 0.0;
      if constexpr (RBX <= 4) {  // (three waves per SIMD: one row at a time, 8 entries in flight)
      for (int j = lane; j < n; j += kWave) {
        const double Px = sp_row_P(pl, it, j, xl), Aty = sp_row_At(pl, it, j, yl), qj = it.q[j];
        pn = fmax(pn, fabs(Px));
        qn = fmax(qn, fabs(qj));
        an = fmax(an, fabs(Aty));
        rn = fmax(rn, fabs(Px + (qj + Aty)));
      }
      } else
      for (int j0 = lane; j0 < n; j0 += kWave * RB) {  // rows of P x and A'y, RB per lane at a time (each row's chain as before)
        double Px[RB], Aty[RB], qj[RB];
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) qj[rr] = (j0 + rr * kWave < n) ? it.q[j0 + rr * kWave] : 0.0;
        sp_rows_P<RB, CE>(Px, pl, it, j0, xl);
        sp_rows_At<RB, CE>(Aty, pl, it, j0, yl);
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {  // (rows beyond n contribute |0|)
          pn = fmax(pn, fabs(Px[rr]));
          qn = fmax(qn, fabs(qj[rr]));
          an = fmax(an, fabs(Aty[rr]));
          rn = fmax(rn, fabs(Px[rr] + (qj[rr] + Aty[rr])));
        }
      }
      const double dual_scale = fmax(fmax(wave_max(pn), wave_max(qn)), wave_max(an)), rn_norm = wave_max(rn);
      if (rn_norm <= kp.eps_abs + kp.eps_rel * dual_scale) return SFB_QP_OPTIMAL;
      if (score != nullptr && lane == 0) *score = (float)(rn_norm / (kp.eps_abs + kp.eps_rel * dual_scale));
    }
  }
  {  // PRIMAL INFEASIBILITY: max(|A'dy|, certificate sum) < thr.  The cheap certificate sum is formed first and
     // A'dy only when the sum leaves the verdict open (same result, NaN included).
    constexpr int UBV     = RBX > 4 ? 6 : 1;  // rows of the short vectors fetched together (round 6; the standard form has no register to spare)
    const double Edy_norm = lane_max_abs_b<UBV>(w.dyus, m, lane);
    const double thr      = kp.eps_pinf * Edy_norm;
    // FAST PATH.  What the verdict needs of the ordered sum is the side of thr it lies on, and its terms summed in ANY order
    // (per lane, then across the wave) differ from the ordered sum by at most 2 gamma_N S, S = the sum of the |terms|, N = 2 m
    // (both are within gamma_(N-1) S of the exact sum: Higham, Accuracy and Stability of Numerical Algorithms, eq. 4.4).
    // With err = 4 N eps S, four times that bound:  sum - err >= thr  =>  acc >= thr, the test is over;
    // sum + err < thr  =>  acc < thr, and then max(|A'dy|, acc) < thr  <=>  |A'dy| < thr (either acc is the larger one, which
    // is < thr and the smaller |A'dy| with it, or |A'dy| is).  The ordered sum itself -- 2 m dependent additions fed from LDS,
    // a quarter of a check for a lone wave -- is formed only when the bound does not decide, when a row breaks to +inf, or
    // when a term is not finite.
    int side = 2;  // 0: acc >= thr, 1: acc < thr, 2: undecided
    {
      double ps = 0.0, pa = 0.0;
      bool brk0 = false;
      for (int i0 = lane; i0 < m; i0 += kWave * UBV) {  // (per lane the rows in ascending order, as before; UBV rows per round trip)
        double uv[UBV], lv[UBV], dv[UBV];
#pragma unroll
        for (int e = 0; e < UBV; ++e) {
          const int i   = i0 + e * kWave;
          const bool on = i < m;
          uv[e] = on ? it.u[i] : 0.0;
          lv[e] = on ? it.l[i] : 0.0;
          dv[e] = on ? w.dyus[i] : 0.0;
        }
#pragma unroll
        for (int e = 0; e < UBV; ++e) {
          if (i0 + e * kWave < m) {
            const double ui = uv[e], li = lv[e], dyi = dv[e];
            const double ta = (ui != inf) ? ui * fmax(0.0, dyi) : 0.0, tb = (li != -inf) ? li * fmin(0.0, dyi) : 0.0;
            ps += ta;
            ps += tb;
            pa += fabs(ta);
            pa += fabs(tb);
            brk0 = brk0 || (ui == inf && dyi > thr) || (li == -inf && dyi < -thr);
          }
        }
      }
      const double sum = wave_sum(ps), S = wave_sum(pa);
      const double err = 4.0 * (2.0 * (double)m) * DBL_EPSILON * S;
      if (!wave_ballot(brk0) && S < inf) {  // (S is NaN or +inf as soon as one term is not finite)
        if (sum - err >= thr) side = 0;
        else if (sum + err < thr) side = 1;
      }
    }
    double acc = 0.0;
    bool brk   = false;
    if (side == 2) {
    // Certificate sum, sequential over the rows with an early exit to +inf (:607-621).  Equivalent form: the
    // result is +inf iff SOME row has an unbounded side with dy beyond the threshold; otherwise it is the
    // ordered sum of u_i max(0,dy_i) then l_i min(0,dy_i) over the rows, where a skipped term adds +0.0 (exact:
    // the running sum starts at +0.0 and therefore is never -0.0).
    for (int c0 = 0; c0 < m; c0 += chunk) {
      const int c1 = min(m, c0 + chunk);
      for (int i = c0 + lane; i < c1; i += kWave) {
        const double ui = it.u[i], li = it.l[i], dyi = w.dyus[i];
        t[2 * (i - c0)]     = (ui != inf) ? ui * fmax(0.0, dyi) : 0.0;
        t[2 * (i - c0) + 1] = (li != -inf) ? li * fmin(0.0, dyi) : 0.0;
        brk = brk || (ui == inf && dyi > thr) || (li == -inf && dyi < -thr);
      }
      wave_sync();
      {  // the sequential sum: 16 terms per LDS round trip (one read per add exposed the LDS latency 1 480 times per check:
         // 49 of the 90 us a check cost a lone wave)
        const int cnt = 2 * (c1 - c0);
        int e = 0;
        for (; e + 16 <= cnt; e += 16) {
          vdouble2 v[8];
#pragma unroll
          for (int qq = 0; qq < 8; ++qq) v[qq] = *reinterpret_cast<const vdouble2 *>(t + e + 2 * qq);
#pragma unroll
          for (int qq = 0; qq < 8; ++qq) {
            acc += v[qq].x;
            acc += v[qq].y;
          }
        }
        for (; e < cnt; ++e) acc += t[e];
      }
      wave_sync();
    }
    if (wave_ballot(brk)) acc = inf;
    }
    if (side == 1 || (side == 2 && !(acc >= thr))) {
      for (int e = lane; e < m; e += kWave) t[e] = w.dyus[e];  // dy into the work vector: the gathers of A'dy stay on chip
      wave_sync();
      double an = 0.0;
      if constexpr (RBX <= 4) {
        for (int j = lane; j < n; j += kWave) an = fmax(an, fabs(sp_row_At(pl, it, j, t)));
      } else
      for (int j0 = lane; j0 < n; j0 += kWave * RBX) {
        double Atdy[RBX];
        sp_rows_At<RBX, 2>(Atdy, pl, it, j0, t);
#pragma unroll
        for (int rr = 0; rr < RBX; ++rr) an = fmax(an, fabs(Atdy[rr]));
      }
      const double Aty_norm = wave_max(an);
      wave_sync();
      const double mxv      = (side == 1 || !(Aty_norm < acc)) ? Aty_norm : acc;  // (side 1: acc < thr is known, see above)
      if (mxv < thr) return SFB_QP_PRIMAL_INFEASIBLE;
    }
  }
  {  // DUAL INFEASIBILITY: |P dx| <= thr, q'dx <= thr and the row conditions on A dx, each evaluated only while
     // the verdict is still open.
    double dmx = 0.0;  // |dx|_inf, and dx into the work vector for the gathers of P dx (LDS instead of global round trips)
    constexpr int UBV = RBX > 4 ? 6 : 1;
    for (int e0 = lane; e0 < n; e0 += kWave * UBV) {
      double v[UBV];
#pragma unroll
      for (int e = 0; e < UBV; ++e) v[e] = (e0 + e * kWave < n) ? w.dxus[e0 + e * kWave] : 0.0;
#pragma unroll
      for (int e = 0; e < UBV; ++e)
        if (e0 + e * kWave < n) {
          t[e0 + e * kWave] = v[e];
          dmx               = fmax(dmx, fabs(v[e]));
        }
    }
    const double dx_norm = wave_max(dmx);
    const double thr     = kp.eps_dinf * dx_norm;
    wave_sync();
    double pn            = 0.0;
    for (int j0 = lane; j0 < n; j0 += kWave * 8) {
      double Pdx[8];
      sp_rows_P<8, 1>(Pdx, pl, it, j0, t);
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) pn = fmax(pn, fabs(Pdx[rr]));
    }
    const double Pdx_n = wave_max(pn);
    if (!(Pdx_n <= thr)) return -1;
    wave_sync();  // (every lane is done with dx in t: the chain below stages its operands there)
    double qdx = 0.0;  // q' dx, sequential fma chain (:633) fed from LDS
    for (int c0 = 0; c0 < n; c0 += chunk) {
      const int c1 = min(n, c0 + chunk);
      for (int j = c0 + lane; j < c1; j += kWave) {
        t[2 * (j - c0)]     = it.q[j];
        t[2 * (j - c0) + 1] = w.dxus[j];
      }
      wave_sync();
      {
        const int cnt = c1 - c0;
        int e = 0;
        for (; e + 8 <= cnt; e += 8) {
          vdouble2 v[8];
#pragma unroll
          for (int qq = 0; qq < 8; ++qq) v[qq] = *reinterpret_cast<const vdouble2 *>(t + 2 * (e + qq));
#pragma unroll
          for (int qq = 0; qq < 8; ++qq) qdx = fma(v[qq].x, v[qq].y, qdx);
        }
        for (; e < cnt; ++e) qdx = fma(t[2 * e], t[2 * e + 1], qdx);
      }
      wave_sync();
    }
    if (!(qdx <= thr)) return -1;
    bool rowok = true;
    for (int i = lane; i < m; i += kWave) {
      const double Adx = sp_row_A(pl, it, i, w.dxus), ui = it.u[i], li = it.l[i];
      if (ui == inf) rowok = rowok && (Adx >= -thr);
      else if (li == -inf) rowok = rowok && (Adx <= thr);
      else rowok = rowok && (fabs(Adx) < thr);
    }
    if (!wave_ballot(!rowok)) return SFB_QP_DUAL_INFEASIBLE;
  }
  return -1;
}

// One row of the reference's verbose table (qp_solver.hpp:490-501) after a stopping check: ITER, OBJ = (0.5 P x + q) . x,
// PRI_RES = |A x - z|_inf, DUA_RES = |P x + q + A' y|_inf on the unscaled iterate, TIME in microseconds of the device clock
// since the item's solve began -- the three columns in the reference's expressions and the order of
// oracle/qp_sparse_oracle.c's trace (products row by row like the check's, the dot product a sequential mul + add chain).
// Only the TRACE instance of the kernel calls this (sfb_sparse_qp_solve_batch_trace); t is scratch like in the check.
__device__ __noinline__ void sp_trace_row(const SparsePlanDev &pl, const Item &it, const Ws &w, double *t, const int lane,
                                          const uint32_t iter, const unsigned long long t0_ticks, double *row)
{
  const int n = uni(pl.n), m = uni(pl.m);
  const int chunk = (uni(pl.k) + 1) / 2;
  double pri = 0.0, dua = 0.0, o = 0.0;
  for (int i = lane; i < m; i += kWave) pri = fmax(pri, fabs(sp_row_A(pl, it, i, w.xus) - w.zus[i]));
  for (int c0 = 0; c0 < n; c0 += chunk) {
    const int c1 = min(n, c0 + chunk);
    for (int j = c0 + lane; j < c1; j += kWave) {
      const double Px = sp_row_P(pl, it, j, w.xus), qj = it.q[j];
      dua = fmax(dua, fabs(Px + qj + sp_row_At(pl, it, j, w.yus)));
      t[2 * (j - c0)]     = 0.5 * Px + qj;
      t[2 * (j - c0) + 1] = w.xus[j];
    }
    wave_sync();
    for (int e = 0; e < c1 - c0; ++e) o += t[2 * e] * t[2 * e + 1];
    wave_sync();
  }
  pri = wave_max(pri);
  dua = wave_max(dua);
  if (lane == 0) {
    row[0] = (double)iter;
    row[1] = o;
    row[2] = pri;
    row[3] = dua;
    row[4] = (double)(wall_clock64() - t0_ticks) * 0.01;
  }
}

// detail::polish_qp (sparse), embedded in the full pattern.  In/out: scaled xs / ys in the workspace.
// wf: the workspace view whose factor fields the polish factorisation may overwrite (w itself, or polish_ws(w))
// acc[r] <- the sequential chain fm(r, position, values, acc[r]) over the entries [ptr(i).a, ptr(i).b) of the rows i = i0 + 64 r,
// RB rows of a lane together, CE entries per row and step: idx(q) = the positions / indices of entry q (shared arrays), load(.) =
// what they point to (item data, vectors).  The idx loads of step s + 1 are issued before the products of step s are formed:
// one memory round trip per step on the dependent path.  Rows beyond nrows and entries beyond a row's end are not applied.
template<int RB, int CE, class PtrF, class IdxF, class LoadF, class FmaF>
__device__ __forceinline__ void sp_rows_chain(double (&acc)[RB], const int i0, const int nrows, PtrF ptr, IdxF idx, LoadF load, FmaF fm)
{
  using IdxT = decltype(idx(0));
  using ValT = decltype(load(idx(0)));
  int p[RB], pe[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const int i = i0 + r * kWave;
    p[r] = pe[r] = 0;
    if (i < nrows) {
      const auto pp = ptr(i);
      p[r]  = pp.a;
      pe[r] = pp.b;
    }
  }
  IdxT ix[RB][CE];
  auto fetch = [&] {
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int e = 0; e < CE; ++e) ix[r][e] = idx((p[r] + e < pe[r]) ? p[r] + e : 0);
  };
  fetch();
  for (;;) {
    bool more = false;
#pragma unroll
    for (int r = 0; r < RB; ++r) more = more || p[r] < pe[r];
    if (!more) break;
    ValT v[RB][CE];
    IdxT ic[RB][CE];
    bool on[RB][CE];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int e = 0; e < CE; ++e) {
        ic[r][e] = ix[r][e];
        v[r][e]  = load(ic[r][e]);
        on[r][e] = p[r] + e < pe[r];
      }
#pragma unroll
    for (int r = 0; r < RB; ++r) p[r] += CE;
    fetch();
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int e = 0; e < CE; ++e)
        if (on[r][e]) acc[r] = fm(r, ic[r][e], v[r][e], acc[r]);
  }
}

template<int SD, bool WIDE = false>
__device__ inline void sp_polish(const SparsePlanDev &pl, const Item &it, const Ws &w, const Ws &wf, const DenseKernelParams &kp,
                                 double *t, const double c, const int lane, const bool lean)  // (its fill + factorisation count as "Polish", qp_solver.hpp:563)
{
  const int n = uni(pl.n), m = uni(pl.m), k = uni(pl.k);
  const double inf = INFINITY, eps = DBL_EPSILON;
  for (int i = lane; i < m; i += kWave) {  // :113-123
    const double yi = w.ys[i];
    double a        = 0.0;
    if (yi < -100 * eps && it.l[i] != -inf) a = 1.0;
    if (yi > 100 * eps && it.u[i] != inf) a = 2.0;
    w.act[i] = a;
  }
  for (int e = lane; e < k; e += kWave) w.tv[e] = 0.0;
  wave_sync();
  if (!ldl_numeric_dev<8>(pl, it, wf, t, 1, c, kp.sigma, kp.delta, lane)) return;  // :187-190
  for (uint32_t iter = 0; iter != kp.polish_iter; ++iter) {                      // :193-195
    // residual rows h - K tv (:193-195), entries in storage order as in the oracle.  Like sp_row_dot the entries of
    // a row are fetched in chunks: positions / indices first, then everything they point to, then the fma chain
    // (two memory round trips per chunk of 8 entries instead of three per entry).
    // RB rows of a lane advance together, CE entries each per step (sp_rows_chain: the positions of the next step are on
    // their way while the products of this one are formed); every row's chain is the sequential one of the oracle.
#ifndef SFB_POLISH_RB
#define SFB_POLISH_RB 2
#define SFB_POLISH_CE 2
#endif
#ifndef SFB_POLISH_WIDE_RB
#define SFB_POLISH_WIDE_RB 2
#define SFB_POLISH_WIDE_CE 4
#endif
    // (WIDE: the instances with a whole SIMD's registers to themselves -- polishers, LAT waves helping them -- fetch more per step)
    constexpr int RB = WIDE ? SFB_POLISH_WIDE_RB : SFB_POLISH_RB, CE = WIDE ? SFB_POLISH_WIDE_CE : SFB_POLISH_CE;
    struct Ix { int a, b; };
    struct V3 { double a, b, c; };
    struct V4 { double a, b, c, d; };
    // (first round: tv = 0, every chain below is a sum of exact zeros -- the data are finite: polish follows a solve that ended
    //  Optimal, i.e. with finite residuals of P x, A x and A' y -- so the rows are taken as empty and the residual is h, bit for
    //  bit what the chains would give)
    const bool zero_tv = iter == 0;
    for (int i0 = lane; i0 < n; i0 += kWave * RB) {
      double acc[RB], sxi[RB], qi[RB];
      int pv[RB];
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int i = i0 + r * kWave;
        const bool on = i < n;
        acc[r] = 0.0;
        sxi[r] = on ? w.sx[i] : 0.0;
        qi[r]  = on ? it.q[i] : 0.0;
        pv[r]  = on ? pl.pinv[i] : k;
      }
      // P as selfadjointView<Upper>: entry (min, max)
      sp_rows_chain<RB, CE>(acc, i0, n, [&](int i) { return zero_tv ? Ix{0, 0} : Ix{pl.Sp[i], pl.Sp[i + 1]}; },
                            [&](int q) { return Ix{pl.Spos[q], pl.Sj[q]}; },
                            [&](const Ix &x) { return V3{it.Px[x.a], w.sx[x.b], w.tv[x.b]}; },
                            [&](int r, const Ix &x, const V3 &v, double a) {
                              const int i = i0 + r * kWave;
                              const double s_ec = (x.b > i) ? v.b : sxi[r], s_er = (x.b > i) ? sxi[r] : v.b;  // row er <= column ec
                              return fma(c * s_ec * s_er * v.a, v.c, a);
                            });
      // column i of A, active rows only
      sp_rows_chain<RB, CE>(acc, i0, n, [&](int i) { return zero_tv ? Ix{0, 0} : Ix{pl.Acp[i], pl.Acp[i + 1]}; },
                            [&](int q) { return Ix{pl.Acpos[q], pl.Aci[q]}; },
                            [&](const Ix &x) { return V4{w.act[x.b], w.sy[x.b], it.Ax[x.a], w.tv[n + x.b]}; },
                            [&](int r, const Ix &, const V4 &v, double a) { return (v.a != 0.0) ? fma(v.b * sxi[r] * v.c, v.d, a) : a; });
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const double h = -c * (sxi[r] * qi[r]);  // :180
        if (i0 + r * kWave < n) t[pv[r]] = h - acc[r];
      }
    }
    for (int r0 = lane; r0 < m; r0 += kWave * RB) {
      double acc[RB], a_[RB], syr[RB], lb[RB], ub[RB];
      int pv[RB];
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int rr = r0 + r * kWave;
        const bool on = rr < m;
        acc[r] = 0.0;
        a_[r]  = on ? w.act[rr] : 0.0;
        syr[r] = on ? w.sy[rr] : 0.0;
        lb[r]  = on ? it.l[rr] : 0.0;
        ub[r]  = on ? it.u[rr] : 0.0;
        pv[r]  = on ? pl.pinv[n + rr] : k;
      }
      sp_rows_chain<RB, CE>(acc, r0, m,
                            [&](int rr) { return (!zero_tv && w.act[rr] != 0.0) ? Ix{pl.Ap[rr], pl.Ap[rr + 1]} : Ix{0, 0}; },  // inactive rows: h - acc = 0
                            [&](int q) { return Ix{q, pl.Aj[q]}; },
                            [&](const Ix &x) { return V3{it.Ax[x.a], w.sx[x.b], w.tv[x.b]}; },
                            [&](int r, const Ix &, const V3 &v, double a) { return fma(syr[r] * v.b * v.a, v.c, a); });
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        double h = 0.0;
        if (a_[r] != 0.0) h = (a_[r] == 1.0) ? syr[r] * lb[r] : syr[r] * ub[r];  // :181-182
        if (r0 + r * kWave < m) t[pv[r]] = h - acc[r];
      }
    }
    wave_sync();
    ldl_solve_dev<SD>(pl, wf, t, lane, lean);
    for (int e = lane; e < k; e += kWave) w.tv[e] += t[pl.pinv[e]];
    wave_sync();
  }
  for (int j = lane; j < n; j += kWave) w.xs[j] = w.tv[j];  // :199
  for (int i = lane; i < m; i += kWave)
    if (w.act[i] != 0.0) w.ys[i] = w.tv[n + i];             // :200-201
  wave_sync();
}

// Queue of a time-sliced launch (device memory, zeroed by the launcher; counters on their own cache lines):
//   q[kQFresh]  fresh items handed out so far (ticket counter, items 0 .. batch-1 in launch order)
//   q[kQHead], q[kQTail]  ring of SUSPENDED items: pushes reserve q[kQTail]++, pops claim q[kQHead] by CAS
//   q[kQRing + i]  ring entries (item + 1, 0 = empty), capacity = batch
constexpr int kQFresh = 0, kQHead = 16, kQTail = 32, kQRing = 48;

enum { SP_DONE = 0, SP_SUSPENDED = 2, SP_PAUSED = 3 };

// GUARD of a pruned plan: the entries of A the plan's creator declared zero must be zero in this item (NaN counts
// as non-zero).  Amasked is padded: branch-free batches.
__device__ __forceinline__ bool sp_guard_ok(const SparsePlanDev &pl, const double *__restrict__ Ax, const int lane)
{
  constexpr int UB = 8;
  const int nmasked = uni(pl.nmasked);
  bool bad = false;
  int srcn[UB];  // (the positions of batch b + 1 are requested before the entries of batch b; Amasked is padded by a batch)
#pragma unroll
  for (int e = 0; e < UB; ++e) srcn[e] = pl.Amasked[lane + e * kWave];
  for (int p0 = lane; p0 < nmasked; p0 += kWave * UB) {
    int src[UB];
    double v[UB];
#pragma unroll
    for (int e = 0; e < UB; ++e) src[e] = srcn[e];
#pragma unroll
    for (int e = 0; e < UB; ++e) v[e] = Ax[src[e]];
    if (p0 + kWave * UB < nmasked) {
#pragma unroll
      for (int e = 0; e < UB; ++e) srcn[e] = pl.Amasked[p0 + kWave * UB + e * kWave];
    }
#pragma unroll
    for (int e = 0; e < UB; ++e) bad = bad || !(v[e] == 0.0);
  }
  return wave_ballot(bad) == 0ull;
}

// One item.  `slot` = workspace slot.  resume == false: from the start (scaling, factorisation, initial iterate);
// resume == true: continue an item another block has suspended (its state is in its workspace).
// Runs until the item is finished (SP_DONE), or -- time-sliced launches only, queue != nullptr -- until the item
// has used its slice while others are waiting for a wave (SP_SUSPENDED: state saved, the caller queues the item).
// LAT: the form for launches with few waves, each nearly alone on its SIMD (second launch of the predicted order): one
// wave per SIMD pair of registers more (256 VGPRs) and the ADMM vectors of the loop in LDS, see the loop
template<bool LAT, bool TRACE = false, bool WIDE = false>
__device__ __forceinline__ int sp_solve_item(const SparsePlanDev &pl, const DenseKernelParams &kp, const double *__restrict__ gPx,
                                             const double *__restrict__ gq, const double *__restrict__ gAx,
                                             const double *__restrict__ gl, const double *__restrict__ gu,
                                             const double *__restrict__ gwx, const double *__restrict__ gwy,
                                             double *__restrict__ gx, double *__restrict__ gy, double *__restrict__ gobj,
                                             uint32_t *__restrict__ giter, int32_t *__restrict__ gcode,
                                             double *__restrict__ gws, const size_t ws_doubles, const int lean_waves,
                                             bool lean, const size_t b, const size_t slot, double *t, const int lane,
                                             bool resume, const int32_t *queue, const int batch,
                                             const uint32_t slice, const bool allow_reuse, const int phases, float *score,
                                             double *trace = nullptr, const int trace_cap = 0, double *phase_us = nullptr)
{
  // phase_us (TRACE instance, nullable): six doubles per item, microseconds of the device's wall clock -- scaling and
  // pre-check (before the reference's t0, qp_solver.hpp:376), then its summary (:559-563) "Matrix filling", "Factorization",
  // "Iteration", "Polish", and un-scale / report.  Their sum is the time the item held its wave.
  [[maybe_unused]] unsigned long long ph_clk = 0, ph_fill = 0;
  [[maybe_unused]] unsigned long long ph_t[6] = {0, 0, 0, 0, 0, 0};
  [[maybe_unused]] auto phase_lap = [&](const int i) {
    if constexpr (TRACE) {
      const unsigned long long now = wall_clock64();
      ph_t[i] += now - ph_clk;
      ph_clk = now;
    }
  };
  if constexpr (TRACE) ph_clk = wall_clock64();
  const int ph0 = phases & 15, ph1 = (phases >> 4) & 15;  // the phases this launch performs
  const uint32_t pause_at = ((uint32_t)phases >> 8) & 0xFFFFFu;  // != 0: leave the ADMM loop open at the first check from here on
  const int n = uni(pl.n), m = uni(pl.m), k = uni(pl.k);
  const int nnzP = uni(pl.nnzP), nnzA = uni(pl.nnzA);  // (nnzA: what the kernel works on, the kept entries of a pruned plan)
  Item it{gPx + b * (size_t)nnzP, gq + b * (size_t)n, gAx + b * (size_t)uni(pl.nnzA_io), gl + b * (size_t)m,
          gu + b * (size_t)m};
  const Ws w = carve_ws(gws + slot * ws_doubles, n, m, uni(pl.nnzL), uni(pl.funits), uni(pl.bunits));
#ifdef SFB_SP_TIMELINE  // profiling build (scripts/build_prof.sh): wall-clock stamps (100 MHz) of the item's phases
  unsigned long long tl0 = wall_clock64(), tl1 = 0, tl2 = 0;
#endif
  double c = 1.0;
  int ret_code = -1;
  uint32_t iter = 0;
  const uint32_t sci   = kp.stop_check_iter;
  const uint32_t maxit = kp.max_iter;
  uint32_t next_chk    = (sci >= 2) ? 1u : 0xFFFFFFFFu;
  [[maybe_unused]] int trace_rows = 0;  // TRACE: rows of the verbose table written so far (one block per item: never suspended)
  if (lane == 0) t[k] = 0.0;  // padding slot of the packed sweeps
  const double inf = INFINITY;
  unsigned long long t0_ticks = wall_clock64();  // start of the item's solve (:376), kept across suspensions
  if (ph0 >= PH_ADMM) {  // phased launch: every item continues from the state the previous phase left in its workspace
    if (w.hdr[kHdrComplete] == 1.0) return SP_DONE;
#ifdef SFB_SP_TIMELINE
    const bool tl_continued = !resume;  // first visit of this launch: tl1 = when the launch took the item up
#endif
    if (!resume) ret_code = (int)w.hdr[kHdrCode];  // (an item suspended inside this phase is open by construction)
    resume = true;
#ifdef SFB_SP_TIMELINE
    if (tl_continued && lane == 0) w.hdr[5] = (double)wall_clock64();
    wave_sync();
#endif
  }
  if (resume) {
    c        = w.hdr[0];
    iter     = (uint32_t)w.hdr[1];
    next_chk = (uint32_t)w.hdr[2];
    t0_ticks = (unsigned long long)w.hdr[3];
#ifdef SFB_SP_TIMELINE
    tl0 = (unsigned long long)w.hdr[4];
    tl1 = (unsigned long long)w.hdr[5];
#endif
    if (pl.Aorig != nullptr) it.Ax = w.Axc;
  } else {
  // reuse_factor (sfb.h): the caller vouches that P and A of this item are what the previous call on this workspace
  // solved.  What the workspace holds from that call -- the compacted A, the scaling (it depends on P, A and, through
  // c, on |q|), the factor (on the scaling, sigma and rho, i.e. on which rows are equalities) -- is re-used as far as
  // it is PROVABLY what this call would compute: c and rho are recomputed and compared bit for bit.
  const bool keep = allow_reuse && kp.reuse != 0 && (unsigned long long)__double_as_longlong(w.hdr[kHdrStamp]) == kFactorStamp &&
                    w.hdr[kHdrSigma] == kp.sigma && w.hdr[kHdrScaling] == (double)kp.scaling;
  if (pl.Aorig != nullptr && keep) it.Ax = w.Axc;
  else if (pl.Aorig != nullptr) {
    // Pruned plan (its guard has passed, see the kernel): the kept entries of A are compacted into the workspace;
    // everything below works on the compressed pattern.  (Aorig is padded: branch-free batches.)
    constexpr int UB = 8;
    int srcn[UB];  // (the positions of batch b + 1 are requested before the entries of batch b are gathered)
#pragma unroll
    for (int e = 0; e < UB; ++e) srcn[e] = pl.Aorig[lane + e * kWave];
    for (int p0 = lane; p0 < nnzA; p0 += kWave * UB) {
      int src[UB];
      double v[UB];
#pragma unroll
      for (int e = 0; e < UB; ++e) src[e] = srcn[e];
#pragma unroll
      for (int e = 0; e < UB; ++e) v[e] = it.Ax[src[e]];
      if (p0 + kWave * UB < nnzA) {
#pragma unroll
        for (int e = 0; e < UB; ++e) srcn[e] = pl.Aorig[p0 + kWave * UB + e * kWave];
      }
#pragma unroll
      for (int e = 0; e < UB; ++e)
        if (p0 + e * kWave < nnzA) w.Axc[p0 + e * kWave] = v[e];
    }
    it.Ax = w.Axc;
    wave_sync();
  }

  bool keep_scaling = keep && !kp.scaling;  // (no scaling: sx = sy = 1, c = 1 are there already)
  // ---- scale :673-730 ----
  if (kp.scaling) {
    // :681-693: column inf-norms of P as stored, then c
    for (int j = lane; j < n; j += kWave) {
      double v = 0.0;
      for (int p = pl.Pp[j]; p < pl.Pp[j + 1]; ++p) v = fmax(v, fabs(it.Px[p]));
      if (v == 0.0) v = 1.0;
      t[j] = v;
    }
    wave_sync();
    double sum = t[0];
    {  // sequential sum (:687), 16 terms per LDS round trip
      int j = 1;
      for (; j + 16 <= n; j += 16) {
        double v[16];
#pragma unroll
        for (int qq = 0; qq < 16; ++qq) v[qq] = t[j + qq];
#pragma unroll
        for (int qq = 0; qq < 16; ++qq) sum += v[qq];
      }
      for (; j < n; ++j) sum += t[j];
    }
    const double qn = lane_max_abs(it.q, n, lane);
    c               = 1.0 / fmax(fmax(1e-6, sum / (double)n), qn);
    wave_sync();
    keep_scaling = keep && c == w.hdr[kHdrC];  // same P, A and c: the Ruiz iteration below would reproduce sx, sy
  }
  if (!keep_scaling) {
    // ---- analyze(): :306-308 ----
    for (int j = lane; j < n; j += kWave) w.sx[j] = 1.0;
    for (int i = lane; i < m; i += kWave) w.sy[i] = 1.0;
    wave_sync();
  }
  if (kp.scaling && !keep_scaling) {
    int pass = 0;
    double crit;
    do {
      // New increments into t[0..n) (columns) and t[n..k) (rows), computed from the OLD sx, sy.  An increment
      // is a maximum over the entries of a column / row, so the entries are simply streamed in storage order
      // (batched, independent loads) and folded in with LDS atomic maxima: same values as the nested
      // per-column / per-row loops of the reference (:698-717) -- a maximum does not depend on the order, and
      // |sy sx A| is the same product for the column and the row an entry belongs to.
      for (int e = lane; e < k; e += kWave) t[e] = 0.0;
      wave_sync();
      // The entries are streamed in batches of UB per lane; the pattern and the values of batch b + 1 are requested before the
      // scaling factors batch b points to are gathered (one memory round trip per batch on the dependent path, not two).
      constexpr int UB = 8;
      {
        int rr[UB], cc[UB];
        double pv[UB];
        auto fetchP = [&](const int p0) {
#pragma unroll
          for (int e = 0; e < UB; ++e) {
            const int p = p0 + e * kWave;
            const bool on = p < nnzP;
            rr[e] = on ? pl.Pi[p] : 0;
            cc[e] = on ? pl.Pcol[p] : 0;
            pv[e] = on ? it.Px[p] : 0.0;
          }
        };
        fetchP(lane);
        for (int p0 = lane; p0 < nnzP; p0 += kWave * UB) {
          int c0[UB];
          double v0[UB], sr[UB], sc[UB];
#pragma unroll
          for (int e = 0; e < UB; ++e) {
            sr[e] = w.sx[rr[e]];
            sc[e] = w.sx[cc[e]];
            c0[e] = cc[e];
            v0[e] = pv[e];
          }
          fetchP(p0 + kWave * UB);
#pragma unroll
          for (int e = 0; e < UB; ++e)
            if (p0 + e * kWave < nnzP)
              __hip_atomic_fetch_max(&t[c0[e]], fabs(c * sr[e] * sc[e] * v0[e]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      {
        int rr[UB], cc[UB];
        double av[UB];
        auto fetchA = [&](const int p0) {
#pragma unroll
          for (int e = 0; e < UB; ++e) {
            const int p = p0 + e * kWave;
            const bool on = p < nnzA;
            rr[e] = on ? pl.Arow[p] : 0;
            cc[e] = on ? pl.Aj[p] : 0;
            av[e] = on ? it.Ax[p] : 0.0;
          }
        };
        fetchA(lane);
        for (int p0 = lane; p0 < nnzA; p0 += kWave * UB) {
          int r0[UB], c0[UB];
          double v0[UB], sr[UB], sc[UB];
#pragma unroll
          for (int e = 0; e < UB; ++e) {
            sr[e] = w.sy[rr[e]];
            sc[e] = w.sx[cc[e]];
            r0[e] = rr[e];
            c0[e] = cc[e];
            v0[e] = av[e];
          }
          fetchA(p0 + kWave * UB);
#pragma unroll
          for (int e = 0; e < UB; ++e) {
            if (p0 + e * kWave < nnzA) {
              const double v = fabs(sr[e] * sc[e] * v0[e]);
              __hip_atomic_fetch_max(&t[c0[e]], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              __hip_atomic_fetch_max(&t[n + r0[e]], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          }
        }
      }
      wave_sync();
      double cm = 0.0;
      for (int e = lane; e < k; e += kWave) {
        double inc = t[e];
        if (inc == 0.0) inc = 1.0;
        t[e] = inc;
        cm   = fmax(cm, fabs(inc - 1.0));
      }
      wave_sync();
      for (int j = lane; j < n; j += kWave) w.sx[j] = sqrt(1.0 / fmax(t[j], 1e-8)) * w.sx[j];
      for (int i = lane; i < m; i += kWave) w.sy[i] = sqrt(1.0 / fmax(t[n + i], 1e-8)) * w.sy[i];
      crit = wave_max(cm);
      wave_sync();
    } while (pass++ < 10 && crit > 0.1);
  }

  // ---- pre-check, rho and loop constants :361-374, :450, :473-474 ----
  bool keep_factor = keep_scaling;
  {
    bool bad = false, rho_changed = false;
    for (int i = lane; i < m; i += kWave) {
      const double li = it.l[i], ui = it.u[i], syi = w.sy[i];
      bad = bad || (li == inf) || (ui == -inf) || (ui - li < 0.0);
      double rho;
      if (li == -inf && ui == inf) rho = 1e-6;
      else if (syi * fabs(li - ui) < 1e-5) rho = 1e3 * kp.rho_bar;
      else rho = kp.rho_bar;
      if (keep_scaling) rho_changed = rho_changed || !(w.rho[i] == rho);
      w.rho[i]  = rho;
      w.rinv[i] = 1.0 / rho;
      w.lo[i]   = syi * li;
      w.hi[i]   = syi * ui;
    }
    for (int j = lane; j < n; j += kWave) w.qc[j] = c * w.sx[j] * it.q[j];
    if (wave_ballot(bad)) ret_code = SFB_QP_PRIMAL_INFEASIBLE;
    if (keep_scaling && wave_ballot(rho_changed)) keep_factor = false;
  }
  wave_sync();

  // ---- KKT fill + numeric factorisation :379-433 ----
  phase_lap(0);
  if (!keep_factor) {
    if (lane == 0) w.hdr[kHdrStamp] = 0.0;  // until the new factor is complete
    if (!ldl_numeric_dev<8>(pl, it, w, t, 0, c, kp.sigma, kp.delta, lane, TRACE ? &ph_fill : nullptr)) {
      ret_code = SFB_QP_UNKNOWN;
      // Dinv of the remaining columns is never used: the loop below does not run
    } else if (lane == 0 && allow_reuse) {  // (a slot that is not the item's own -- ordered launch, pool -- stays unstamped)
      w.hdr[kHdrStamp]   = __longlong_as_double((long long)kFactorStamp);
      w.hdr[kHdrC]       = c;
      w.hdr[kHdrSigma]   = kp.sigma;
      w.hdr[kHdrScaling] = (double)kp.scaling;
    }
  }

  if constexpr (TRACE) {  // the factorisation's time, split into filling and the rest
    phase_lap(2);
    ph_t[1] = ph_fill;
    ph_t[2] -= ph_fill;
  }
#ifdef SFB_SP_TIMELINE
  tl1 = wall_clock64();
#endif
  // ---- initial iterate :436-445 ----
  if (gwx != nullptr) {
    const double *wx = gwx + b * (size_t)n, *wy = gwy + b * (size_t)m;
    for (int j = lane; j < n; j += kWave) w.xs[j] = (1.0 / w.sx[j]) * wx[j];
    for (int i = lane; i < m; i += kWave) {
      const double syi = w.sy[i];
      w.ys[i]          = c * ((1.0 / syi) * wy[i]);
      double acc       = 0.0;
      for (int q = pl.Ap[i], q1 = pl.Ap[i + 1]; q < q1; q += kRowChunk) {  // chunked like sp_row_dot
        int j[kRowChunk];
        double av[kRowChunk], xv[kRowChunk];
#pragma unroll
        for (int a = 0; a < kRowChunk; ++a) {
          j[a]  = (q + a < q1) ? pl.Aj[q + a] : 0;
          av[a] = (q + a < q1) ? it.Ax[q + a] : 0.0;
        }
#pragma unroll
        for (int a = 0; a < kRowChunk; ++a) xv[a] = wx[j[a]];
#pragma unroll
        for (int a = 0; a < kRowChunk; ++a)
          if (q + a < q1) acc = fma(syi * av[a], xv[a], acc);
      }
      w.zs[i] = acc;
    }
  } else {
    for (int j = lane; j < n; j += kWave) w.xs[j] = 0.0;
    for (int i = lane; i < m; i += kWave) {
      w.ys[i] = 0.0;
      w.zs[i] = 0.0;
    }
  }
  wave_sync();
  }  // !resume

  if (ph1 == PH_SETUP) {  // phased launch: the ADMM phase continues from here
    if (lane == 0) {
      w.hdr[0] = c;
      w.hdr[1] = 0.0;
      w.hdr[2] = (double)next_chk;
      w.hdr[3] = (double)t0_ticks;
#ifdef SFB_SP_TIMELINE
      w.hdr[4] = (double)tl0;
      w.hdr[5] = (double)tl1;
#endif
      w.hdr[kHdrCode]     = (double)ret_code;
      w.hdr[kHdrComplete] = 0.0;
    }
    return SP_DONE;
  }

  // ---- ADMM loop :447-510 ----
  const uint32_t iter0 = iter;  // start of this slice
  bool need_rhs        = true;
  // The vectors of the loop: in the item's workspace -- or, LAT, in LDS behind the work vector for as long as this wave
  // iterates on the item (x, z, y copied in here and back when the wave lets go of the item; c q, 1/rho, rho, the scaled
  // bounds and the permutation are constants of the loop).  A lone wave spent 6 of its 23.5 us per iteration waiting for
  // the five global round trips of the update phases, and the vectors are a quarter of the loop's traffic.
  double *vxs = w.xs, *vqc = w.qc, *vys = w.ys, *vzs = w.zs, *vrinv = w.rinv, *vrho = w.rho, *vlo = w.lo, *vhi = w.hi;
  const int32_t *vpinv = pl.pinv;
  [[maybe_unused]] double *vdinv = nullptr;  // LAT: 1 / D of the ADMM factor in LDS
  // LAT, round 5: the second launch now holds the whole batch and its length is its total work, i.e. (waves) / (time of an
  // iteration) -- so the LDS block of a LAT wave is cut from 77 KB (two waves per CU) to 45 KB (three): only what an iteration
  // WRITES or gathers stays on chip (x, y, z, 1 / D, the permutation as 16-bit entries); rho and 1 / rho take three values
  // (:361-374) and become a class byte per row; c q and the scaled bounds are read-only, read once per iteration with
  // coalesced loads at a point where nothing waits for them: they go back to the workspace (L2-resident).
  [[maybe_unused]] const uint16_t *vp16 = nullptr;
  [[maybe_unused]] const uint8_t *vcls  = nullptr;
  const double rho_c0 = 1e-6, rho_c1 = 1e3 * kp.rho_bar, rho_c2 = kp.rho_bar;  // the three values of :367-373 ...
  const double rinv_c0 = 1.0 / rho_c0, rinv_c1 = 1.0 / rho_c1, rinv_c2 = 1.0 / rho_c2;  // ... and their reciprocals as the setup forms them
  auto pinv_at = [&](const int e) -> int {
    if constexpr (LAT) return (int)vp16[e];
    else return vpinv[e];
  };
  auto rho_at = [&](const int i) -> double {
    if constexpr (LAT) { const int cl = vcls[i]; return cl == 0 ? rho_c0 : (cl == 1 ? rho_c1 : rho_c2); }
    else return vrho[i];
  };
  auto rinv_at = [&](const int i) -> double {
    if constexpr (LAT) { const int cl = vcls[i]; return cl == 0 ? rinv_c0 : (cl == 1 ? rinv_c1 : rinv_c2); }
    else return vrinv[i];
  };
  const bool iterates  = ph0 <= PH_ADMM && iter != maxit && ret_code < 0;
  // LAT, round 6: the ITERATE (x, y, z) lives in the wave's registers for as long as it iterates on the item -- lane l owns the
  // elements l, l + 64, ... (kLatRows rows per vector: n, m <= 64 kLatRows, the launcher's condition for this form) -- and what
  // the update phases read besides it, c q and the scaled bounds, sits in LDS where the iterate used to be: an iteration of the
  // loop makes no global access outside the factor stream (the round trips to the L2 for c q and the bounds cost 1.7 of the
  // 25.5 us of an iteration with the chip full, the iterate's LDS traffic comes on top; scripts/r6/iter_parts.sh).  Same
  // expressions on the same operands as the standard form below: bit-identical.
  // Between registers and the workspace the iterate always travels THROUGH LDS (unrolled register <-> LDS moves with immediate
  // offsets, then compact loops LDS <-> global): unrolled global accesses make the compiler hoist a hundred 64-bit addresses
  // out of the loop and park them in the AccVGPRs, which belong to the resident head of the factor stream.
  constexpr int kLatRows = kLatRegRows;
  if constexpr (LAT) {
    if (n > kLatRows * kWave || m > kLatRows * kWave) __builtin_trap();  // (never launched: qp_sparse_launch)
  }
  [[maybe_unused]] double xr[kLatRows], yr[kLatRows], zr[kLatRows];
  [[maybe_unused]] double *lqc = nullptr, *llo = nullptr, *lhi = nullptr;
  [[maybe_unused]] const double *lrtab = nullptr;
  // registers -> LDS block `dst` / back, lane l <-> elements l + 64 r (len <= 64 kLatRows)
  // (predicates as `lane < scalar`, addresses as one lane pointer plus immediates: no per-row lane index is kept in a register)
  auto regs_to_lds = [&](const double (&v)[kLatRows], double *dst, const int len) {
    double *const dl = dst + lane;
#pragma unroll
    for (int r = 0; r < kLatRows; ++r)
      if (lane < len - r * kWave) dl[r * kWave] = v[r];
  };
  auto lds_to_regs = [&](double (&v)[kLatRows], const double *src, const int len) {
    const double *const sl = src + lane;
#pragma unroll
    for (int r = 0; r < kLatRows; ++r) v[r] = (lane < len - r * kWave) ? sl[r * kWave] : 0.0;
  };
  // workspace -> LDS, six rows of 64 per memory round trip (the vectors of the LAT form are 12 or 24 rows, and a loaded wave waits
  // ~2 us for every trip: one row per trip was 4 % of the loop launch in item set-up alone)
  [[maybe_unused]] auto copy_in = [&](double *dst, const double *src, const int len) {
    constexpr int UB = 6;
    for (int e0 = lane; e0 < len; e0 += kWave * UB) {
      double v[UB];
#pragma unroll
      for (int e = 0; e < UB; ++e) v[e] = (e0 + e * kWave < len) ? src[e0 + e * kWave] : 0.0;
#pragma unroll
      for (int e = 0; e < UB; ++e)
        if (e0 + e * kWave < len) dst[e0 + e * kWave] = v[e];
    }
  };
  // the iterate from the workspace into the registers (t is free: x and y through t[0 .. n + m), z through the block of the upper
  // bounds, which is restored from the workspace afterwards)
  auto iterate_in = [&] {
    if constexpr (LAT) {
      copy_in(t, w.xs, n);
      copy_in(t + n, w.ys, m);
      copy_in(lhi, w.zs, m);
      wave_sync();
      lds_to_regs(xr, t, n);
      lds_to_regs(yr, t + n, m);
      lds_to_regs(zr, lhi, m);
      wave_sync();
      copy_in(lhi, w.hi, m);
      wave_sync();
    }
  };
  if constexpr (LAT) {
    double *cq = t + ((k + 2) & ~1), *clo = cq + n, *chi = clo + m;
    lqc = cq; llo = clo; lhi = chi;
    vdinv = chi + m;
    constexpr int kPad = kLatRows * kWave;
    double *ltab = vdinv + k;  // {1 / rho of class 0, 1, 2, rho of class 0, 1, 2, -, -}: a class byte is 8 x class
    lrtab = ltab;
    uint16_t *lp = reinterpret_cast<uint16_t *>(ltab + 8);  // [x: kPad][y / z: kPad] byte offsets into t
    uint8_t *lc  = reinterpret_cast<uint8_t *>(lp + 2 * kPad);
    vp16 = lp;
    vcls = lc;
#pragma unroll
    for (int r = 0; r < kLatRows; ++r) xr[r] = yr[r] = zr[r] = 0.0;
    if (iterates) {
      copy_in(cq, w.qc, n);
      copy_in(clo, w.lo, m);
#pragma unroll 1  // (a constant trip count: unrolled, its lane indices are hoisted and kept for the life of the kernel)
      for (int e0 = lane; e0 < kPad; e0 += 4 * kWave) {  // (four rows per round trip)
        int px[4], py[4];
        double rho[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int e = e0 + q * kWave;
          px[q]  = e < n ? pl.pinv[e] : k;
          py[q]  = e < m ? pl.pinv[n + e] : k;
          rho[q] = e < m ? w.rho[e] : rho_c2;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int e  = e0 + q * kWave;
          lp[e]        = (uint16_t)(8 * px[q]);
          lp[kPad + e] = (uint16_t)(8 * py[q]);
          lc[e]        = (uint8_t)(8 * (rho[q] == rho_c0 ? 0 : (rho[q] == rho_c1 ? 1 : 2)));
        }
      }
      if (lane < 3) {
        ltab[lane]     = lane == 0 ? rinv_c0 : (lane == 1 ? rinv_c1 : rinv_c2);
        ltab[3 + lane] = lane == 0 ? rho_c0 : (lane == 1 ? rho_c1 : rho_c2);
      }
      copy_in(vdinv, w.Dinv, k);
      iterate_in();
    }
  }
  // LAT, round 5: the head of the forward stream resident in the wave's AccVGPRs for as long as it iterates on the item (the
  // launch runs at the fabric's read rate: what is on chip is not streamed)
  [[maybe_unused]] bool lat_resident = false;
  if constexpr (LAT) {
    lat_resident = iterates && uni(pl.idx_scale) == 8 && uni(pl.bunits) >= 8 && lat_resident_ok(pl);
    if (lat_resident) lat_resident_load(w.LxF, lane);
  }
  auto vectors_home = [&] {  // LAT: x, z, y back to the workspace (polish, report, or the next wave that takes the item up)
    if constexpr (LAT) {
      if (iterates) {  // (t is free wherever this is called: behind a stopping check or at the end of the loop)
        wave_sync();
        regs_to_lds(xr, t, n);
        regs_to_lds(yr, t + n, m);
        regs_to_lds(zr, lhi, m);
        wave_sync();
        for (int j = lane; j < n; j += kWave) w.xs[j] = t[j];
        for (int i = lane; i < m; i += kWave) { w.ys[i] = t[n + i]; w.zs[i] = lhi[i]; }
        wave_sync();
      }
    }
  };
#if defined(SFB_ITER_EXP) && SFB_ITER_EXP >= 3  // ... c q and the scaled bounds read from LDS (whatever is there)
  if constexpr (LAT) { vqc = vxs; vlo = vys; vhi = vzs; }
#endif
  [[maybe_unused]] const bool lat_chain = LAT && uni(pl.idx_scale) == 8 && uni(pl.funits) >= 8 && uni(pl.bunits) >= 8;
  for (; ph0 <= PH_ADMM && iter != maxit && ret_code < 0; ++iter) {
    // element-wise phases: the loads of UNR strided elements are issued together (one memory round
    // trip per UNR elements instead of one per element -- matters for a wave that runs alone)
    // (the batch sizes are what the 168-VGPR budget of three waves per SIMD allows)
    constexpr int UNR_A = LAT ? 8 : 4, UNR_B = 4;
    // The right-hand side of the NEXT solve is written by the update phase below from the values it has just
    // computed (same expressions, no re-read of x, z, y, 1/rho); only the first iteration and the one after a
    // stopping check (which uses t as scratch) build it here.
    // (LAT) an element of the work vector by its byte offset; rows of 64 lanes run without predicates: the padding lanes of a
    // vector's last row (and whole padding rows of a smaller problem) carry the scratch slot t[k] and whatever the LDS holds
    // behind the constants
    [[maybe_unused]] auto t_at = [&](const int off) -> double & { return *reinterpret_cast<double *>(reinterpret_cast<char *>(t) + off); };
    // (the loop's scalars in VGPRs: the kernel's SGPRs are spilled to VGPR lanes and every use of a spilled one is a v_readlane)
    [[maybe_unused]] double v_alpha = kp.alpha, v_alphac = kp.alpha_comp, v_sigma = kp.sigma;
    if constexpr (LAT) asm volatile("" : "+v"(v_alpha), "+v"(v_alphac), "+v"(v_sigma));
    [[maybe_unused]] auto tab_at = [&](const int off) -> double { return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(lrtab) + off); };
    [[maybe_unused]] auto rhs_from_regs = [&] {
      constexpr int kPad = kLatRows * kWave;
      int pv[kLatRows];
      double qv[kLatRows];
#pragma unroll
      for (int r = 0; r < kLatRows; ++r) {  // :450
        pv[r] = (int)vp16[lane + r * kWave];
        qv[r] = lqc[lane + r * kWave];
      }
#pragma unroll
      for (int r = 0; r < kLatRows; ++r) t_at(pv[r]) = v_sigma * xr[r] - qv[r];
#pragma unroll
      for (int r = 0; r < kLatRows; ++r) {  // :451
        pv[r] = (int)vp16[kPad + lane + r * kWave];
        qv[r] = tab_at((int)vcls[lane + r * kWave]);
      }
#pragma unroll
      for (int r = 0; r < kLatRows; ++r) t_at(pv[r]) = zr[r] - qv[r] * yr[r];
    };
    if constexpr (LAT) {
      if (iter == next_chk) {
        // the old iterate is parked in dx_us / dy_us (the oracle's memcpy, qp_solver.hpp:466-467; x and y do not change during
        // the solve, so it can happen here) -- through t, whose right-hand side is then built again from the registers: the same
        // expressions on the same values the update phase wrote it from
        wave_sync();
        regs_to_lds(xr, t, n);
        regs_to_lds(yr, t + n, m);
        wave_sync();
        for (int j = lane; j < n; j += kWave) w.dxus[j] = t[j];
        for (int i = lane; i < m; i += kWave) w.dyus[i] = t[n + i];
        wave_sync();
        need_rhs = true;
      }
      if (need_rhs) rhs_from_regs();
    } else
    if (need_rhs) {
    constexpr int UNR = UNR_A;
    for (int j0 = lane; j0 < n; j0 += kWave * UNR) {  // :450
      double xv[UNR], qv[UNR];
      int pv[UNR];
#pragma unroll
      for (int e = 0; e < UNR; ++e) {
        const int j = j0 + e * kWave;
        const bool on = j < n;
        xv[e] = on ? vxs[j] : 0.0;
        qv[e] = on ? vqc[j] : 0.0;
        pv[e] = on ? pinv_at(j) : k;
      }
#pragma unroll
      for (int e = 0; e < UNR; ++e)
        if (j0 + e * kWave < n) t[pv[e]] = kp.sigma * xv[e] - qv[e];
    }
    for (int i0 = lane; i0 < m; i0 += kWave * UNR) {  // :451
      double zv[UNR], rv[UNR], yv[UNR];
      int pv[UNR];
#pragma unroll
      for (int e = 0; e < UNR; ++e) {
        const int i = i0 + e * kWave;
        const bool on = i < m;
        zv[e] = on ? vzs[i] : 0.0;
        rv[e] = on ? rinv_at(i) : 0.0;
        yv[e] = on ? vys[i] : 0.0;
        pv[e] = on ? pinv_at(n + i) : k;
      }
#pragma unroll
      for (int e = 0; e < UNR; ++e)
        if (i0 + e * kWave < m) t[pv[e]] = zv[e] - rv[e] * yv[e];
    }
    }
    const bool chk = (iter == next_chk);
    wave_sync();
#if defined(SFB_ITER_EXP) && SFB_ITER_EXP == 2  // measurement builds (scripts/r6/iter_parts.sh; results are garbage): no sweeps
    if (false)
#endif
    if constexpr (LAT) {
      if (lat_chain && !lean) {
        // chained sweeps (ldl_solve_lat): the backward stream's head is fetched by the forward sweep's last block.  Carrying the
        // chain on into the next iteration's forward sweep (prefetch registers live across the update phases) was built and
        // is NOT done: the compiler is free to copy loop-carried registers while their loads are in flight (wrong results).
        vdouble2 lat_lx[8];
        vint2 lat_ix[8];
        ldl_solve_lat(pl, w, t, lane, vdinv, lat_lx, lat_ix, false, false, lat_resident);
      } else {
        ldl_solve_dev<8>(pl, w, t, lane, lean, vdinv);
      }
    } else {
      ldl_solve_dev<SFB_SWEEP_DEPTH>(pl, w, t, lane, lean, nullptr);  // :456-460
    }
    if (chk) next_chk += sci;
    need_rhs = chk;
    // UPDATE PHASES :470-477.  Whole rows of 64 elements run without predicates, U rows per batch (their loads are issued
    // together: one round trip per batch); only the last, partial row is predicated.  Nothing of the stopping check is in
    // here: per element it cost a lone wave more instructions than the update itself (the check's pointers and divisions,
    // reloaded from spilled SGPRs).  At a check the old iterate is parked in dx_us / dy_us first (the oracle's memcpy,
    // qp_solver.hpp:466-467) and the unscaled vectors are formed by a pass of their own below -- same expressions, same bits.
    if constexpr (!LAT) {
      if (chk) {
        for (int j = lane; j < n; j += kWave) w.dxus[j] = vxs[j];
        for (int i = lane; i < m; i += kWave) w.dyus[i] = vys[i];
      }
    }
    auto xrows = [&]<int U, bool PRED>(std::integral_constant<int, U>, std::bool_constant<PRED>, const int r0) {
      double xo[U], qv[U];
      int pv[U];
#pragma unroll
      for (int e = 0; e < U; ++e) {
        const int j   = lane + (r0 + e) * kWave;
        const bool on = !PRED || j < n;
        xo[e] = on ? vxs[j] : 0.0;
        qv[e] = on ? vqc[j] : 0.0;
        pv[e] = on ? pinv_at(j) : k;
      }
#pragma unroll
      for (int e = 0; e < U; ++e) {
        const int j = lane + (r0 + e) * kWave;
        if (!PRED || j < n) {
          const double xn = kp.alpha * t[pv[e]] + kp.alpha_comp * xo[e];
          vxs[j]   = xn;
          t[pv[e]] = kp.sigma * xn - qv[e];  // rhs of the next solve (:450)
        }
      }
    };
    auto yrows = [&]<int U, bool PRED>(std::integral_constant<int, U>, std::bool_constant<PRED>, const int r0) {
      double yo[U], zo[U], ri[U], rh[U], lo[U], hi[U];
      int pv[U];
#pragma unroll
      for (int e = 0; e < U; ++e) {
        const int i   = lane + (r0 + e) * kWave;
        const bool on = !PRED || i < m;
        yo[e] = on ? vys[i] : 0.0;
        zo[e] = on ? vzs[i] : 0.0;
        ri[e] = on ? rinv_at(i) : 0.0;
        rh[e] = on ? rho_at(i) : 0.0;
        lo[e] = on ? vlo[i] : 0.0;
        hi[e] = on ? vhi[i] : 0.0;
        pv[e] = on ? pinv_at(n + i) : k;
      }
#pragma unroll
      for (int e = 0; e < U; ++e) {
        const int i = lane + (r0 + e) * kWave;
        if (!PRED || i < m) {
          const double nu = t[pv[e]];
          double zn = kp.alpha * (ri[e] * nu) + kp.alpha_comp * (ri[e] * yo[e]) + zo[e];
          zn = (zn < lo[e]) ? lo[e] : zn;
          zn = (hi[e] < zn) ? hi[e] : zn;
          const double yn = kp.alpha_comp * yo[e] + kp.alpha * nu + rh[e] * zo[e] - rh[e] * zn;
          vys[i]   = yn;
          vzs[i]   = zn;
          t[pv[e]] = zn - ri[e] * yn;  // rhs of the next solve (:451)
        }
      }
    };
    auto all_rows = [&]<int U>(std::integral_constant<int, U> uu, const int len, auto &&rows) {
      const int full = len / kWave;  // rows in which every lane has an element
      int r = 0;
      for (; r + U <= full; r += U) rows(uu, std::false_type{}, r);
      if constexpr (U > 4) if (r + 4 <= full) { rows(std::integral_constant<int, 4>{}, std::false_type{}, r); r += 4; }
      if constexpr (U > 2) if (r + 2 <= full) { rows(std::integral_constant<int, 2>{}, std::false_type{}, r); r += 2; }
      if constexpr (U > 1) if (r + 1 <= full) { rows(std::integral_constant<int, 1>{}, std::false_type{}, r); r += 1; }
      if (full * kWave < len) rows(std::integral_constant<int, 1>{}, std::true_type{}, r);
    };
#if defined(SFB_ITER_EXP) && SFB_ITER_EXP == 1  // ... no update phases
    if (false)
#endif
    if constexpr (LAT) {
      // the register form of the two phases above: every LDS read of a batch of rows is issued before the first result is used
      constexpr int kPad = kLatRows * kWave;
      auto xbatch = [&]<int R0, int RN>(std::integral_constant<int, R0>, std::integral_constant<int, RN>) {
        if (R0 * kWave >= n) return;
        int pv[RN];
        double qv[RN], tv[RN];
#pragma unroll
        for (int e = 0; e < RN; ++e) {
          pv[e] = (int)vp16[lane + (R0 + e) * kWave];
          qv[e] = lqc[lane + (R0 + e) * kWave];
        }
#pragma unroll
        for (int e = 0; e < RN; ++e) tv[e] = t_at(pv[e]);
#pragma unroll
        for (int e = 0; e < RN; ++e) {
          const double xn = v_alpha * tv[e] + v_alphac * xr[R0 + e];
          xr[R0 + e]  = xn;
          t_at(pv[e]) = v_sigma * xn - qv[e];  // rhs of the next solve (:450)
        }
      };
      auto ybatch = [&]<int R0, int RN>(std::integral_constant<int, R0>, std::integral_constant<int, RN>) {
        if (R0 * kWave >= m) return;
        int pv[RN], cl[RN];
        double lo[RN], hi[RN], tv[RN], riv[RN], rhv[RN];
#pragma unroll
        for (int e = 0; e < RN; ++e) {
          const int i = lane + (R0 + e) * kWave;
          pv[e] = (int)vp16[kPad + i];
          cl[e] = (int)vcls[i];
          lo[e] = llo[i];
          hi[e] = lhi[i];
        }
#pragma unroll
        for (int e = 0; e < RN; ++e) {
          tv[e]  = t_at(pv[e]);
          riv[e] = tab_at(cl[e]);
          rhv[e] = tab_at(cl[e] + 24);
        }
#pragma unroll
        for (int e = 0; e < RN; ++e) {
          const double ri = riv[e], rh = rhv[e];
          const double nu = tv[e], yo = yr[R0 + e], zo = zr[R0 + e];
          double zn = v_alpha * (ri * nu) + v_alphac * (ri * yo) + zo;
          zn = (zn < lo[e]) ? lo[e] : zn;
          zn = (hi[e] < zn) ? hi[e] : zn;
          const double yn = v_alphac * yo + v_alpha * nu + rh * zo - rh * zn;
          yr[R0 + e]  = yn;
          zr[R0 + e]  = zn;
          t_at(pv[e]) = zn - ri * yn;  // rhs of the next solve (:451)
        }
      };
      static_assert(kLatRows == 12, "batches of the register form");
      xbatch(std::integral_constant<int, 0>{}, std::integral_constant<int, 6>{});
      xbatch(std::integral_constant<int, 6>{}, std::integral_constant<int, 6>{});
      ybatch(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
      ybatch(std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{});
      ybatch(std::integral_constant<int, 8>{}, std::integral_constant<int, 4>{});
    } else
    {
    all_rows(std::integral_constant<int, UNR_A>{}, n, xrows);
    all_rows(std::integral_constant<int, UNR_B>{}, m, yrows);
    }
    if constexpr (LAT) {
      if (chk) {
        // the new iterate goes HOME here (x, y through t, z through the block of the upper bounds: the pass below then reads them
        // where the standard form has them), and comes back behind the check if the loop goes on: during the check no register
        // of the wave holds it
        vectors_home();
        vxs = t; vys = t + n; vzs = lhi;
      }
    }
    if (chk) {  // :481-488 unscaled iterate and differences for the check
      // (LAT: six rows of 64 per memory round trip -- the same expressions element by element)
      constexpr int UBU = LAT ? 6 : 1;
      for (int j0 = lane; j0 < n; j0 += kWave * UBU) {
        double sxv[UBU], xov[UBU];
#pragma unroll
        for (int e = 0; e < UBU; ++e) {
          const bool on = j0 + e * kWave < n;
          sxv[e] = on ? w.sx[j0 + e * kWave] : 0.0;
          xov[e] = on ? w.dxus[j0 + e * kWave] : 0.0;
        }
#pragma unroll
        for (int e = 0; e < UBU; ++e) {
          const int j = j0 + e * kWave;
          if (j < n) {
            const double xn = vxs[j], sxj = sxv[e], xo = xov[e];
            w.xus[j]  = sxj * xn;
            t[j]      = sxj * xn;  // (the check's mat-vecs gather x and y from the work vector, free until the next right-hand side)
            w.dxus[j] = sxj * (xn - xo);
          }
        }
      }
      for (int i0 = lane; i0 < m; i0 += kWave * UBU) {
        double syv[UBU], yov[UBU];
#pragma unroll
        for (int e = 0; e < UBU; ++e) {
          const bool on = i0 + e * kWave < m;
          syv[e] = on ? w.sy[i0 + e * kWave] : 1.0;
          yov[e] = on ? w.dyus[i0 + e * kWave] : 0.0;
        }
#pragma unroll
        for (int e = 0; e < UBU; ++e) {
          const int i = i0 + e * kWave;
          if (i < m) {
            const double yn = vys[i], zn = vzs[i], syi = syv[e], yo = yov[e];
            w.yus[i]  = syi * yn / c;
            t[n + i]  = syi * yn / c;
            w.zus[i]  = (1.0 / syi) * zn;
            w.dyus[i] = syi * (yn - yo) / c;
          }
        }
      }
    }
    wave_sync();
    if (chk) {
      ret_code = sp_check_stopping<LAT ? 6 : 4>(pl, it, w, kp, t, lane, pause_at != 0 ? score : nullptr);
      if constexpr (TRACE) {  // the reference's verbose table as data (:490-501)
        if (trace != nullptr && trace_rows < trace_cap) sp_trace_row(pl, it, w, t, lane, iter, t0_ticks, trace + ((size_t)b * (size_t)trace_cap + (size_t)trace_rows) * 5);
        ++trace_rows;
      }
      if (ret_code < 0 && max_time_exceeded(kp.max_time_ns, t0_ticks)) ret_code = SFB_QP_MAX_TIME;  // :504-507
      wave_sync();
      if constexpr (LAT) iterate_in();  // (restores the block of the upper bounds as well)
      lean = __builtin_amdgcn_readfirstlane(__hip_atomic_load(pl.dev_active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >
             lean_waves;
      // Items that are still iterating after many checks are the ones the whole launch waits for:
      // raise their issue priority over the co-resident waves that are in their first iterations.
      if (iter > 600) __builtin_amdgcn_s_setprio(3);
      else if (iter > 300) __builtin_amdgcn_s_setprio(2);
      else if (iter > 100) __builtin_amdgcn_s_setprio(1);
      // Pause (launches in predicted order, see qp_sparse_launch): the item has iterated long enough for its residual
      // to tell how long it will go on; it waits in its workspace for the launch that orders the survivors.
      if (pause_at != 0 && ret_code < 0 && iter + 1 >= pause_at && iter + 1 != maxit) {
        if (lane == 0) {
          w.hdr[0] = c;
          w.hdr[1] = (double)(iter + 1);
          w.hdr[2] = (double)next_chk;
          w.hdr[3] = (double)t0_ticks;
#ifdef SFB_SP_TIMELINE
          w.hdr[4] = (double)tl0;
          w.hdr[5] = (double)tl1;
#endif
          w.hdr[kHdrCode]     = -1.0;
          w.hdr[kHdrComplete] = 0.0;
        }
        __builtin_amdgcn_s_setprio(0);
        vectors_home();
        return SP_PAUSED;
      }
      // Time slicing: an item that has used its slice gives its wave back when fresh items are left or suspended
      // ones are waiting -- the long runners then share the waves round-robin and finish together, instead of
      // the ones that happened to start late running alone at the end of the launch.  Everything an item needs
      // to continue is in its workspace already (iterate, factor, scaling); a stopping check is the natural
      // point: the next iteration rebuilds its right-hand side anyway.
      if (queue != nullptr && ret_code < 0 && iter + 1 != maxit && iter + 1 - iter0 >= slice) {
        int wait = 0;
        if (lane == 0) {
          const int fresh = __hip_atomic_load(&queue[kQFresh], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int head  = __hip_atomic_load(&queue[kQHead], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int tail  = __hip_atomic_load(&queue[kQTail], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          wait            = (fresh < batch) || (tail - head > 0);
        }
        if (__builtin_amdgcn_readfirstlane(wait)) {
          if (lane == 0) {
            w.hdr[0] = c;
            w.hdr[1] = (double)(iter + 1);
            w.hdr[2] = (double)next_chk;
            w.hdr[3] = (double)t0_ticks;  // (exact: below 2^53 ticks of 10 ns)
#ifdef SFB_SP_TIMELINE
            w.hdr[4] = (double)tl0;
            w.hdr[5] = (double)tl1;
#endif
          }
          __builtin_amdgcn_s_setprio(0);
          vectors_home();
          return SP_SUSPENDED;
        }
      }
    }
  }
  __builtin_amdgcn_s_setprio(0);
  vectors_home();

#ifdef SFB_SP_TIMELINE
  tl2 = wall_clock64();
#endif
  if (ph1 == PH_ADMM) {  // phased launch: polish and report belong to the next phase
    if (lane == 0) {
      w.hdr[0] = c;
      w.hdr[1] = (double)iter;
      w.hdr[kHdrCode] = (double)ret_code;
#ifdef SFB_SP_TIMELINE
      w.hdr[kHdrTl2] = (double)tl2;
#endif
    }
    return SP_DONE;
  }
#ifdef SFB_SP_TIMELINE
  if (ph0 == PH_FINISH) tl2 = (unsigned long long)w.hdr[kHdrTl2];
#endif
  phase_lap(3);
  // ---- polish :515-539 ----
  if (ret_code == SFB_QP_OPTIMAL && kp.polish) {
    if (kp.reuse != 0 && allow_reuse) {  // keep the ADMM factor for the next call: the polish system goes to the second block
      const Ws wp = polish_ws(w, gws + slot * ws_doubles,
                              qp_sparse_polish_offset(n, m, uni(pl.nnzL), uni(pl.funits), uni(pl.bunits), pl.Aorig ? nnzA : 0), n, m,
                              uni(pl.nnzL), uni(pl.funits), uni(pl.bunits));
      sp_polish<SFB_SWEEP_DEPTH, WIDE>(pl, it, w, wp, kp, t, c, lane, lean);
    } else {
      if (lane == 0) w.hdr[kHdrStamp] = 0.0;  // the polish factorisation overwrites the ADMM factor
      sp_polish<SFB_SWEEP_DEPTH, WIDE>(pl, it, w, w, kp, t, c, lane, lean);
    }
  }

  phase_lap(4);
  // ---- un-scale and report :544-548 ----
  double *ox = gx + b * (size_t)n, *oy = gy + b * (size_t)m;
  for (int j = lane; j < n; j += kWave) {
    const double v = w.sx[j] * w.xs[j];
    ox[j]    = v;
    w.xus[j] = v;
  }
  for (int i = lane; i < m; i += kWave) oy[i] = w.sy[i] * w.ys[i] / c;
  wave_sync();
  if (gobj != nullptr) {
    // primal . (0.5 P primal + q): the rows in parallel, the dot product as the sequential fma chain it is -- fed from
    // LDS in chunks of pairs (x_i, row_i), 8 pairs per LDS round trip (one global load per term cost a lone wave ~0.1 ms)
    const int chunk = (k + 1) / 2;
    double o        = 0.0;
    for (int c0 = 0; c0 < n; c0 += chunk) {
      const int c1 = min(n, c0 + chunk);
      struct Ix { int a, b; };
      struct V2 { double a, b; };
      constexpr int RB = 4, CE = 2;  // rows of 0.5 P x, RB per lane together (one entry at a time cost two round trips per ENTRY)
      for (int i0 = c0 + lane; i0 < c1; i0 += kWave * RB) {
        double acc[RB], xi[RB], qi[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          const int i = i0 + r * kWave;
          acc[r] = 0.0;
          xi[r]  = (i < c1) ? w.xus[i] : 0.0;
          qi[r]  = (i < c1) ? it.q[i] : 0.0;
        }
        sp_rows_chain<RB, CE>(acc, i0, c1, [&](int i) { return Ix{pl.Prp[i], pl.Prp[i + 1]}; },
                              [&](int q) { return Ix{pl.Prpos[q], pl.Prj[q]}; },
                              [&](const Ix &x) { return V2{it.Px[x.a], w.xus[x.b]}; },
                              [&](int, const Ix &, const V2 &v, double a) { return fma(0.5 * v.a, v.b, a); });
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          const int i = i0 + r * kWave;
          if (i < c1) {
            t[2 * (i - c0)]     = xi[r];
            t[2 * (i - c0) + 1] = acc[r] + qi[r];
          }
        }
      }
      wave_sync();
      const int cnt = c1 - c0;
      int e = 0;
      for (; e + 8 <= cnt; e += 8) {
        vdouble2 v[8];
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) v[qq] = *reinterpret_cast<const vdouble2 *>(t + 2 * (e + qq));
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) o = fma(v[qq].x, v[qq].y, o);
      }
      for (; e < cnt; ++e) o = fma(t[2 * e], t[2 * e + 1], o);
      wave_sync();
    }
    if (lane == 0) gobj[b] = o;
  }
  if (lane == 0) {
    gcode[b] = (ret_code >= 0) ? ret_code : SFB_QP_MAX_ITERATIONS;
    if (giter != nullptr) giter[b] = iter;
#ifdef SFB_SP_TIMELINE  // the stamps replace the first four dual entries (profiling build only)
    oy[0] = (double)tl0; oy[1] = (double)tl1; oy[2] = (double)tl2; oy[3] = (double)wall_clock64();
#endif
  }
  if constexpr (TRACE) {
    wave_sync();  // (the report's stores are on their way: the lap below ends the item's time in the wave)
    phase_lap(5);
    if (phase_us != nullptr && lane == 0)
      for (int i = 0; i < 6; ++i) phase_us[b * 6 + i] = (double)ph_t[i] * 0.01;  // 100 MHz wall clock
  }
  return SP_DONE;
}

}  // namespace

// Launch-wide auxiliary memory (zeroed by the launcher): the queue of a time-sliced launch, then the flags of the
// fallback pool of a pruned plan (kFbSlots ints, 0 = free).
constexpr int kFbSlots = 64;
// ring of the items whose ADMM loop has ended (see the kernel): counters on their own cache lines, then one entry per item
constexpr int kDqTail = 0, kDqHead = 32, kDqOver = 64, kDqOwn = 80, kDqRing = 96;  // ([kDqOwn]: busy waves of the loop launch + its polishers)
constexpr int kDqStarted = 88;  // != 0: the loop launch's waves are on the chip (its block 0 has begun), see sp_gate_kernel

// POLISHER: the instance the polishers run (dq_mode 2); the standard instance has no register to spare for their loop.
template<bool LAT, bool TRACE = false, bool POLISHER = false>
__global__ void __launch_bounds__(64, LAT ? 1 : ((TRACE || POLISHER) ? 2 : 3)) qp_sparse_kernel(const SparsePlanDev *__restrict__ plp, const DenseKernelParams kp,
                                                       const double *__restrict__ gPx, const double *__restrict__ gq,
                                                       const double *__restrict__ gAx, const double *__restrict__ gl,
                                                       const double *__restrict__ gu, const double *__restrict__ gwx,
                                                       const double *__restrict__ gwy, double *__restrict__ gx,
                                                       double *__restrict__ gy, double *__restrict__ gobj,
                                                       uint32_t *__restrict__ giter, int32_t *__restrict__ gcode,
                                                       double *__restrict__ gws, const size_t ws_doubles,
                                                       const int lean_waves, const int32_t *__restrict__ order,
                                                       int32_t *__restrict__ queue, const int batch, const uint32_t slice,
                                                       const SparsePlanDev *__restrict__ plf, double *__restrict__ gwsf,
                                                       const size_t wsf_doubles, int32_t *__restrict__ fbflags, const int phases, const int nfb,
                                                       float *__restrict__ keys, const int32_t *__restrict__ nfresh_dev,
                                                       const int mode_sel, double *__restrict__ trace, const int trace_cap,
                                                       double *__restrict__ phase_us)
{
  // (no kernel arguments of their own -- the standard form has no register to spare: the mode rides in `phases`, the ring sits
  //  behind the other auxiliary arrays of the launch: [fallback flags: kFbSlots][scores: batch][order: batch][count: 16][ring])
  constexpr int dq_mode = POLISHER ? 2 : 0;  // (the producer side: every loop launch of the LAT instance pushes, see below)
  // doneq / dq_mode (launches in predicted order with the LAT loop launch): the ring of items whose ADMM loop has ended.
  //   dq_mode 1 / 3 (the loop launch): a wave that ends an item's loop pushes it (3: and helps polishing once it has no item left);
  //   dq_mode 2 (the POLISHERS, a small launch of this kernel's standard form on a second stream, running NEXT TO the loop launch
  //   on the SIMD and the LDS its three LAT waves per CU leave free): a wave pops an item, polishes and reports it (phase FINISH)
  //   and marks it complete; it leaves when the host has flagged the end of the loop launch.  The finish launch that follows
  //   skips what the polishers did.  [kDqTail] pushes, [kDqHead] pops, [kDqOver] != 0: the loop launch has ended, [kDqRing + i].
  // keys (nullable): per item, the predictor score of a paused item (SP_PAUSED) or -1 (nothing left to do for the next launch)
  // nfresh_dev (nullable): the fresh items of this launch are order[0 .. *nfresh_dev - 1] (the survivors of the previous one)
  const int ph0    = phases & 15;
  if constexpr (LAT) {  // (the loop launch announces itself to the gate in front of its polishers, whichever form the rank kernel chose)
    if (blockIdx.x == 0 && threadIdx.x == 0 && ((phases >> 28) & 1) == 1)
      __hip_atomic_store(&(fbflags + kFbSlots + 2 * batch + 16)[kDqStarted], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const int nfresh = nfresh_dev ? __builtin_amdgcn_readfirstlane(nfresh_dev[0]) : batch;
  // ... worked on by the first nfresh_dev[1] blocks of the grid only (the rank kernel sizes the launch, see there)
  const int nwaves = nfresh_dev ? __builtin_amdgcn_readfirstlane(nfresh_dev[1]) : (queue == nullptr ? (int)gridDim.x : batch);
  if (nfresh_dev != nullptr && (int)blockIdx.x >= nwaves) return;
  // ... and only if the rank kernel chose this launch's form (nfresh_dev[2]; mode_sel < 0: unconditional)
  if (nfresh_dev != nullptr && mode_sel >= 0 && __builtin_amdgcn_readfirstlane(nfresh_dev[2]) != mode_sel) return;
  extern __shared__ __attribute__((aligned(16))) double t[];  // work / solution vector, factorisation scratch
  // The plan (some forty pointers) is read from device memory where it is used: as a by-value kernel argument it
  // would sit in SGPRs for the whole life of the loop below and push the kernel into register spills.
  const int lane = threadIdx.x;
  // queue == nullptr: ONE ITEM PER BLOCK (batches that fit the chip at once).
  // queue != nullptr: TIME-SLICED LAUNCH: a persistent grid (as many blocks as the chip holds) works through the
  // batch.  Fresh items first, in launch order; an item that has used its slice while others wait goes to the back
  // of a ring and is continued later by whichever block is free (its state lives in ITS workspace slot = item).
  for (bool first = true;; first = false) {
    int item = -1, resume = 0;
    if constexpr (dq_mode == 2) {
      // (as many polishers as the rank kernel asked for, count[10]: one per compute unit when the iterations are the bulk of
      //  the work, two when polish is -- see sp_rank_kernel; the launch always holds the larger number)
      if (first && (int)blockIdx.x >= __builtin_amdgcn_readfirstlane((fbflags + kFbSlots + 2 * batch)[10])) return;
      if (lane == 0) {
        int32_t *const doneq = fbflags + kFbSlots + 2 * batch + 16;
        for (;;) {
          // (the loop launch has ended: what is still in the ring is done faster by the finish launch on the whole chip)
          if (__hip_atomic_load(&doneq[kDqOver], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
          const int head = __hip_atomic_load(&doneq[kDqHead], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int tail = __hip_atomic_load(&doneq[kDqTail], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (head < tail) {
            if (atomicCAS(&doneq[kDqHead], head, head + 1) != head) continue;
            int v;
            while ((v = __hip_atomic_load(&doneq[kDqRing + head], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) __builtin_amdgcn_s_sleep(4);
            item = v - 1;
            break;
          }
          __builtin_amdgcn_s_sleep(100);
        }
      }
      resume = 1;  // (an acquire below: the item's state was written by a wave of the loop launch)
    } else if (queue == nullptr) {
      if (first) item = order ? order[blockIdx.x] : (int)blockIdx.x;
    } else if (lane == 0) {
      if (__hip_atomic_load(&queue[kQFresh], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nfresh) {
        const int tk = atomicAdd(&queue[kQFresh], 1);
        if (tk < nfresh) item = order ? order[tk] : tk;
      }
      while (item < 0) {
        const int head = __hip_atomic_load(&queue[kQHead], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int tail = __hip_atomic_load(&queue[kQTail], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tail - head <= 0) break;  // nobody waits: once the fresh items are out, no item is suspended any more
        if (atomicCAS(&queue[kQHead], head, head + 1) != head) continue;
        int32_t *e = &queue[kQRing + (unsigned)head % (unsigned)batch];
        int v;
        while ((v = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) __builtin_amdgcn_s_sleep(4);
        __hip_atomic_store(e, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        item   = v - 1;
        resume = 1;
      }
      if constexpr (LAT) {
        // A wave of the loop launch that finds no item left to iterate HELPS THE POLISHERS while their ring has a backlog (the
        // tail of the launch belongs to a few long items: most of its waves would leave here), and leaves when it is empty.
        // (Only while nothing ELSE wants the chip -- a LAT wave owns its SIMD's whole register file: with a second batch on
        //  another stream the chip does better when the wave leaves, that batch's waves take the SIMD and the finish launch
        //  polishes three to a SIMD.  Two signs, both kept per device (SparsePlanDev::dev_busy / dev_active): the HOST marks a
        //  caller stream busy from the moment a solve is ENQUEUED on it until it is known finished -- a launch that is still
        //  waiting for a SIMD shows there --, and [kDqOwn] against dev_active shows the busy waves of other launches.
        //  nfresh_dev[8] = this launch's own stream bit.)
        if (item < 0 && ((phases >> 28) & 3) == 3) {
          int32_t *const doneq = fbflags + kFbSlots + 2 * batch + 16;
          const unsigned long long own_bit = 1ull << (nfresh_dev[8] & 63);
          for (;;) {
            if ((__hip_atomic_load(plp->dev_busy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) & ~own_bit) != 0ull) break;
            if (__hip_atomic_load(plp->dev_active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) -
                    __hip_atomic_load(&doneq[kDqOwn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0)
              break;
            const int head = __hip_atomic_load(&doneq[kDqHead], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int tail = __hip_atomic_load(&doneq[kDqTail], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (head >= tail) break;
            if (atomicCAS(&doneq[kDqHead], head, head + 1) != head) continue;
            int v;
            while ((v = __hip_atomic_load(&doneq[kDqRing + head], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) __builtin_amdgcn_s_sleep(4);
            item   = v - 1;
            resume = 2;
            break;
          }
        }
      }
    }
    item   = __builtin_amdgcn_readfirstlane(item);
    resume = __builtin_amdgcn_readfirstlane(resume);
    const int lean_waves_item = lean_waves;
    if (item < 0) break;
    if (resume) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the suspending block's stores (other CU / XCD)
    const bool helper = LAT && resume == 2;  // (this item: polish and report, as a polisher would)
    if (dq_mode == 2 || helper) resume = 0;  // (a polisher takes the item up like the finish launch does: from its workspace header)
    wave_sync();
    // Pruned plan: the declaration "these stored entries of A are zero" is checked for the item.  An item that
    // violates it is solved on the WHOLE pattern (plan plf, same elimination order) right here, in a workspace slot
    // of the small fallback pool (whole-pattern items need more room); it runs to completion, never time-sliced.
    const SparsePlanDev *use = plp;
    double *wsb              = gws;
    size_t wsd = ws_doubles, slot = (queue == nullptr) ? (size_t)blockIdx.x : (size_t)item;
    int fbslot = -1;
    if (plf != nullptr && !resume && ph0 <= PH_SETUP && !sp_guard_ok(*plp, gAx + (size_t)item * (size_t)uni(plp->nnzA_io), lane)) {
      if (lane == 0) {
        for (int probe = blockIdx.x % nfb;; probe = (probe + 1) % nfb) {  // nfb = slots the pool really has
          if (atomicCAS(&fbflags[probe], 0, 1) == 0) { fbslot = probe; break; }
          __builtin_amdgcn_s_sleep(8);
        }
      }
      fbslot = __builtin_amdgcn_readfirstlane(fbslot);
      use = plf; wsb = gwsf; wsd = wsf_doubles; slot = (size_t)fbslot;
    }
    // Waves busy with an item right now (all launches on this device: SparsePlanDev::dev_active).  While there are many, the launch is
    // HBM-bound and the sweeps skip the padding of the factor stream (masked loads: fewer bytes, a few more
    // instructions); when only stragglers are left, latency is what counts and they switch to plain loads.  A
    // heuristic only: both forms compute the same thing.  (Balanced by every wave when it is done; shared by all
    // launches on purpose -- independent batches on other streams fill the chip just the same.)
    int seen = 0;
    // (own count first, global count second -- and the other way round at the end: the difference never shows a wave of this launch)
    const bool counts_own = dq_mode == 2 || (LAT && ((phases >> 28) & 1) == 1);
    if (counts_own && lane == 0) atomicAdd(&(fbflags + kFbSlots + 2 * batch + 16)[kDqOwn], 1);
    int32_t *const active = plp->dev_active;
    if (lane == 0) seen = atomicAdd(active, 1);
    const bool lean = (nwaves > lean_waves_item) || __builtin_amdgcn_readfirstlane(seen) >= lean_waves_item;
    // (phased launches: an item of the fallback pool runs all phases at once, in the setup launch)
    const int st = sp_solve_item<LAT, TRACE, LAT || POLISHER>(*use, kp, gPx, gq, gAx, gl, gu, gwx, gwy, gx, gy, gobj, giter, gcode, wsb, wsd, lean_waves_item,
                                 lean, (size_t)item, slot, t, lane, resume != 0, fbslot >= 0 ? nullptr : queue, nfresh, slice,
                                 /*allow_reuse: the item's own slot of the main workspace*/ fbslot < 0 && slot == (size_t)item,
                                 fbslot >= 0 ? PH_EVERYTHING : (helper ? (PH_FINISH | (PH_FINISH << 4)) : phases), keys ? keys + item : nullptr, trace,
                                 trace_cap, phase_us);
    if (keys != nullptr && st != SP_PAUSED && lane == 0) keys[item] = -1.0f;
    if (fbslot >= 0 && phases != PH_EVERYTHING && lane == 0)  // tell the later launches (the item's own slot is otherwise unused)
      carve_ws(gws + (size_t)item * ws_doubles, uni(plp->n), uni(plp->m), uni(plp->nnzL), uni(plp->funits), uni(plp->bunits)).hdr[kHdrComplete] = 1.0;
    if ((dq_mode == 2 || helper) && lane == 0)  // polished and reported: the finish launch skips it
      carve_ws(gws + (size_t)item * ws_doubles, uni(plp->n), uni(plp->m), uni(plp->nnzL), uni(plp->funits), uni(plp->bunits)).hdr[kHdrComplete] = 1.0;
    wave_sync();
    if constexpr (LAT) {
      // the item's loop has ended in a loop launch (phases ADMM .. ADMM of a launch with auxiliary memory): hand it to the polishers
      // (the ring is there and zeroed whether or not polishers run)
      if (st == SP_DONE && !helper && (phases & 0xFF) == (PH_ADMM | (PH_ADMM << 4)) && ((phases >> 28) & 1) == 1 && lane == 0) {
        int32_t *const doneq = fbflags + kFbSlots + 2 * batch + 16;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int tl = atomicAdd(&doneq[kDqTail], 1);
        __hip_atomic_store(&doneq[kDqRing + tl], item + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (lane == 0) {
      atomicSub(active, 1);
      if (counts_own) atomicSub(&(fbflags + kFbSlots + 2 * batch + 16)[kDqOwn], 1);
      if (fbslot >= 0) __hip_atomic_store(&fbflags[fbslot], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      if (st == SP_SUSPENDED) {
        // publish the item's state (plain and non-temporal stores of this wave) before its id enters the ring
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int tl = atomicAdd(&queue[kQTail], 1);
        int32_t *e   = &queue[kQRing + (unsigned)tl % (unsigned)batch];
        while (__hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) __builtin_amdgcn_s_sleep(4);
        __hip_atomic_store(e, item + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    wave_sync();
    if (queue == nullptr && dq_mode != 2) break;
  }
}

// ORDER OF THE SURVIVORS (launches in predicted order): items with a score (>= 0 or NaN) sorted by descending score
// into order2[0 .. count[0] - 1].  Counting sort on the leading 12 bits of the float (16 bins per octave -- the score
// predicts the iterations left to a few percent at best); one workgroup, the batch is small next to the solves.
// count[1] = WAVES of the second launch.  The launch is shortest when the longest item, iterating all the time, ends
// together with the rest of the work: waves = (sum of the survivors' remaining iterations) / (those of the longest).
// Before the active set settles ADMM's residuals fall like 1 / iteration, so the remaining iterations go like the score
// itself (headline batch: 25, 30, 50, 77, 127, 327, 968 iterations on average for scores in [2, 4), [4, 8), ... --
// sum / max of the scores 493, of the true counts 443): waves = sum_k score_k / score_max, clamped to [g_lo, g_hi] --
// g_lo: the items whose factors fit the Infinity Cache (fewer waves do not saturate it), g_hi: what the chip holds.
// GATE in front of the polishers (round 6): one wave on the polishers' stream that waits until the loop launch's waves are on the
// chip.  A LAT wave needs a SIMD's whole register file; polishers that are placed FIRST -- two fit a SIMD -- take SIMDs away from
// the loop launch for its whole length (round 5: 512 polishers, 48.8 instead of 42.8 ms).  Behind the gate they only ever get
// what the LAT waves leave: the fourth SIMD of every compute unit at once, and the SIMDs of LAT waves that have run out of items.
// Bounded (a loop launch that never starts must not hang the stream: the polishers then simply start early).
__global__ void sp_gate_kernel(const int32_t *__restrict__ started)
{
  for (int spin = 0; spin < 200000; ++spin) {  // (~0.1 s)
    if (__hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
    __builtin_amdgcn_s_sleep(32);
  }
}

constexpr int kRankBins = 4096, kRankThreads = 1024;
__global__ __launch_bounds__(kRankThreads) void sp_rank_kernel(const float *__restrict__ keys, const int batch,
                                                               int32_t *__restrict__ order2, int32_t *__restrict__ count,
                                                               const int g_lo, const int g_hi, const int lat_lo, const int lat_hi,
                                                               const int lat_max, const int stream_bit, const int pol_lo, const int pol_hi)
{
  __shared__ int hist[kRankBins];
  __shared__ int psum[kRankThreads];
  const int tid = threadIdx.x;
  auto bin_of = [](float k) { return (kRankBins - 1) - (int)(__float_as_uint(k) >> 19); };  // (sign bit clear: < 4096)
  for (int b = tid; b < kRankBins; b += kRankThreads) hist[b] = 0;
  __syncthreads();
  for (int i = tid; i < batch; i += kRankThreads) {
    const float k = keys[i];
    if (!(k < 0.0f)) atomicAdd(&hist[bin_of(fabsf(k))], 1);
  }
  __syncthreads();
  constexpr int per = kRankBins / kRankThreads;
  int mine[per], tot = 0;
  for (int e = 0; e < per; ++e) { mine[e] = hist[tid * per + e]; tot += mine[e]; }
  psum[tid] = tot;
  __syncthreads();
  for (int d = 1; d < kRankThreads; d <<= 1) {  // inclusive scan of the per-thread totals
    const int v = (tid >= d) ? psum[tid - d] : 0;
    __syncthreads();
    psum[tid] += v;
    __syncthreads();
  }
  int start = psum[tid] - tot;
  for (int e = 0; e < per; ++e) { hist[tid * per + e] = start; start += mine[e]; }
  const int total = psum[kRankThreads - 1];
  __syncthreads();
  {  // score of a bin, from its exponent and leading mantissa bits: log2 = (bits >> 19) / 16 - 127; scores beyond 2^40
     // (and infinities / NaNs) count as 2^40, scores below 1 as 1
    auto lg = [](int b) { return exp2f(fminf(40.0f, fmaxf(0.0f, (float)((kRankBins - 1) - b) * (1.0f / 16.0f) - 127.0f))); };
    float part = 0.0f;
    int first  = kRankBins;
    for (int e = 0; e < per; ++e) {
      part += (float)mine[e] * lg(tid * per + e);
      if (mine[e] > 0 && first == kRankBins) first = tid * per + e;
    }
    __shared__ float fsum[kRankThreads];
    fsum[tid] = part;
    psum[tid] = first;
    __syncthreads();
    for (int d = kRankThreads / 2; d > 0; d >>= 1) {
      if (tid < d) {
        fsum[tid] += fsum[tid + d];
        psum[tid] = min(psum[tid], psum[tid + d]);
      }
      __syncthreads();
    }
    if (tid == 0) {
      const float want = (psum[0] < kRankBins) ? fsum[0] / lg(psum[0]) : 0.0f;
      // count[0..2]: survivors, waves, form of the second launch (1 = the LAT kernel on at most lat_hi waves, chosen when
      // that is enough for the work; 0 = the standard kernel); count[4..6]: the same for a launch on the whole chip
      const int w64 = ((int)fminf(want, 1e9f) + 32) / 64 * 64;
      const int lat = lat_hi > 0 && w64 <= lat_max;  // lat_max: the wanted waves up to which the LAT form is chosen
      count[0] = total;
      count[1] = lat ? max(lat_lo, min(lat_hi, w64)) : max(g_lo, min(g_hi, w64));
      count[2] = lat;
      count[4] = total;
      count[5] = g_hi;
      count[6] = lat;
      count[8] = stream_bit;  // (the caller stream's bit in SparsePlanDev::dev_busy: the helpers look past their own)
      // POLISHERS at work next to the loop launch: pol_lo (one per compute unit) or pol_hi (two).  Polish under load costs an
      // item ~1.7 ms of wave time whatever it iterated; the polishers' traffic slows the LAT waves' iterations down.  With a
      // cold start's ~90 iterations per item the iterations are the bulk and a second polisher per unit costs more than it
      // brings (headline 38.6 -> 45.2 ms); with the ~38 of a warm-started tick polish is two thirds of the launch's wave
      // time and it pays (25.5 -> 24.7 ms).  What tells the two apart at this point is the mean score (residual over tolerance
      // at the first check): ~600 for the cold headline batch, ~20 for warm ticks of the same swarm.
      // ... and only when the LAT waves have many rounds of items in front of them: with a few items per wave they run out soon
      // and polish themselves, a second polisher per unit only adds its traffic (warm ticks, one / two per unit: 2 048 agents
      // 8.5 / 9.5 ms, 4 096: 13.5 / 14.5, 8 192: 25.5 / 24.7, 16 384: 49.3 / 47.6).
      count[10] = (total >= 8 * max(1, lat_hi) && fsum[0] / (float)total < 100.0f) ? pol_hi : pol_lo;
    }
  }
  __syncthreads();
  for (int i = tid; i < batch; i += kRankThreads) {
    const float k = keys[i];
    if (!(k < 0.0f)) order2[atomicAdd(&hist[bin_of(fabsf(k))], 1)] = i;
  }
}

// LAUNCH BOOKKEEPING, per device, kept by the HOST (created on first use, never freed):
//   * what a launch needs to know about the device and the kernels, asked ONCE (compute units, resident blocks per LDS size,
//     the LDS opt-in of the LAT form) -- a steady-state solve makes no HIP call besides launches, memsets and event record / wait;
//   * per caller stream ("slot", at most 64; further streams share slots): its bit in the busy word, the polishers' stream that
//     goes with it (lowest priority -- when both have workgroups to place, the loop launch's go first --, not synchronising with
//     the null stream), the two events of the hand-off between the two, and the event that marks the end of the stream's most
//     recent solve;
//   * the two words the kernels read (SparsePlanDev::dev_active / dev_busy).  The busy word lives in mapped host memory: the
//     host sets a stream's bit when it ENQUEUES a solve there and clears it when it next looks (at an enqueue) and finds the
//     stream's last solve finished.  A stale bit costs the helpers of other streams' launches, never a result.
struct SparseDeviceBook {
  struct Slot {
    int bit = 0;
    hipStream_t polish = nullptr;  // nullptr: could not be created -- no polishers for this stream
    hipEvent_t ev1 = nullptr, ev2 = nullptr, last = nullptr;
    int inflight = 0;    // solves enqueued whose end `last` has not been seen yet
    int enqueueing = 0;  // ... of which the enqueue is still going on (another thread): `last` is not recorded yet -- busy, no query
  };
  std::mutex mu;
  int dev = -1, cus = 0;
  int lat_opt_in = -1;  // hipFuncSetAttribute(LAT form, 80 KB of dynamic LDS): -1 not asked, 0 refused, 1 granted
  std::map<size_t, int> per_cu_std, per_cu_lat;  // dynamic LDS bytes -> resident blocks per CU
  int32_t *active = nullptr;                     // device memory
  unsigned long long *busy_host = nullptr;       // mapped host memory (or nullptr: `busy_dev` points at a device word that stays 0)
  const unsigned long long *busy_dev = nullptr;
  std::map<hipStream_t, Slot> slots;
  int next_bit = 0;
};

static SparseDeviceBook *sparse_book()
{
  static std::mutex mu;
  static std::map<int, SparseDeviceBook *> books;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  std::lock_guard<std::mutex> lk(mu);
  const auto it = books.find(dev);
  if (it != books.end()) return it->second;
  auto *b = new SparseDeviceBook();
  b->dev  = dev;
  int32_t *words = nullptr;  // [0]: busy waves; [32]: the always-zero stand-in of the busy word
  if (hipDeviceGetAttribute(&b->cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&words), 64 * sizeof(int32_t)) != hipSuccess ||
      hipMemset(words, 0, 64 * sizeof(int32_t)) != hipSuccess) {
    (void)hipGetLastError();
    if (words) (void)hipFree(words);
    delete b;
    return nullptr;
  }
  b->active   = words;
  b->busy_dev = reinterpret_cast<const unsigned long long *>(words + 32);
  void *hostw = nullptr, *devw = nullptr;
  if (hipHostMalloc(&hostw, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&devw, hostw, 0) == hipSuccess) {
    b->busy_host  = static_cast<unsigned long long *>(hostw);
    *b->busy_host = 0ull;
    b->busy_dev   = static_cast<const unsigned long long *>(devw);
  } else {
    (void)hipGetLastError();
    if (hostw) (void)hipHostFree(hostw);
  }
  books[dev] = b;
  return b;
}

hipError_t sparse_device_words(int32_t **active, const unsigned long long **busy)
{
  SparseDeviceBook *b = sparse_book();
  if (b == nullptr) return hipErrorOutOfMemory;
  *active = b->active;
  *busy   = b->busy_dev;
  return hipSuccess;
}

// the slot of a caller stream (created on first use; book.mu held)
static SparseDeviceBook::Slot &sparse_slot(SparseDeviceBook &b, hipStream_t stream)
{
  const auto it = b.slots.find(stream);
  if (it != b.slots.end()) return it->second;
  if (b.slots.size() >= 64) {  // more caller streams than bits: share a slot (and its polishers' stream: still correct, see the kernel)
    auto sh = b.slots.begin();
    std::advance(sh, (size_t)(reinterpret_cast<uintptr_t>(stream) >> 6) % b.slots.size());
    return sh->second;
  }
  SparseDeviceBook::Slot sl;
  sl.bit = b.next_bit++ & 63;
  int least = 0, greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess ||
      hipStreamCreateWithPriority(&sl.polish, hipStreamNonBlocking, least) != hipSuccess) {
    (void)hipGetLastError();
    sl.polish = nullptr;
  }
  if (hipEventCreateWithFlags(&sl.ev1, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&sl.ev2, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&sl.last, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    if (sl.polish) (void)hipStreamDestroy(sl.polish);
    sl.polish = nullptr;  // (without the events: no polishers and no busy bit for this stream)
    sl.ev1 = sl.ev2 = sl.last = nullptr;
  }
  return b.slots.emplace(stream, sl).first->second;
}

// A solve is about to be enqueued on `stream`: look at the other streams' last solves (clearing the bits of those that have
// finished), mark this one busy.  Returns the stream's slot (a copy) and whether ANOTHER stream has work in flight.
static SparseDeviceBook::Slot sparse_enqueue_begin(SparseDeviceBook &b, hipStream_t stream, bool *others_busy)
{
  std::lock_guard<std::mutex> lk(b.mu);
  SparseDeviceBook::Slot &mine = sparse_slot(b, stream);
  bool others = false;
  for (auto &kv : b.slots) {
    SparseDeviceBook::Slot &sl = kv.second;
    if (&sl == &mine || sl.inflight == 0) continue;
    if (sl.enqueueing > 0) {  // (its event still marks the end of an OLDER solve)
      others = true;
      continue;
    }
    const hipError_t q = hipEventQuery(sl.last);
    if (q == hipErrorNotReady) {
      (void)hipGetLastError();
      others = true;
      continue;
    }
    (void)hipGetLastError();  // finished (or the event is in error: nothing to wait for either way)
    sl.inflight = 0;
    if (b.busy_host) __atomic_fetch_and(b.busy_host, ~(1ull << sl.bit), __ATOMIC_RELEASE);
  }
  if (mine.last != nullptr) {
    mine.inflight = 1;
    ++mine.enqueueing;
    if (b.busy_host) __atomic_fetch_or(b.busy_host, 1ull << mine.bit, __ATOMIC_RELEASE);
  }
  *others_busy = others;
  return mine;
}
// ... and its last launch has been enqueued (or the enqueue failed half-way: the stream's work up to here still ends there)
static void sparse_enqueue_end(SparseDeviceBook &b, const SparseDeviceBook::Slot &sl, hipStream_t stream)
{
  if (sl.last == nullptr) return;
  if (hipEventRecord(sl.last, stream) != hipSuccess) (void)hipGetLastError();
  std::lock_guard<std::mutex> lk(b.mu);
  for (auto &kv : b.slots)
    if (kv.second.last == sl.last && kv.second.enqueueing > 0) --kv.second.enqueueing;
}

// the LDS opt-in of the LAT form (80 KB of dynamic LDS), asked once per device
static bool sparse_lat_opt_in(SparseDeviceBook &b)
{
  std::lock_guard<std::mutex> lk(b.mu);
  if (b.lat_opt_in < 0) {
    b.lat_opt_in = hipFuncSetAttribute(reinterpret_cast<const void *>(qp_sparse_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess;
    if (!b.lat_opt_in) (void)hipGetLastError();
  }
  return b.lat_opt_in == 1;
}

// blocks of qp_sparse_kernel (standard / LAT form) the device holds at once with `lds` bytes of dynamic LDS each
static int sparse_resident_blocks(SparseDeviceBook &b, size_t lds, bool lat = false)
{
  std::lock_guard<std::mutex> lk(b.mu);
  auto &cache = lat ? b.per_cu_lat : b.per_cu_std;
  const auto it = cache.find(lds);
  if (it != cache.end()) return it->second * b.cus;
  int per_cu = 0;
  const hipError_t e = lat ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, qp_sparse_kernel<true>, kWave, lds)
                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, qp_sparse_kernel<false>, kWave, lds);
  if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
  cache[lds] = per_cu;
  return per_cu * b.cus;
}

// auxiliary memory of a launch: [queue: kQRing + batch][flags of the fallback pool: kFbSlots] (zeroed per kernel) and,
// for launches in predicted order, [scores: batch floats][order of the survivors: batch][their count: 16]
static size_t sparse_aux_queue_ints(int64_t batch) { return (size_t)batch + kQRing + kFbSlots; }
size_t qp_sparse_aux_bytes(int64_t batch) { return (sparse_aux_queue_ints(batch) + 2 * (size_t)batch + 16 + kDqRing + (size_t)batch) * sizeof(int32_t); }
int qp_sparse_fallback_slots() { return kFbSlots; }

hipError_t qp_sparse_launch(const SparsePlanDev &pl, const DenseKernelParams &kp, int64_t batch, const double *Px,
                            const double *q, const double *Ax, const double *l, const double *u, const double *wx,
                            const double *wy, double *x, double *y, double *obj, uint32_t *iter, int32_t *code,
                            double *workspace, hipStream_t stream, const int32_t *order, int32_t *aux,
                            const SparsePlanDev *fallback, double *fallback_ws, double *trace, int trace_cap, double *phase_us)
{
  if (pl.Aorig != nullptr && (fallback == nullptr || fallback_ws == nullptr || aux == nullptr)) return hipErrorInvalidValue;
  SparseDeviceBook *const book = sparse_book();  // (exists: the plan's device copy was uploaded with its words)
  if (book == nullptr) return hipErrorInvalidDevice;
  const bool pruned = pl.Aorig != nullptr;
  const size_t lds = (size_t)std::max(pl.lds_doubles, pruned ? fallback->lds_doubles : 0) * sizeof(double);
  const size_t wsd = qp_sparse_ws_doubles(pl);
  // below this many busy waves the sweeps use plain loads (see the kernel); SFB_SP_LEAN_WAVES overrides (tuning)
  const char *lw        = sfb::knob("SFB_SP_LEAN_WAVES");
  const int lean_waves  = lw ? atoi(lw) : 512;
  // Time slicing (see the kernel): only when the batch does not fit the chip at once.  SFB_SP_SLICE = iterations
  // per slice (0 = off: one block per item, the hardware dispatcher is the queue).
  const char *sl   = sfb::knob("SFB_SP_SLICE");
  const int slice  = sl ? atoi(sl) : 50;
  unsigned grid    = (unsigned)batch;
  int32_t *qarg    = nullptr;
  bool sliced      = false;
  if (aux != nullptr && slice > 0) {
    int resident = sparse_resident_blocks(*book, lds);
    if (const char *g = sfb::knob("SFB_SP_GRID"); g && atoi(g) > 0) resident = std::min(resident, atoi(g));  // tests: force slicing
    // (round 6: a batch beyond what the LAT form holds at once -- three per CU -- goes through the queue even when the standard form
    //  would hold it at once: the launch in predicted order with its LAT loop launch is the faster way through the iterations.
    //  Host entry, headline model: 1 024 QPs 31.7 -> 28.2 ms, 2 048: 41.3 -> 34.5, 3 072: 52.6 -> 40.9; up to 768 the whole launch
    //  runs in the LAT form, see below)
    const bool lat_ok  = pl.n <= kLatRegRows * kWave && pl.m <= kLatRegRows * kWave && lat_lds_doubles(pl.n, pl.m) * sizeof(double) <= 80 * 1024;
    const int64_t hold = lat_ok ? std::min<int64_t>(resident, 3 * (int64_t)book->cus) : resident;
    if (resident > 0 && batch > hold) {
      grid   = (unsigned)std::min<int64_t>(resident, batch);
      qarg   = aux;
      sliced = true;
    }
  }
  // PHASED LAUNCH (time-sliced launches only, SFB_SP_PHASED=1): three kernels on the stream instead of one -- setup
  // (scaling, factorisation, initial iterate), the ADMM loop, polish + report -- each with its own grid; the item's
  // state travels through its workspace header, exactly like a suspended item of a time-sliced launch, and the results
  // are bit-identical (tests/test_mpc_gpu.py).  The idea: setup and polish are latency-bound per item and want every
  // wave the chip holds, while the ADMM loop streams the factor twice per iteration and is fastest with only as many
  // items in flight as the 256 MB Infinity Cache (MALL) holds factor copies of (loop alone, headline plan: 21.3
  // item-iterations/us with 768 waves and plain loads against 16.8 with 3 072 waves and non-temporal loads).
  // Measured on the headline batch (scripts/r3/phased_sweep.py): 78 ms against 76 ms for the single kernel -- what the
  // ADMM phase gains, the lost overlap of setup / polish (7.6 + 10 ms on their own) with it costs again, and the tail of
  // long-running items is the same.  So the phased launch is OFF by default; it stays as the instrument that separates
  // the three phases for the profiler (profiles/r3_mpc_phases).
  const char *ph    = sfb::knob("SFB_SP_PHASED");
  const bool phased = sliced && ph && atoi(ph) == 1;
  struct DoneQ { int32_t *q = nullptr; int mode = 0; hipStream_t on = nullptr; };  // ring of finished items (see the kernel); `on`: the polishers' stream
  auto launch = [&](unsigned g, int32_t *qa, int lw, int phases, const int32_t *ord = nullptr, uint32_t slc = 0, float *keys = nullptr,
                    const int32_t *nfresh = nullptr, int mode_sel = -1, bool lat = false, size_t lds_lat = 0, DoneQ dq = DoneQ{}) -> hipError_t {
    if ((qa != nullptr || pruned) && dq.mode != 2) {  // (the polishers take no tickets and run next to a launch that does)
      hipError_t e = hipMemsetAsync(aux, 0, sparse_aux_queue_ints(batch) * sizeof(int32_t), stream);
      if (e != hipSuccess) return e;
    }
    auto *kern = (trace || phase_us) ? qp_sparse_kernel<false, true>
                                     : (dq.mode == 2 ? qp_sparse_kernel<false, false, true> : (lat ? qp_sparse_kernel<true> : qp_sparse_kernel<false>));
    hipLaunchKernelGGL(kern, dim3(g), dim3(kWave), lat ? lds_lat : lds, dq.mode == 2 ? dq.on : stream, pl.self, kp, Px, q, Ax, l, u, wx, wy, x, y,
                       obj, iter, code, workspace, wsd, lw, ord, qa, (int)batch, slc ? slc : (uint32_t)std::max(1, slice),
                       pruned ? fallback->self : nullptr, fallback_ws, pruned ? qp_sparse_ws_doubles(*fallback) : 0,
                       aux ? aux + batch + kQRing : nullptr, phases, (int)std::min<int64_t>(batch, kFbSlots), keys, nfresh, mode_sel,
                       trace, trace_cap, phase_us);
    return hipGetLastError();
  };
  // TRACE (sfb_sparse_qp_solve_batch_trace): one block per item whatever the batch size -- the row count of an item's table
  // lives in its wave -- through the TRACE instance of the kernel; same arithmetic, same results.
  if (trace != nullptr || phase_us != nullptr) return launch((unsigned)batch, nullptr, lean_waves, PH_EVERYTHING, order);
  // LAUNCH IN PREDICTED ORDER (time-sliced launches, default; SFB_SP_PREDICT=0 turns it off).  ADMM iteration counts are
  // heavy-tailed (headline batch: mean 88, p99 627, max 1 152) and a wave alone needs 22-27 us per iteration, so the
  // longest items set the time of a launch unless they START first -- and what identifies them is cheap: the ratio of
  // the primal (or dual) residual to its tolerance at the first stopping check after ~26 iterations orders the items
  // almost like their final iteration counts (rank correlation 0.97 on the headline batch; the ten longest items are
  // among the first fifteen).  So: (1) one launch takes every item through setup and its first `pause` iterations --
  // items that are done by then are polished and reported right there, the others leave their score and wait in their
  // workspace (same state as a suspended item); (2) a one-block counting sort orders the survivors by score; (3) a second
  // launch runs them longest-first, first come first served, on a grid small enough that a wave iterates at nearly the
  // speed of a lone wave (its factor stays in the 256 MB Infinity Cache).  The order is a schedule only: the results are
  // bit-identical (tests/test_mpc_gpu.py).
  // WHERE the first launch pauses (round 5): at the FIRST stopping check (iteration 1; debug knob SFB_SP_PAUSE, rounds 3-4: 27).
  // The iterations of the first launch stream their factors from HBM with every wave of the chip (18 item-iterations per
  // microsecond), the second launch's from the Infinity Cache (21); while setup and polish cost 18 of a step's 53 ms the
  // better-informed order after 26 iterations paid for the difference, with the unit engine of the factorisation it no
  // longer does: 8 192 agents 51.3 -> 48.3 ms, 4 096: 36.8 -> 30.7, 16 384: 94.5 -> 94.0, a warm swarm tick 35 -> 30 ms.
  // The residual-to-tolerance ratio after one iteration still puts the long runners first (the second launch now holds
  // the whole batch, its length is its total work rather than its longest item).
  const char *pr       = sfb::knob("SFB_SP_PREDICT");
  const char *pa       = sfb::knob("SFB_SP_PAUSE");
  const unsigned pause = pa ? (unsigned)std::max(0, atoi(pa)) : 2u;
  const bool predicted = sliced && !phased && !(pr && atoi(pr) == 0) && pause > 0 && kp.stop_check_iter >= 2 && kp.max_iter > 2 * pause;
  if (predicted) {
    int32_t *xtra   = aux + sparse_aux_queue_ints(batch);
    float *keys     = reinterpret_cast<float *>(xtra);
    int32_t *order2 = xtra + batch, *count = xtra + 2 * batch;
    // the stream's slot of the device's bookkeeping; from here on the stream counts as busy, until the event recorded when this
    // block is left (the last launch is enqueued, or the enqueue failed half-way) is seen finished
    bool others_busy = false;
    const SparseDeviceBook::Slot slot = sparse_enqueue_begin(*book, stream, &others_busy);
    struct EnqueueEnd {
      SparseDeviceBook &bk;
      const SparseDeviceBook::Slot &sl;
      hipStream_t st;
      ~EnqueueEnd() { sparse_enqueue_end(bk, sl, st); }
    } enqueue_end{*book, slot, stream};
    hipError_t e = launch(grid, qarg, lean_waves, phases_pack(PH_SETUP, PH_FINISH, pause), order, 0, keys);
    if (e != hipSuccess) return e;
    // Waves of the second launch: at least as many items as keep their two schedule-ordered factor copies in ~70 % of the
    // 256 MB Infinity Cache (headline plan: 640 -- measured 512: 70.5 ms, 576: 66.7, 640: 64.0, 768: 65.2, 1 024: 72.1), more
    // when the survivors' work is large against the longest item (the rank kernel decides: 16 384 agents 896, 32 768 agents
    // 1 792); the grid is the chip's, the blocks beyond the chosen number leave at once.  Cacheable loads while the active
    // waves' factors fit the Infinity Cache, non-temporal masked ones beyond (as in the single launch).
    const double stream_bytes = (double)(pl.funits + pl.bunits) * 128.0 * sizeof(double), mall = 256.0 * 1024.0 * 1024.0;
    unsigned g_lo = std::min(grid, ((unsigned)std::max(256.0, 0.70 * mall / stream_bytes) + 32u) / 64u * 64u), g_hi = grid;
    // LAT form of the second launch (see sp_solve_item): the loop's vectors in LDS, two waves per CU -- when the vectors fit
    // (the rank kernel sizes the launch); polish and
    // report of the survivors then follow as a launch of their own on the whole chip (they are latency-bound and want every
    // wave; inside the few waves of the loop they took 40 % of the wave time).  Otherwise: the standard kernel, loop and
    // polish in one launch.  Both are enqueued, the blocks of the form that was not chosen leave at once.
    const size_t lds_lat = std::max(lds, lat_lds_doubles(pl.n, pl.m) * sizeof(double));
    int lat_hi = 0, lat_lo = 0;
    const bool lat_fits = lds_lat <= 80 * 1024 && pl.n <= kLatRegRows * kWave && pl.m <= kLatRegRows * kWave;  // (the iterate in registers)
    if (const char *lt = sfb::knob("SFB_SP_LAT"); !(lt && atoi(lt) == 0) && lat_fits) {
      // (asked once per device -- the attribute belongs to the device, the shards of a *_multi call run on several.  A runtime
      // that refuses the opt-in leaves lat_hi = 0: the standard form runs.)
      if (sparse_lat_opt_in(*book)) lat_hi = std::min<int>(sparse_resident_blocks(*book, lds_lat, true), (int)grid);
      // (measurements: 448 waves 56.3 ms, 512: 51.8, 640: 47.0, 768: 44.0 -- the launch is bound by waves x iteration rate, not by
      //  bytes, up to three waves per CU; a FOURTH, bought by leaving part of 1 / D in the workspace, loses again: 896 waves 45.6,
      //  1 024: 46.3, scripts/r5/experiments/lat_four_waves_partial_dinv.diff)
      if (const char *lw = sfb::knob("SFB_SP_LAT_WAVES"); lw && atoi(lw) > 0) lat_hi = std::min(lat_hi, atoi(lw));
      lat_lo = std::min(lat_hi, 448);
    }
    // ... for any amount of work: two LAT waves per CU (loop vectors in LDS, factors from the Infinity Cache) also move more
    // item-iterations per microsecond (21) than the standard form on the whole chip (17-18.5): 12 288 agents 90.7 -> 80.9 ms,
    // 16 384: 120.6 -> 106.4, 32 768: 218.6 -> 205.3 (scripts/r3/batch_sweep.sh)
    const int lat_max = 0x7FFFFFFF;
    // polishers next to the loop launch (see the rank kernel and below): SFB_SP_POLISHERS=N forces N, 0 none
    int pol_lo = book->cus, pol_hi = 2 * book->cus;
    if (const char *pk = sfb::knob("SFB_SP_POLISHERS"); pk && atoi(pk) > 0) pol_lo = pol_hi = atoi(pk);
    pol_lo = (int)std::min<int64_t>(pol_lo, batch);
    pol_hi = (int)std::min<int64_t>(pol_hi, batch);
    hipLaunchKernelGGL(sp_rank_kernel, dim3(1), dim3(kRankThreads), 0, stream, keys, (int)batch, order2, count, (int)g_lo, (int)g_hi,
                       lat_lo, lat_hi, lat_max, slot.bit, pol_lo, pol_hi);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    const uint32_t slice3 = 0x40000000u;
    if (lat_hi > 0) {
      // POLISHERS (round 5).  Three LAT waves per CU leave one SIMD and 25 KB of LDS idle for the 30 ms of the loop launch, while
      // polish and report of the whole batch wait behind it (7.6 ms on the whole chip).  One standard-form wave per CU, launched
      // on a low-priority stream that belongs to the CALLER's stream (round 6: polishers spin until their loop launch is over,
      // so two callers must not queue theirs on one stream), takes the items whose loop has ended from a ring the LAT waves
      // push them into, polishes and reports them; the finish launch does what is left when the loop launch ends (the
      // last items' polish) and skips the rest.  Which wave polishes an item changes nothing in it.
      DoneQ prod{}, cons{};
      // 0: no polishers (measurements, tests); N > 0: N of them; default: one or two per CU, the rank kernel's choice (pol_lo / pol_hi)
      const char *po  = sfb::knob("SFB_SP_POLISHERS");
      const int npolish = pol_hi;
      if (hipStream_t hs = (po && atoi(po) == 0) ? nullptr : slot.polish; hs != nullptr) {
        int32_t *dq = count + 16;  // (== fbflags + kFbSlots + 2 batch + 16: where the kernel looks for it)
        if (hipMemsetAsync(dq, 0, (size_t)(kDqRing + batch) * sizeof(int32_t), stream) == hipSuccess && hipEventRecord(slot.ev1, stream) == hipSuccess) {
          // helpers (3: LAT waves without an item help the polishers, see the kernel) unless another stream has work in flight
          // right now -- those that do start look at the device's busy word before every item they take; SFB_SP_LAT_HELP=0:
          // never (measurements)
          const char *hp = sfb::knob("SFB_SP_LAT_HELP");
          prod = DoneQ{dq, ((hp && atoi(hp) == 0) || others_busy) ? 1 : 3, nullptr};
          cons = DoneQ{dq, 2, hs};
        }
      }
      (void)hipGetLastError();
      e = launch((unsigned)lat_hi, qarg, 0x7FFFFFFF, phases_pack(PH_ADMM, PH_ADMM, 0, prod.mode), order2, slice3, nullptr, count, 1, true, lds_lat, prod);
      if (e == hipSuccess && cons.mode == 2) {
        // (the end of the loop launch, for the polishers: whatever happens from here on, this flag is set before the stream goes on)
        e = hipMemsetAsync(cons.q + kDqOver, 0xFF, sizeof(int32_t), stream);
        bool polishers_out = false;
        if (e == hipSuccess) e = hipStreamWaitEvent(cons.on, slot.ev1, 0);
        if (e == hipSuccess) {
          hipLaunchKernelGGL(sp_gate_kernel, dim3(1), dim3(1), 0, cons.on, cons.q + kDqStarted);
          e = hipGetLastError();
        }
        if (e == hipSuccess) {
          e = launch((unsigned)npolish, qarg, 0, phases_pack(PH_FINISH, PH_FINISH, 0, 2), nullptr, slice3, nullptr, nullptr, -1, false, 0, cons);
          polishers_out = e == hipSuccess;
        }
        if (e == hipSuccess) e = hipEventRecord(slot.ev2, cons.on);
        if (e == hipSuccess) e = hipStreamWaitEvent(stream, slot.ev2, 0);
        if (e != hipSuccess && polishers_out) {
          // the polishers are out and the stream could not be made to wait for them: they may still write the caller's
          // workspace and outputs when this call returns its error -- wait here, on the host (the flag above ends their spin
          // once the stream gets there)
          (void)hipStreamSynchronize(stream);
          (void)hipStreamSynchronize(cons.on);
        }
      }
      if (e != hipSuccess) return e;
      e = launch(grid, qarg, lean_waves, phases_pack(PH_FINISH, PH_FINISH), order2, slice3, nullptr, count + 4, 1);
      if (e != hipSuccess) return e;
    }
    return launch(g_hi, qarg, (int)std::max(512.0, mall / stream_bytes), phases_pack(PH_ADMM, PH_FINISH), order2, slice3,
                  nullptr, count, lat_hi > 0 ? 0 : -1);
  }
  if (const char *fl = sfb::knob("SFB_SP_FORCE_LAT"); !phased && fl && atoi(fl) == 1 && pl.n <= kLatRegRows * kWave && pl.m <= kLatRegRows * kWave) {  // measurements: the LAT form for a whole launch
    const size_t lds_lat = std::max(lds, lat_lds_doubles(pl.n, pl.m) * sizeof(double));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(qp_sparse_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return launch(grid, qarg, lean_waves, PH_EVERYTHING, order, 0, nullptr, nullptr, -1, true, lds_lat);
  }
  // SMALL BATCHES (round 6): a launch of one block per item that the LAT form holds at once -- up to three per compute unit -- runs
  // in the LAT form as a whole (setup, loop with the iterate in registers and the resident stream head, polish in the wide form):
  // what a single controller or a small swarm waits for is the latency of its items' chains, and the LAT form's iteration is 18.6
  // against 24.6 us.  One QP of the headline model through the host entry 1.89 -> 1.81 ms, 64: 12.2 -> 10.8 ms, 512: 24.2 -> 21.0 ms
  // (scripts/single_agent_latency.py); same results (the form is a schedule: fuzzed as SFB_SP_FORCE_LAT=1).  SFB_SP_LAT=0: never.
  if (!phased && !sliced) {
    const char *lt       = sfb::knob("SFB_SP_LAT");
    const size_t lds_lat = std::max(lds, lat_lds_doubles(pl.n, pl.m) * sizeof(double));
    if (!(lt && atoi(lt) == 0) && batch <= 3 * (int64_t)book->cus && lds_lat <= 80 * 1024 && pl.n <= kLatRegRows * kWave &&
        pl.m <= kLatRegRows * kWave && sparse_lat_opt_in(*book))
      return launch(grid, qarg, 0x7FFFFFFF /* plain loads: the latency form of the sweeps */, PH_EVERYTHING, order, 0, nullptr, nullptr, -1, true, lds_lat);
  }
  if (!phased) return launch(grid, qarg, lean_waves, PH_EVERYTHING, order);
  // grid of the ADMM phase: items whose two schedule-ordered factor copies fit ~70 % of the MALL
  const double stream_bytes = (double)(pl.funits + pl.bunits) * 128.0 * sizeof(double);
  unsigned grid2 = (unsigned)std::max(256.0, 0.70 * 256.0 * 1024.0 * 1024.0 / stream_bytes);
  grid2 = std::min(grid2, grid);
  // plain (cache-allocating) loads in the ADMM phase unless the grid is too large for the MALL anyway
  const int lean_waves2 = grid2 < grid ? 0x7FFFFFFF : lean_waves;
  hipError_t e = launch(grid, qarg, lean_waves, phases_pack(PH_SETUP, PH_SETUP), order);
  if (e != hipSuccess) return e;
  e = launch(grid2, qarg, lean_waves2, phases_pack(PH_ADMM, PH_ADMM), order);
  if (e != hipSuccess) return e;
  return launch(grid, qarg, lean_waves, phases_pack(PH_FINISH, PH_FINISH), order);
}

}  // namespace sfb
