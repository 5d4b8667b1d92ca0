// Dense ADMM QP solver for gfx950, problems with 64 < n+m <= 1024: ONE QP PER WAVEFRONT, the KKT matrix and its
// pivoted LDL' factor in the QP's HBM workspace (they no longer fit a lane-per-row register / LDS layout), vectors in
// LDS, every lane owning rows lane, lane + 64, ...
//
// Replaces, per batch item, smooth::feedback::solve_qp for QuadraticProgram<M,N,double> (reference
// qp_solver.hpp:343-568, :574-644, :673-730, :92-204) with the DENSE branch's Eigen::LDLT<.,Upper> (:259,:428,:462):
// the reference's own ASIF example and test live at these sizes (examples/mpc_asif_vehicle.cpp:105-129: n = 3,
// m = 203; tests/test_asif.cpp:103-131: n = 4, m = 301).
//
// Arithmetic = oracle/qp_oracle.c, operation for operation (bit-identical results): association of every
// element-wise expression, k-ascending fma chains for every dot product, Eigen 3.4's unblocked pivoted LDL'
// (left-looking, pivot = first largest |entry| of the NOT yet updated trailing diagonal, dot-then-subtract, true
// divisions, zero-pivot bookkeeping), triangular solves in the oracle's per-row order -- executed column by column,
// which gives every row the same ascending (forward) / descending (backward) update sequence.  Compiled with
// -ffp-contract=off; fma() only where the oracle spells it.
//
// Parallelism inside a QP: rows over lanes for the dot products of the factorisation (each row a sequential chain),
// entries over lanes for fills and element-wise phases.  The triangular sweeps are k dependent steps each; the
// forward sweep reads a transposed copy of L (column j contiguous), the backward sweep L itself (row j contiguous),
// with the next step's entries in flight while the current one updates the LDS-resident vector.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "knobs.h"
#include "../../include/sfb.h"
#include "qp_dense_kernel.h"
#include "wave_util.h"

namespace sfb {

namespace {

constexpr int kBigMaxK = 1024;
constexpr int64_t kRoomyBatch = 257;  // launches with fewer QPs (at most one per CU) get the large on-chip row pool (see big_row_cap)
constexpr int kRB      = kBigMaxK / kWave;  // row blocks per lane at the largest size

// The sweeps and the factorisation are real (outlined) device functions: a plain `double *` parameter is a GENERIC
// pointer there and every access through it becomes a flat_load / flat_store (slow path into LDS, waits on both
// counters).  Their LDS arguments are therefore re-typed to the LDS address space on entry -> ds_read / ds_write.
using lds_d = __attribute__((address_space(3))) double;
using lds_i = __attribute__((address_space(3))) int;
#define LDS_D(name, arg) lds_d *const name = (lds_d *)(arg)
#define LDS_CD(name, arg) const lds_d *const name = (const lds_d *)(arg)
#define LDS_I(name, arg) lds_i *const name = (lds_i *)(arg)
#define LDS_CI(name, arg) const lds_i *const name = (const lds_i *)(arg)

__device__ __forceinline__ int ubig(const int v) { return __builtin_amdgcn_readfirstlane(v); }

// smallest index among the lanes' candidates (cand >= 0), wave-uniform
__device__ __forceinline__ int wave_min_index(const int cand)
{
  return (int)(-wave_max(-(double)cand));  // exact: indices are far below 2^53
}

// Eigen 3.4 LDLT<., Upper> == oracle_ldlt_factor (oracle/qp_oracle.c:67-151) on the row-major lower triangle W
// (leading dimension ld, K x K), in global memory.  (Measured: a column-major layout -- coalesced reads in the column
// update, strided ones in the row swaps of the pivoting -- is no faster for one QP and mixed for batches: +30 % for the
// safety-filter shape (3, 203), -5 % for (100, 156).)  perm[K] (LDS): composed transpositions, (P b)[i] = b[perm[i]].
// temp[K] (LDS) scratch.  Returns 1 on success, 0 on failure (info() == NumericalIssue).  Wave-uniform.
// nzj[K] (LDS, ints) scratch: per column the indices j with temp(j) != 0.  A zero temp(j) adds row(j) * 0 = +-0 to the
// dot product of every row, which leaves it unchanged as long as row(j) is finite (the running sum is never -0: it
// starts at +0) -- so while every entry of L computed so far is finite (checked as the entries are written), the dot
// products run over the non-zero terms only, in the same order.  A safety filter's constraint columns have no non-zero
// term at all: their update vanishes.  Once a non-finite entry appears the full loops run again.
// Where the matrix lives: the row-major square in the QP's workspace (global memory).
struct BigMatGlobal {
  double *W;
  size_t ld;
  __device__ __forceinline__ double &at(const int i, const int j) const { return W[(size_t)i * ld + (size_t)j]; }
  __device__ __forceinline__ const double *row(const int i) const { return W + (size_t)i * ld; }
};
template<class Mat>
__device__ inline int big_ldlt_factor(const int K_, const Mat W, int *perm_, double *temp_, int *nzj_, const int lane)
{
  const int K = ubig(K_);
  LDS_I(perm, perm_);
  LDS_D(temp, temp_);
  LDS_I(nzj, nzj_);
  bool all_finite = true;  // wave-uniform: every entry of columns < kk (rows below the diagonal) and their pivots
#define WB(i, j) W.at((i), (j))
  for (int i = lane; i < K; i += kWave) perm[i] = i;
  wave_sync();
  if (K <= 1) return 1;
  int found_zero = 0, ret = 1;
  for (int kk = 0; kk < K; ++kk) {
    // pivot: first index of the largest |diag| among rows kk..K-1 (strict '>' scan; a NaN at kk keeps kk)
    const double dkk = fabs(WB(kk, kk));
    int p            = kk;
    if (!(dkk != dkk)) {
      double best = -1.0;
      int bi      = K;
      for (int i = kk + lane; i < K; i += kWave) {
        const double a = fabs(WB(i, i));
        if (a > best) { best = a; bi = i; }
      }
      const double mx = wave_max(best);
      p               = wave_min_index((best == mx) ? bi : K);
      if (p >= K) p = kk;
    }
    if (p != kk) {
      for (int t = lane; t < kk; t += kWave) { const double a = WB(kk, t); WB(kk, t) = WB(p, t); WB(p, t) = a; }
      for (int i = p + 1 + lane; i < K; i += kWave) { const double a = WB(i, kk); WB(i, kk) = WB(i, p); WB(i, p) = a; }
      for (int i = kk + 1 + lane; i < p; i += kWave) { const double a = WB(i, kk); WB(i, kk) = WB(p, i); WB(p, i) = a; }
      if (lane == 0) {
        const double a = WB(kk, kk); WB(kk, kk) = WB(p, p); WB(p, p) = a;
        const int t = perm[kk]; perm[kk] = perm[p]; perm[p] = t;
      }
      wave_sync();
    }
    // temp(j) = D(j) * L(kk, j), j < kk (and the list of its non-zero entries)
    int nn = 0;
    for (int j0 = 0; j0 < kk; j0 += kWave) {
      const int j    = j0 + lane;
      const double v = (j < kk) ? WB(j, j) * WB(kk, j) : 0.0;
      if (j < kk) temp[j] = v;
      const unsigned long long mk = wave_ballot(j < kk && v != 0.0);
      if (j < kk && v != 0.0) nzj[nn + __popcll(mk & lanemask_lt(lane))] = j;
      nn += __popcll(mk);
    }
    wave_sync();
    if (kk > 0 && (!all_finite || 2 * nn > kk)) {  // (mostly non-zero terms: the plain loop streams better than the indexed one)
      for (int i = kk + lane; i < K; i += kWave) {  // rows kk (the diagonal) .. K-1: dot, then subtract
        const auto row = W.row(i);
        double s          = 0.0;
        for (int j = 0; j < kk; ++j) s = fma(row[j], temp[j], s);
        WB(i, kk) -= s;
      }
      wave_sync();
    } else if (nn > 0) {
      for (int i = kk + lane; i < K; i += kWave) {
        const auto row = W.row(i);
        double s          = 0.0;
        for (int q = 0; q < nn; ++q) {
          const int j = nzj[q];
          s           = fma(row[j], temp[j], s);
        }
        WB(i, kk) -= s;
      }
      wave_sync();
    }
    const double akk = WB(kk, kk);
    const bool valid = fabs(akk) > 0.0;
    if (kk == 0 && !valid) {  // whole diagonal zero: success iff the strictly lower triangle is zero (perm = identity)
      bool nz = false;
      for (int i = lane; i < K; i += kWave)
        for (int j = 0; j < i; ++j) nz = nz || !(WB(i, j) == 0.0);
      return wave_ballot(nz) ? 0 : 1;
    }
    bool nzcol = false, fin = fabs(akk) < INFINITY;
    for (int i = kk + 1 + lane; i < K; i += kWave) {
      double v = WB(i, kk);
      if (valid) {
        v /= akk;
        WB(i, kk) = v;
      } else {
        nzcol = nzcol || !(v == 0.0);
      }
      fin = fin && fabs(v) < INFINITY;
    }
    if (all_finite && wave_ballot(!fin)) all_finite = false;
    if (!valid && wave_ballot(nzcol)) ret = 0;
    if (found_zero && valid) ret = 0;
    else if (!valid) found_zero = 1;
    wave_sync();
  }
  return ret;
#undef WB
}

constexpr int kBW = 16;  // column block of the triangular sweeps
constexpr int kBP     = 17;  // row stride of the 16-wide tiles kept in LDS (odd: lane = row reads hit distinct banks)
constexpr int kDiagCacheK = 384;  // up to this size the diagonal blocks of L are kept in LDS (k * 16 doubles)
// Compact on-chip copy of the OFF-diagonal part of L for factors that are almost empty (a safety filter: n = 3
// variables, hundreds of barrier rows with a diagonal (2,2) block): per 16-column block the rows that have a
// non-zero entry in it, each with its 16 values.  big_row_cap(n, m) rows in all (forward and backward lists share the
// pool); a factor with more keeps streaming its non-zero tiles from the workspace.
// Pool size: what is left of the 160 KB of a CU after everything else the kernel keeps in LDS (one wave per CU then --
// a lone QP is what this matters for; batches of such QPs are served by fewer, faster waves), at most 2 k + 64 rows
// (a safety filter needs n rows per block forward and, for the blocks holding the variables, all rows backward).
__host__ __device__ constexpr size_t big_lds_fixed_bytes(const int n, const int m, const bool diag_cache = true)
{
  const size_t k = (size_t)n + (size_t)m;
  const size_t rb = (k + 63) / 64 <= 2 ? 2 : ((k + 63) / 64 <= 4 ? 4 : ((k + 63) / 64 <= 8 ? 8 : 16));
  const size_t tiles = ((k + 15) / 16) * rb;  // non-zero maps of the sweep tiles, forward and backward
  const size_t dcache = (diag_cache && k <= (size_t)kDiagCacheK) ? ((k + 15) / 16) * 16 * kBP : 0;
  return (4 * k + 2 * (size_t)n + 6 * (size_t)m + 8 + dcache) * sizeof(double) +
         ((k + 1) / 2 * 2 + 2 * tiles + 5 * ((k + 15) / 16 + 1) + 4) * sizeof(int);
}
// `roomy`: few QPs in the launch (at most one wave per CU anyway): take what the lists of such a factor can need;
// otherwise k + 64 rows, which keeps two or three waves per CU for batches (measured at (40, 60): 3 waves per CU with
// k + 64, 2 with k + 128: 90 vs 118 ms for 4 096 QPs) (the lists are an optimisation: a factor
// they do not hold is streamed tile by tile, same results).
__host__ __device__ constexpr int big_row_cap(const int n, const int m, const bool roomy, const bool diag_cache = true)
{
  const size_t budget = 158 * 1024, fixed = big_lds_fixed_bytes(n, m, diag_cache), per_row = kBP * sizeof(double) + sizeof(int);
  const size_t k = (size_t)n + m, fit = budget > fixed ? (budget - fixed) / per_row : 0;
  // rows a completely dense factor puts into the lists (every row below / left of each 16-column block, both sweeps):
  // a lone QP of up to k = 128 keeps even that on chip
  const size_t nb = (k + 15) / 16;
  size_t dense_rows = 0;
  for (size_t jb = 0; jb < nb; ++jb) dense_rows += (k > 16 * (jb + 1) ? k - 16 * (jb + 1) : 0) + 16 * jb;
  const size_t roomy_want = dense_rows > 2 * k + 64 ? dense_rows : 2 * k + 64;
  const size_t want = roomy ? roomy_want : (k <= (size_t)kDiagCacheK ? k + 64 : 192);
  size_t cap        = fit < want ? fit : want;
  if (cap < 64) cap = 64;
  if (!roomy) {
    // batches: resident blocks per CU come in steps of the LDS request, and several sizes sit just above a step
    // ((3, 203): 84 KB = one block per CU, 255 rows instead of 270 make it two).  Give up to 40 % of the pool for one
    // more resident wave -- occupancy is what batches are short of, the lists are an optimisation.
    const size_t cu = 160 * 1024, total = fixed + cap * per_row, blocks = total ? cu / total : 1;
    const size_t share = (cu / (blocks + 1)) & ~(size_t)1023;  // per-block LDS for one more block, below any allocation granule
    if (share > fixed + 1024) {
      const size_t cap2 = (share - 1024 - fixed) / per_row;
      if (cap2 >= 64 && 10 * cap2 >= 6 * cap && cap2 < cap) cap = cap2;
    }
  }
  return (int)cap;
}

// LT(j, i) = W(i, j) for i > j (column j of L contiguous), Dg[i] = W(i, i), and the non-zero maps of the tiles the
// blocked sweeps work on: fnz[jb * RB + r] != 0 iff L has a non-zero entry in rows 64 r .. 64 r + 63, columns
// 16 jb .. 16 jb + 15 (forward sweep); bnz[jb * RB + r] the same for rows 16 jb .. + 15, columns 64 r .. + 63
// (backward sweep).  An all-zero tile contributes exact zeros to every chain it takes part in and is skipped --
// the KKT matrix of a safety filter (n = 3 variables, hundreds of barrier rows, a diagonal (2,2) block) leaves L
// almost empty.
template<int RB>
__device__ inline void big_transpose(const int K, const double *__restrict__ W, const int ld, double *__restrict__ LT,
                                     double *__restrict__ Dg, int *fnz_, int *bnz_, double *dblk_, double *LDg_, int *cidx_,
                                     double *cval_, int *cflag_, const int cap, const int lane)
{
  LDS_I(fnz, fnz_);
  LDS_I(bnz, bnz_);
  LDS_D(dblk, dblk_);
  LDS_D(LDg, LDg_);
  LDS_I(cidx, cidx_);
  LDS_D(cval, cval_);
  LDS_I(cflag, cflag_);
  // LT: the column-major copy of the strictly lower triangle (column j of L contiguous: the forward sweep's layout)
  for (int i = 0; i < K; ++i)
    for (int j = lane; j < i; j += kWave) LT[(size_t)j * ld + i] = W[(size_t)i * ld + j];
  // cidx: [0, nb] forward block offsets, [nb + 1, 2 nb + 1] backward block offsets, nb flags "the diagonal block has
  // entries below its diagonal", nb run lengths of the forward sweep and nb skip counts of the backward sweep (below), then the rows of the lists (forward lists
  // first, one pool of `cap` rows); cval: the rows' 16 values (stride kBP); cflag[0]: the lists are complete
  {
    const int nbk = (K + kBW - 1) / kBW;
    lds_i *const foff = cidx, *const boff = cidx + nbk + 1, *const dflag = boff + nbk + 1, *const frun = dflag + nbk,
                 *const bskip = frun + nbk, *const rows = bskip + nbk;
    int np = 0;
    for (int dir = 0; dir < 2; ++dir) {
      lds_i *const off = dir ? boff : foff;
      for (int jb = 0; jb < nbk; ++jb) {
        const int j0 = jb * kBW;
        if (lane == 0) off[jb] = np;
        const int ibeg = dir ? 0 : j0 + kBW, iend = dir ? j0 : K;
        for (int i0 = ibeg - ibeg % kWave; i0 < iend; i0 += kWave) {
          const int i = i0 + lane;
          bool nz = false;
          double v[kBW];
#pragma unroll
          for (int jj = 0; jj < kBW; ++jj) {
            const int j = j0 + jj;
            v[jj] = (i >= ibeg && i < iend && j < K) ? (dir ? W[(size_t)j * ld + i] : W[(size_t)i * ld + j]) : 0.0;
            nz    = nz || !(v[jj] == 0.0);
          }
          const unsigned long long mk = wave_ballot(nz);
          const int pp = np + __popcll(mk & lanemask_lt(lane));
          if (nz && pp < cap) {
            rows[pp] = i;
#pragma unroll
            for (int jj = 0; jj < kBW; ++jj) cval[pp * kBP + jj] = v[jj];
          }
          np += __popcll(mk);
        }
        if (dir == 0) {  // diagonal block: lane = row
          bool nz = false;
          const int i = j0 + (lane & (kBW - 1));
          for (int cc = 0; cc < (lane & (kBW - 1)); ++cc) nz = nz || (i < K && !(W[(size_t)i * ld + j0 + cc] == 0.0));
          const unsigned long long mk = wave_ballot(nz);
          if (lane == 0) dflag[jb] = (mk != 0ull || dblk_ == nullptr) ? 1 : 0;
        }
      }
      if (lane == 0) off[nbk] = np;
    }
    if (lane == 0) cflag[0] = (np <= cap) ? 1 : 0;
    wave_sync();
    // frun[jb] > 0: blocks jb .. jb + frun[jb] - 1 have no entries below the diagonal inside their diagonal blocks, the
    // SAME (at most 64) rows below them, and all of those rows lie beyond the run's columns -- the constraint columns
    // of a safety filter, whose only dependants are the variables' rows.  The forward sweep then carries those rows in
    // registers through the whole run (big_solve).  0: the block takes the general path.
    if (np <= cap) {
      int end_same = nbk;  // wave-uniform: end of the run of identical lists the block after jb belongs to
      for (int jb = nbk - 1; jb >= 0; --jb) {
        const int e0 = ubig(foff[jb]), c = ubig(foff[jb + 1]) - e0;
        const bool plain = ubig(dflag[jb]) == 0 && c > 0 && c <= kWave;
        bool same = false;
        if (plain && jb + 1 < nbk && ubig(frun[jb + 1]) > 0 && ubig(foff[jb + 2]) - ubig(foff[jb + 1]) == c) {
          const int e1 = ubig(foff[jb + 1]);
          same = wave_ballot(lane < c && rows[e0 + (lane < c ? lane : 0)] != rows[e1 + (lane < c ? lane : 0)]) == 0ull;
        }
        if (!same) end_same = jb + 1;
        int len = 0;
        if (plain) {
          const int lim = ubig(rows[e0]) / kBW;  // the smallest dependent row must lie beyond the run
          len = (end_same < lim ? end_same : lim) - jb;
          if (len < 1) len = 0;
        }
        if (!plain) end_same = jb;  // nothing continues through a general block
        if (lane == 0) frun[jb] = len;
        wave_sync();
      }
    } else {
      for (int jb = lane; jb < nbk; jb += kWave) frun[jb] = 0;
    }
    // bskip[jb]: how many blocks jb, jb - 1, ... in a row the backward sweep has nothing to do for (no chain, nothing
    // depends on them: the constraint columns of a safety filter) -- it jumps over them with one LDS read
    {
      int cnt = 0;
      for (int jb = 0; jb < nbk; ++jb) {
        const bool idle = np <= cap && ubig(dflag[jb]) == 0 && ubig(boff[jb]) == ubig(boff[jb + 1]);
        cnt = idle ? cnt + 1 : 0;
        if (lane == 0) bskip[jb] = cnt;
      }
    }
  }
  for (int i = lane; i < K; i += kWave) LDg[i] = W[(size_t)i * ld + i];
  if (dblk_ != nullptr) {  // dblk[jb][r][c] = L(16 jb + r, 16 jb + c), r > c (other entries 0)
    const int nbk = (K + kBW - 1) / kBW;
    for (int e = lane; e < nbk * kBW * kBW; e += kWave) {
      const int jb = e / (kBW * kBW), r = (e / kBW) % kBW, cc = e % kBW, i = jb * kBW + r, j = jb * kBW + cc;
      dblk[(jb * kBW + r) * kBP + cc] = (r > cc && i < K) ? W[(size_t)i * ld + j] : 0.0;
    }
  }
  for (int i = lane; i < K; i += kWave) Dg[i] = W[(size_t)i * ld + i];
  const int nb = (K + kBW - 1) / kBW;
  for (int jb = 0; jb < nb; ++jb)
    for (int r = 0; r < RB; ++r) {
      const int i = lane + kWave * r;
      bool nzf = false, nzb = false;
      for (int jj = 0; jj < kBW; ++jj) {
        const int j = jb * kBW + jj;
        if (j < K && i < K && i > j) nzf = nzf || !(W[(size_t)i * ld + j] == 0.0);
        if (j < K && i < j) nzb = nzb || !(W[(size_t)j * ld + i] == 0.0);
      }
      const unsigned long long bf = wave_ballot(nzf), bb = wave_ballot(nzb);
      if (lane == 0) { fnz[jb * RB + r] = bf != 0ull; bnz[jb * RB + r] = bb != 0ull; }
    }
  wave_sync();
}

// LDLT::_solve_impl == oracle_ldlt_solve: temp (LDS, K entries, original order) <- P^T L^-T D^-1 L^-1 P temp; t (LDS, K
// entries) is the permuted work vector -- the two permutations are the copies between them.
// Per row the oracle subtracts L(i, j) t_j for j ascending (forward) / L(j, i) t_j for j descending (backward); here
// the columns come in BLOCKS of 16: the 16 rows of the diagonal block are finished first (in registers: lane = row,
// the pivot travels by v_readlane), then every other row takes the block's 16 updates in the same ascending
// (descending) order -- ONE memory round trip per block instead of one per column.  |d| <= DBL_MIN -> 0, true division.
#ifdef SFB_BIG_PROF
__device__ unsigned long long g_bigprof[12];
#define BP_T(x) const unsigned long long x = wall_clock64()
#define BP_ADD(i, a, b) if (lane == 0 && blockIdx.x == 0) g_bigprof[i] += (b) - (a)
#else
#define BP_T(x)
#define BP_ADD(i, a, b)
#endif
template<int RB>
__device__ inline void big_solve(const int K, const double *__restrict__ W, const double *__restrict__ LT,
                                 const double *__restrict__ Dg, const int ld, const int *perm_, const int *fnz_, const int *bnz_,
                                 const double *dblk_, const double *LDg_, const int *cidx_, const double *cval_,
                                 const int *cflag_, double *t_, double *temp_, const int lane)
{
  LDS_CI(perm, perm_);
  LDS_CI(fnz, fnz_);
  LDS_CI(bnz, bnz_);
  LDS_CD(dblk, dblk_);
  LDS_CD(LDg, LDg_);
  LDS_CI(cidx, cidx_);
  LDS_CD(cval, cval_);
  LDS_CI(cflag, cflag_);
  LDS_D(t, t_);
  LDS_D(temp, temp_);
  const int nbk0    = (K + kBW - 1) / kBW;
  const bool compact = cflag[0] != 0;  // wave-uniform (LDS)
  const lds_i *const foff = cidx, *const boff = cidx + nbk0 + 1, *const dflag = boff + nbk0 + 1, *const frun = dflag + nbk0,
                     *const bskip = frun + nbk0, *const frow = bskip + nbk0, *const brow = frow;
  const lds_d *const fval = cval, *const bval = cval;
  BP_T(p0);
  for (int i = lane; i < K; i += kWave) t[i] = temp[perm[i]];  // the right-hand side comes in temp (natural order)
  wave_sync();
  const int nb = (K + kBW - 1) / kBW;
  BP_T(p1);
  BP_ADD(0, p0, p1);
  constexpr int RC = 2;  // row blocks handled per memory round trip (register budget)
  // The hot loops are written branch-free: loads go to clamped (always valid) addresses and lanes / columns that do
  // not take part are switched off by selects -- predicated `if`s compile to exec-mask juggling and branches that
  // cost more than the arithmetic.  Only the wave-uniform skip of an all-zero tile is a (scalar) branch.
  // ---- forward ----
  for (int jb = 0; jb < nb; ++jb) {
    const int j0 = jb * kBW, l0 = j0 % kWave, rb = j0 / kWave, nbw = (K - j0 < kBW) ? K - j0 : kBW;
    const bool inblk = lane >= l0 && lane < l0 + nbw;
    const int lrow   = inblk ? lane - l0 : 0, iblk = j0 + lrow;
    double dv[kBW], tb[kBW];
    // wave-uniform shortcuts: a diagonal block without entries below its diagonal needs no chain, a block no other
    // row depends on needs no broadcast -- the constraint columns of a safety filter are both in the backward sweep
    const int e0 = compact ? ubig(foff[jb]) : 0, e1 = compact ? ubig(foff[jb + 1]) : 0;
    const bool chain = !compact || ubig(dflag[jb]) != 0;
    if (compact && !chain && e0 == e1) continue;
    if (const int run = compact ? ubig(frun[jb]) : 0; run > 0) {
      // a run of blocks without chains whose only dependants are the same few rows beyond the run (big_transpose): one
      // lane per dependent row keeps its entry in a register through the whole run; the columns' values are final and
      // read straight from LDS (broadcast reads) -- no cross-lane traffic, no store / fence between the blocks.  The
      // updates of a row happen in the same order as block by block.
      const bool eon = lane < e1 - e0;
      const int erow = frow[e0 + (eon ? lane : 0)];
      double esv     = t[erow];
      const int cnt = e1 - e0;  // the lists of a run have the same length and follow each other in the pool
      for (int b = jb, eb = e0 + (eon ? lane : 0); b < jb + run; ++b, eb += cnt) {
        const int c0 = b * kBW;
        double fv[kBW], tt[kBW];
#pragma unroll
        for (int jj = 0; jj < kBW; ++jj) {
          fv[jj] = fval[eb * kBP + jj];
          tt[jj] = t[c0 + jj];  // (past K in the last block: whatever follows t in LDS, its fv is an exact zero and skipped)
        }
#pragma unroll
        for (int jj = 0; jj < kBW; ++jj) {
          const double nv = fma(-fv[jj], tt[jj], esv);
          esv             = (fv[jj] != 0.0) ? nv : esv;  // (lanes without a row compute on lane 0's data and store nothing)
        }
      }
      if (eon) t[erow] = esv;
      wave_lds_fence();
      jb += run - 1;
      continue;
    }
    // everything the block needs is requested up front (one LDS / memory latency, not one per column): the diagonal
    // block, and the first 64 of the rows below it that have entries in the block
    if (!chain) {
    } else if (dblk_ != nullptr) {  // (lanes outside the block read its first row: empty, like the entries on and above the diagonal)
#pragma unroll
      for (int jj = 0; jj < kBW; ++jj) dv[jj] = dblk[(jb * kBW + lrow) * kBP + jj];
    } else {
#pragma unroll
      for (int jj = 0; jj < kBW; ++jj) {
        const double v = LT[(size_t)(j0 + ((jj < nbw) ? jj : 0)) * ld + iblk];
        dv[jj]         = (inblk && jj < nbw && lrow > jj) ? v : 0.0;
      }
    }
    const bool eon = e0 + lane < e1;
    const int ec   = eon ? e0 + lane : 0;
    double fv[kBW];
#pragma unroll
    for (int jj = 0; jj < kBW; ++jj) fv[jj] = fval[ec * kBP + jj];
    const int erow = eon ? frow[ec] : 0;
    double esv     = t[erow];
    double treg    = t[iblk];
    BP_T(q0);
    if (chain) {
#pragma unroll
      for (int jj = 0; jj < kBW; ++jj) {
        tb[jj]          = lane_bcast(treg, l0 + ((jj < nbw) ? jj : 0));
        const double nv = fma(-dv[jj], tb[jj], treg);
        treg            = (dv[jj] != 0.0) ? nv : treg;  // (a zero entry leaves the row untouched, NaN pivots included)
      }
      if (inblk) t[iblk] = treg;
    } else {
#pragma unroll
      for (int jj = 0; jj < kBW; ++jj) tb[jj] = lane_bcast(treg, l0 + ((jj < nbw) ? jj : 0));
    }
    BP_T(q1);
    BP_ADD(2, q0, q1);
    if (compact) {  // the few rows below the block that have entries in it, from LDS
#pragma unroll
      for (int jj = 0; jj < kBW; ++jj) {
        const double nv = fma(-fv[jj], tb[jj], esv);
        esv             = (eon && fv[jj] != 0.0) ? nv : esv;
      }
      if (eon) t[erow] = esv;
      for (int e = e0 + kWave + lane; e < e1; e += kWave) {
        const int i = frow[e];
        double sv   = t[i];
#pragma unroll
        for (int jj = 0; jj < kBW; ++jj) {
          const double lvv = fval[e * kBP + jj];
          const double nv  = fma(-lvv, tb[jj], sv);
          sv               = (lvv != 0.0) ? nv : sv;
        }
        t[i] = sv;
      }
    } else
    for (int r0 = rb; r0 < RB; r0 += RC) {
      bool any = false;
#pragma unroll
      for (int rr = 0; rr < RC; ++rr) any = any || (r0 + rr < RB && fnz[jb * RB + ((r0 + rr < RB) ? r0 + rr : 0)] != 0);
      if (!any) continue;  // wave-uniform
      double lv[RC][kBW];
      int ic[RC];
      bool on[RC];
#pragma unroll
      for (int rr = 0; rr < RC; ++rr) {
        const int i = lane + kWave * (r0 + rr);
        on[rr]      = i >= j0 + kBW && i < K;
        ic[rr]      = on[rr] ? i : K - 1;
#pragma unroll
        for (int jj = 0; jj < kBW; ++jj) lv[rr][jj] = LT[(size_t)(j0 + ((jj < nbw) ? jj : 0)) * ld + ic[rr]];
      }
#pragma unroll
      for (int rr = 0; rr < RC; ++rr) {
        double sv = t[ic[rr]];
#pragma unroll
        for (int jj = 0; jj < kBW; ++jj) {
          const double nv = fma(-lv[rr][jj], tb[jj], sv);
          sv              = (jj < nbw && lv[rr][jj] != 0.0) ? nv : sv;
        }
        if (on[rr]) t[ic[rr]] = sv;
      }
    }
    wave_lds_fence();
    BP_T(q2);
    BP_ADD(3, q1, q2);
  }
  BP_T(p2);
  BP_ADD(1, p1, p2);
  for (int i = lane; i < K; i += kWave) {
    const double d = LDg[i];
    t[i]           = (fabs(d) > DBL_MIN) ? t[i] / d : 0.0;
  }
  wave_lds_fence();
  BP_T(p3);
  BP_ADD(4, p2, p3);
  // ---- backward ----
  for (int jb = nb - 1; jb >= 0; --jb) {
    const int j0 = jb * kBW, l0 = j0 % kWave, rb = j0 / kWave, nbw = (K - j0 < kBW) ? K - j0 : kBW;
    const bool inblk = lane >= l0 && lane < l0 + nbw;
    const int lrow   = inblk ? lane - l0 : 0, iblk = j0 + lrow, lcol = inblk ? lane - l0 : kBW - 1;
    double dv[kBW], tb[kBW];
    if (const int idle = compact ? ubig(bskip[jb]) : 0; idle > 0) {
      jb -= idle - 1;
      continue;
    }
    const int e0 = compact ? ubig(boff[jb]) : 0, e1 = compact ? ubig(boff[jb + 1]) : 0;
    const bool chain = !compact || ubig(dflag[jb]) != 0;
    BP_T(r0);
    if (!chain) {
    } else if (dblk_ != nullptr) {  // (lanes outside the block read its last column: empty)
#pragma unroll
      for (int jj = 0; jj < kBW; ++jj) dv[jj] = dblk[(jb * kBW + jj) * kBP + lcol];
    } else {
#pragma unroll
      for (int jj = 0; jj < kBW; ++jj) {
        const double v = W[(size_t)(j0 + ((jj < nbw) ? jj : 0)) * ld + iblk];
        dv[jj]         = (inblk && jj < nbw && lrow < jj) ? v : 0.0;
      }
    }
    const bool eon = e0 + lane < e1;
    const int ec   = eon ? e0 + lane : 0;
    double fv[kBW];
#pragma unroll
    for (int jj = 0; jj < kBW; ++jj) fv[jj] = bval[ec * kBP + jj];
    const int erow = eon ? brow[ec] : 0;
    double esv     = t[erow];
    double treg    = t[iblk];
    if (chain) {
#pragma unroll
      for (int jj = kBW - 1; jj >= 0; --jj) {
        tb[jj]          = lane_bcast(treg, l0 + ((jj < nbw) ? jj : 0));
        const double nv = fma(-dv[jj], tb[jj], treg);
        treg            = (dv[jj] != 0.0) ? nv : treg;
      }
      if (inblk) t[iblk] = treg;
    } else {
#pragma unroll
      for (int jj = 0; jj < kBW; ++jj) tb[jj] = lane_bcast(treg, l0 + ((jj < nbw) ? jj : 0));
    }
    if (compact) {
#pragma unroll
      for (int jj = kBW - 1; jj >= 0; --jj) {
        const double nv = fma(-fv[jj], tb[jj], esv);
        esv             = (eon && fv[jj] != 0.0) ? nv : esv;
      }
      if (eon) t[erow] = esv;
      for (int e = e0 + kWave + lane; e < e1; e += kWave) {
        const int i = brow[e];
        double sv   = t[i];
#pragma unroll
        for (int jj = kBW - 1; jj >= 0; --jj) {
          const double lvv = bval[e * kBP + jj];
          const double nv  = fma(-lvv, tb[jj], sv);
          sv               = (lvv != 0.0) ? nv : sv;
        }
        t[i] = sv;
      }
    } else
    for (int r0 = 0; r0 <= rb; r0 += RC) {
      bool any = false;
#pragma unroll
      for (int rr = 0; rr < RC; ++rr) any = any || (r0 + rr <= rb && bnz[jb * RB + ((r0 + rr < RB) ? r0 + rr : 0)] != 0);
      if (!any) continue;  // wave-uniform
      double lv[RC][kBW];
      int ic[RC];
      bool on[RC];
#pragma unroll
      for (int rr = 0; rr < RC; ++rr) {
        const int i = lane + kWave * (r0 + rr);
        on[rr]      = i < j0;
        ic[rr]      = on[rr] ? i : 0;
#pragma unroll
        for (int jj = 0; jj < kBW; ++jj) lv[rr][jj] = W[(size_t)(j0 + ((jj < nbw) ? jj : 0)) * ld + ic[rr]];
      }
#pragma unroll
      for (int rr = 0; rr < RC; ++rr) {
        double sv = t[ic[rr]];
#pragma unroll
        for (int jj = kBW - 1; jj >= 0; --jj) {
          const double nv = fma(-lv[rr][jj], tb[jj], sv);
          sv              = (jj < nbw && lv[rr][jj] != 0.0) ? nv : sv;
        }
        if (on[rr]) t[ic[rr]] = sv;
      }
    }
    wave_lds_fence();
    BP_T(r1);
    BP_ADD(6, r0, r1);
    BP_ADD(11, 0ull, 1ull);
  }
  BP_T(p4);
  BP_ADD(7, p3, p4);
  for (int i = lane; i < K; i += kWave) temp[perm[i]] = t[i];  // ... and the solution leaves in temp (natural order)
  wave_sync();
}

__device__ __forceinline__ double big_norm_inf(const double *v, const int len, const int lane)
{
  double r = 0.0;
  for (int e = lane; e < len; e += kWave) r = fmax(r, fabs(v[e]));
  return wave_max(r);
}

struct BigWs {
  double *H, *LT, *Hs, *Dg;                                                // k*k each, k
  double *sx, *xus, *dxus, *Px, *Aty, *hx;                                 // n each
  double *sy, *rho, *rinv, *lo, *hi, *yus, *zus, *dyus, *Ax, *act, *pos;   // m each
};

}  // namespace

size_t qp_dense_big_ws_doubles(int n, int m)
{
  const size_t k = (size_t)n + m;
  return 3 * k * k + k + 6 * (size_t)n + 11 * (size_t)m + 8;
}
// LDS configuration of a launch: the row pool (big_row_cap) and whether the diagonal blocks are cached.  A lone QP wants
// both; a batch must not be left with ONE wave per CU -- (4, 301) with the diagonal cache is 122 KB, without it 79 KB.
struct BigLds { bool diag_cache; int rcap; size_t bytes; };
static BigLds big_lds_config(int n, int m, int64_t batch)
{
  const size_t per_row = kBP * sizeof(double) + sizeof(int), cu = 160 * 1024;
  const bool roomy = batch < kRoomyBatch;
  auto make = [&](bool dc) {
    const int rc = big_row_cap(n, m, roomy, dc);
    return BigLds{dc, rc, big_lds_fixed_bytes(n, m, dc) + (size_t)rc * per_row};
  };
  const BigLds with = make(true);
  if (roomy || n + m > kDiagCacheK) return with;
  const BigLds without = make(false);
  // measured (scripts/dense_big_time.py): a second resident block is worth the cache ((4, 301): 2.8 -> 3.7 k QP/s), a third
  // or later one is not ((3, 203): 23.9 k QP/s with the cache and 2 blocks, 13.7 k without it and 3)
  return (cu / with.bytes <= 1 && cu / without.bytes >= 2) ? without : with;
}
size_t qp_dense_big_lds_bytes(int n, int m, int64_t batch) { return big_lds_config(n, m, batch).bytes; }

template<int RB>
__global__ void __launch_bounds__(64) qp_dense_big_kernel(const DenseKernelParams kp, const QpBatch g, double *__restrict__ gws,
                                                         const size_t wsd, const int rcap, const int dck)
{
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int lane = threadIdx.x;
  const int n = kp.n, m = kp.m, k = n + m;
  const size_t b = blockIdx.x;
  const double *P = g.P + b * (size_t)n * n, *q = g.q + b * (size_t)n, *A = g.A + b * (size_t)m * n;
  const double *l = g.l + b * (size_t)m, *u = g.u + b * (size_t)m;
  // LDS: work vector t (k), scratch temp (k + 8: the polish system is never larger than k), second scratch (k),
  // iterate x (n), y (m), z (m), permutation (k ints)
  double *t = sm, *temp = t + k, *aux = temp + k + 8, *xs = aux + k, *ys = xs + n, *zs = ys + m;
  // the per-iteration constants of the ADMM loop (a lone wave pays a memory round trip per phase otherwise)
  double *Lrho = zs + m, *Lrinv = Lrho + m, *Llo = Lrinv + m, *Lhi = Llo + m, *Lqc = Lhi + m;
  // diagonal 16 x 16 blocks of L for the blocked sweeps, cached in LDS while they fit (k <= dck: kDiagCacheK, or 0 for
  // batches that gain a resident block per CU without the cache)
  double *dblk = Lqc + n;
  double *LDg  = dblk + ((k <= dck) ? (size_t)((k + kBW - 1) / kBW) * kBW * kBP : 0);  // diagonal of the factor
  double *cval = LDg + k;                                                                        // compact off-diagonal rows
  int *perm = reinterpret_cast<int *>(cval + rcap * kBP);
  int *fnz = perm + (k + 1) / 2 * 2, *bnz = fnz + ((k + kBW - 1) / kBW) * RB;
  int *cidx = bnz + ((k + kBW - 1) / kBW) * RB, *cflag = cidx + 5 * ((k + kBW - 1) / kBW + 1) + rcap;
  BigWs w;
  {
    double *p = gws + b * wsd;
    const size_t kk2 = (size_t)k * k;
    w.H = p; p += kk2;  w.LT = p; p += kk2;  w.Hs = p; p += kk2;  w.Dg = p; p += k;
    w.sx = p; p += n;   w.xus = p; p += n;   w.dxus = p; p += n;  w.Px = p; p += n;  w.Aty = p; p += n;  w.hx = p; p += n;
    w.sy = p; p += m;   w.rho = p; p += m;   w.rinv = p; p += m;  w.lo = p; p += m;  w.hi = p; p += m;
    w.yus = p; p += m;  w.zus = p; p += m;   w.dyus = p; p += m;  w.Ax = p; p += m;  w.act = p; p += m;  w.pos = p; p += m;
  }
  const double inf = INFINITY;
#define PM(i, j) P[(size_t)(i) + (size_t)(j) * (size_t)n]
#define AM(i, j) A[(size_t)(i) + (size_t)(j) * (size_t)m]

  // ---- analyze(): :306-308 ----
  for (int j = lane; j < n; j += kWave) w.sx[j] = 1.0;
  for (int i = lane; i < m; i += kWave) w.sy[i] = 1.0;
  wave_sync();
  double c = 1.0;

  // ---- scale :673-730 ----
  if (kp.scaling) {
    for (int col = 0; col < n; ++col) {  // :681-690 column inf-norms of P
      double v = 0.0;
      for (int row = lane; row < n; row += kWave) v = fmax(v, fabs(PM(row, col)));
      v = wave_max(v);
      if (v == 0.0) v = 1.0;
      if (lane == 0) temp[col] = v;
    }
    wave_sync();
    double sum = temp[0];
    for (int j = 1; j < n; ++j) sum += temp[j];
    const double qn = big_norm_inf(q, n, lane);
    c               = 1.0 / fmax(fmax(1e-6, sum / (double)n), qn);  // :693
    wave_sync();
    int pass = 0;
    double crit;
    do {  // :698-729: increments into temp[0..n) (columns) and aux[0..m) (rows) from the OLD sx, sy
      for (int i = lane; i < m; i += kWave) aux[i] = 0.0;
      wave_sync();
      for (int col = 0; col < n; ++col) {
        const double sxc = w.sx[col];
        double v         = 0.0;
        for (int row = lane; row < n; row += kWave) v = fmax(v, fabs(c * w.sx[row] * sxc * PM(row, col)));  // :704-707
        for (int row = lane; row < m; row += kWave) {                                                        // :712-714
          const double a = fabs(w.sy[row] * sxc * AM(row, col));
          v              = fmax(v, a);
          aux[row]       = fmax(aux[row], a);
        }
        v = wave_max(v);
        if (lane == 0) temp[col] = v;
      }
      wave_sync();
      double cm = 0.0;
      for (int j = lane; j < n; j += kWave) {
        double inc = temp[j];
        if (inc == 0.0) inc = 1.0;
        cm      = fmax(cm, fabs(inc - 1.0));
        w.sx[j] = sqrt(1.0 / fmax(inc, 1e-8)) * w.sx[j];
      }
      for (int i = lane; i < m; i += kWave) {
        double inc = aux[i];
        if (inc == 0.0) inc = 1.0;
        cm      = fmax(cm, fabs(inc - 1.0));
        w.sy[i] = sqrt(1.0 / fmax(inc, 1e-8)) * w.sy[i];
      }
      crit = wave_max(cm);
      wave_sync();
    } while (pass++ < 10 && crit > 0.1);
  }

  // ---- pre-check and rho :361-374 ----
  int ret_code = -1;
  {
    bool bad = false;
    for (int i = lane; i < m; i += kWave) {
      const double li = l[i], ui = u[i], syi = w.sy[i];
      bad = bad || (li == inf) || (ui == -inf) || (ui - li < 0.0);
      double rho;
      if (li == -inf && ui == inf) rho = 1e-6;
      else if (syi * fabs(li - ui) < 1e-5) rho = 1e3 * kp.rho_bar;
      else rho = kp.rho_bar;
      w.rho[i]  = rho;
      Lrho[i]   = rho;
      Lrinv[i]  = 1.0 / rho;
      Llo[i]    = syi * li;
      Lhi[i]    = syi * ui;
    }
    if (wave_ballot(bad)) ret_code = SFB_QP_PRIMAL_INFEASIBLE;
    for (int j = lane; j < n; j += kWave) Lqc[j] = c * w.sx[j] * q[j];  // (c sx_j) q_j of the right-hand side (:450)
  }
  wave_sync();

  const unsigned long long t0_ticks = wall_clock64();  // :376
  BP_T(u0);
  // ---- dense KKT fill :399-404 (row-major lower triangle of the k x k array H) ----
  const BigMatGlobal Hg{w.H, (size_t)k};
  auto kkt_fill = [&](const auto Hm) {
    for (int r = 0; r < n; ++r)
      for (int cc = lane; cc <= r; cc += kWave) {
        double v = c * w.sx[cc] * PM(cc, r) * w.sx[r];
        if (cc == r) v += kp.sigma;
        Hm.at(r, cc) = v;
      }
    for (int e = lane; e < m * n; e += kWave) {
      const int i = e % m, j = e / m;
      Hm.at(n + i, j) = w.sy[i] * A[e] * w.sx[j];
    }
    for (int i = 0; i < m; ++i) {
      for (int i2 = lane; i2 < i; i2 += kWave) Hm.at(n + i, n + i2) = 0.0;
      if (lane == 0) Hm.at(n + i, n + i) = 1.0 / (-w.rho[i]);
    }
  };
  kkt_fill(Hg);
  wave_sync();
  BP_T(u1);
  const int fact_ok = big_ldlt_factor(k, Hg, perm, temp, reinterpret_cast<int *>(t), lane);  // :428-433
  if (!fact_ok) ret_code = SFB_QP_UNKNOWN;
  BP_T(u2);
  const bool dcache = k <= dck;
  big_transpose<RB>(k, w.H, k, w.LT, w.Dg, fnz, bnz, dcache ? dblk : nullptr, LDg, cidx, cval, cflag, rcap, lane);
  BP_T(u3);
  BP_ADD(8, u0, u1);
  BP_ADD(9, u1, u2);
  BP_ADD(10, u2, u3);

  // ---- initial iterate :436-445 ----
  if (g.wx != nullptr) {
    const double *wx = g.wx + b * (size_t)n, *wy = g.wy + b * (size_t)m;
    for (int j = lane; j < n; j += kWave) xs[j] = (1.0 / w.sx[j]) * wx[j];
    for (int i = lane; i < m; i += kWave) {
      ys[i]     = c * ((1.0 / w.sy[i]) * wy[i]);
      double s  = 0.0;
      for (int j = 0; j < n; ++j) s = fma(w.sy[i] * AM(i, j), wx[j], s);
      zs[i] = s;
    }
  } else {
    for (int j = lane; j < n; j += kWave) xs[j] = 0.0;
    for (int i = lane; i < m; i += kWave) { ys[i] = 0.0; zs[i] = 0.0; }
  }
  wave_sync();

  // mat-vecs of check_stopping in the oracle's order: s = 0, inner index ascending, fma
  auto mv_A = [&](const double *v, double *out) {
    for (int i = lane; i < m; i += kWave) {
      double s = 0.0;
      for (int j = 0; j < n; ++j) s = fma(AM(i, j), v[j], s);
      out[i] = s;
    }
  };
  auto mv_At = [&](const double *v, double *out) {
    for (int j = lane; j < n; j += kWave) {
      double s = 0.0;
      for (int i = 0; i < m; ++i) s = fma(AM(i, j), v[i], s);
      out[j] = s;
    }
  };
  auto mv_P = [&](const double *v, double *out) {
    for (int i = lane; i < n; i += kWave) {
      double s = 0.0;
      for (int j = 0; j < n; ++j) s = fma(PM(i, j), v[j], s);
      out[i] = s;
    }
  };

  // check_stopping :574-644 on xus, yus, zus, dxus, dyus in the workspace; returns the status or -1
  auto stop_check = [&]() -> int {
    int code = -1;
    mv_A(w.xus, w.Ax);
    wave_sync();
    const double Ax_norm = big_norm_inf(w.Ax, m, lane);
    double rn = 0.0;
    for (int i = lane; i < m; i += kWave) rn = fmax(rn, fabs(w.Ax[i] - w.zus[i]));
    rn = wave_max(rn);
    if (rn <= kp.eps_abs + kp.eps_rel * fmax(Ax_norm, big_norm_inf(w.zus, m, lane))) {
      mv_P(w.xus, w.Px);
      mv_At(w.yus, w.Aty);
      wave_sync();
      const double dual_scale = fmax(fmax(big_norm_inf(w.Px, n, lane), big_norm_inf(q, n, lane)), big_norm_inf(w.Aty, n, lane));
      double dn = 0.0;
      for (int j = lane; j < n; j += kWave) dn = fmax(dn, fabs(w.Px[j] + (q[j] + w.Aty[j])));  // :592
      if (wave_max(dn) <= kp.eps_abs + kp.eps_rel * dual_scale) code = SFB_QP_OPTIMAL;
    }
    if (code < 0) {  // primal infeasibility :598-621
      mv_At(w.dyus, w.Aty);
      wave_sync();
      const double Edy = big_norm_inf(w.dyus, m, lane);
      const double thr = kp.eps_pinf * Edy;
      // the ordered sum with its early exit to +inf: +inf iff some row has an unbounded side beyond the threshold
      // BEFORE ... no: the oracle breaks at the first such row, having added the finite terms of the rows before
      // it and possibly the u-term of that row -- but then overwrites the sum with +inf, so only "any such row"
      // matters; otherwise the sum is the ordered one (a skipped term adds +0.0, exact: the sum is never -0.0)
      bool brk = false;
      for (int i = lane; i < m; i += kWave) {
        const double ui = u[i], li = l[i], dyi = w.dyus[i];
        aux[i]  = (ui != inf) ? ui * fmax(0.0, dyi) : 0.0;
        temp[i] = (li != -inf) ? li * fmin(0.0, dyi) : 0.0;
        brk = brk || (ui == inf && dyi > thr) || (li == -inf && dyi < -thr);
      }
      wave_sync();
      double s = 0.0;
      for (int i = 0; i < m; ++i) { s += aux[i]; s += temp[i]; }
      if (wave_ballot(brk)) s = inf;
      const double an = big_norm_inf(w.Aty, n, lane);
      if (((an < s) ? s : an) < thr) code = SFB_QP_PRIMAL_INFEASIBLE;
      wave_sync();
    }
    if (code < 0) {  // dual infeasibility :625-641
      mv_A(w.dxus, w.Ax);
      mv_P(w.dxus, w.Px);
      wave_sync();
      const double dxn = big_norm_inf(w.dxus, n, lane);
      const double thr = kp.eps_dinf * dxn;
      for (int j = lane; j < n; j += kWave) { aux[j] = q[j]; temp[j] = w.dxus[j]; }
      wave_sync();
      double qdx = 0.0;
      for (int j = 0; j < n; ++j) qdx = fma(aux[j], temp[j], qdx);
      bool ok = (big_norm_inf(w.Px, n, lane) <= thr) && (qdx <= thr);
      bool rowok = true;
      for (int i = lane; i < m; i += kWave) {
        const double Adx = w.Ax[i];
        if (u[i] == inf) rowok = rowok && (Adx >= -thr);
        else if (l[i] == -inf) rowok = rowok && (Adx <= thr);
        else rowok = rowok && (fabs(Adx) < thr);
      }
      if (ok && !wave_ballot(!rowok)) code = SFB_QP_DUAL_INFEASIBLE;
      wave_sync();
    }
    return code;
  };

  // ---- ADMM loop :447-510 ----
  uint32_t iter        = 0;
  const uint32_t sci   = kp.stop_check_iter;
  const uint32_t maxit = kp.max_iter;
  for (; iter != maxit && ret_code < 0; ++iter) {
    for (int j = lane; j < n; j += kWave) temp[j] = kp.sigma * xs[j] - Lqc[j];               // :450
    for (int i = lane; i < m; i += kWave) temp[n + i] = zs[i] - Lrinv[i] * ys[i];           // :451
    wave_sync();
    BP_T(s0);
    big_solve<RB>(k, w.H, w.LT, w.Dg, k, perm, fnz, bnz, dcache ? dblk : nullptr, LDg, cidx, cval, cflag, t, temp, lane);  // :462
    BP_T(s1);
    BP_ADD(5, s0, s1);
    const bool chk = (sci != 0) && (iter % sci == 1);                                       // :465
    for (int j = lane; j < n; j += kWave) {                                                 // :470
      const double xo = xs[j], xn = kp.alpha * temp[j] + kp.alpha_comp * xo;
      xs[j] = xn;
      if (chk) {
        w.xus[j]  = w.sx[j] * xn;
        w.dxus[j] = w.sx[j] * (xn - xo);
      }
    }
    for (int i = lane; i < m; i += kWave) {                                                 // :471-477
      const double ri = Lrinv[i], rh = Lrho[i], yo = ys[i], zo = zs[i], nu = temp[n + i];
      double zn = kp.alpha * (ri * nu) + kp.alpha_comp * (ri * yo) + zo;
      zn        = (zn < Llo[i]) ? Llo[i] : zn;
      zn        = (Lhi[i] < zn) ? Lhi[i] : zn;
      const double yn = kp.alpha_comp * yo + kp.alpha * nu + rh * zo - rh * zn;
      ys[i] = yn;
      zs[i] = zn;
      if (chk) {
        const double syi = w.sy[i];
        w.yus[i]  = syi * yn / c;
        w.zus[i]  = (1.0 / syi) * zn;
        w.dyus[i] = syi * (yn - yo) / c;
      }
    }
    wave_sync();
    if (chk) {
      ret_code = stop_check();
      if (ret_code < 0 && max_time_exceeded(kp.max_time_ns, t0_ticks)) ret_code = SFB_QP_MAX_TIME;  // :504-507
    }
  }

#ifdef SFB_BIG_PROF
  if (lane == 0 && blockIdx.x == 0) {
    printf("bigprof iters %u compact %d | x10ns: perm-in %llu forward %llu (chains %llu rest %llu) solve %llu loop %llu | kkt fill %llu ldlt %llu transpose+lists %llu | D %llu backward %llu (executed blocks %llu: %llu)\n", iter, cflag[0],
           g_bigprof[0], g_bigprof[1], g_bigprof[2], g_bigprof[3], g_bigprof[5], wall_clock64() - t0_ticks, g_bigprof[8], g_bigprof[9], g_bigprof[10], g_bigprof[4], g_bigprof[7], g_bigprof[11], g_bigprof[6]);
    for (int i = 0; i < 12; ++i) g_bigprof[i] = 0;
  }
#endif
  // ---- polish :92-204, :515-539 (on the scaled iterate; a failed factorisation leaves the ADMM solution) ----
  if (ret_code == SFB_QP_OPTIMAL && kp.polish) {
    const double eps = DBL_EPSILON;
    // active sets: lower indices first, then upper, each ascending (:113-123)
    int nl = 0, nu = 0;
    for (int c0 = 0; c0 < m; c0 += kWave) {
      const int i = c0 + lane;
      int a       = 0;
      if (i < m) {
        const double yi = ys[i];
        if (yi < -100 * eps && l[i] != -inf) a = 1;
        if (yi > 100 * eps && u[i] != inf) a = 2;
      }
      const unsigned long long bl = wave_ballot(a == 1), bu = wave_ballot(a == 2);
      if (i < m) {
        w.act[i] = (double)a;
        w.pos[i] = (a == 1) ? (double)(nl + __popcll(bl & lanemask_lt(lane))) : (double)(nu + __popcll(bu & lanemask_lt(lane)));
      }
      nl += __popcll(bl);
      nu += __popcll(bu);
    }
    wave_sync();
    const int na = nl + nu, K = n + na;
    // Hs: the symmetric K x K matrix of the residual; Hp = Hs + diag(delta, -delta) lower row-major for the LDLT (:159-177)
    double *Hs = w.Hs, *Hp = w.H;
    const BigMatGlobal Hpg{Hp, (size_t)K};
    auto polish_fill = [&](const auto Hm) {
      for (int e = lane; e < K * K; e += kWave) Hs[e] = 0.0;
      for (int e = lane; e < K * K; e += kWave) Hp[e] = 0.0;
      wave_sync();
      for (int i = 0; i < n; ++i)
        for (int j = i + lane; j < n; j += kWave) {
          const double v           = c * w.sx[i] * PM(i, j) * w.sx[j];  // :161 upper entry (i, j)
          Hs[(size_t)i * K + j]    = v;
          Hs[(size_t)j * K + i]    = v;
          Hm.at(j, i)              = v;
        }
      for (int e = lane; e < m * n; e += kWave) {
        const int row = e % m, j = e / m;
        const int a   = (int)w.act[row];
        if (a != 0) {
          const int col  = n + (int)w.pos[row] + (a == 2 ? nl : 0);
          const double v = w.sy[row] * A[e] * w.sx[j];  // :163
          Hs[(size_t)j * K + col] = v;
          Hs[(size_t)col * K + j] = v;
          Hm.at(col, j)           = v;
        }
      }
      wave_sync();
      for (int i = lane; i < n; i += kWave) Hm.at(i, i) += kp.delta;
      for (int a = lane; a < na; a += kWave) Hm.at(n + a, n + a) -= kp.delta;
    };
    polish_fill(Hpg);
    // h (:179-182) and t = 0
    for (int j = lane; j < n; j += kWave) w.hx[j] = -c * (w.sx[j] * q[j]);
    for (int i = lane; i < m; i += kWave) {
      const int a = (int)w.act[i];
      if (a == 1) w.Ax[(int)w.pos[i]] = w.sy[i] * l[i];
      else if (a == 2) w.Ax[nl + (int)w.pos[i]] = w.sy[i] * u[i];
    }
    for (int e = lane; e < K; e += kWave) aux[e] = 0.0;  // aux = t of the refinement
    wave_sync();
    if (big_ldlt_factor(K, Hpg, perm, temp, reinterpret_cast<int *>(t), lane)) {
      big_transpose<RB>(K, Hp, K, w.LT, w.Dg, fnz, bnz, dcache ? dblk : nullptr, LDg, cidx, cval, cflag, rcap, lane);
      for (uint32_t it = 0; it != kp.polish_iter; ++it) {  // :193-195  t += Hp^-1 (h - Hs t)
        for (int i = lane; i < K; i += kWave) {
          double s = 0.0;
          for (int j = 0; j < K; ++j) s = fma(Hs[(size_t)j * K + i], aux[j], s);  // (Hs is symmetric: column i == row i)
          temp[i] = ((i < n) ? w.hx[i] : w.Ax[i - n]) - s;
        }
        wave_sync();
        big_solve<RB>(K, Hp, w.LT, w.Dg, K, perm, fnz, bnz, dcache ? dblk : nullptr, LDg, cidx, cval, cflag, t, temp, lane);
        for (int i = lane; i < K; i += kWave) aux[i] += temp[i];
        wave_sync();
      }
      for (int j = lane; j < n; j += kWave) xs[j] = aux[j];  // :199
      for (int i = lane; i < m; i += kWave) {                // :200-201
        const int a = (int)w.act[i];
        if (a == 1) ys[i] = aux[n + (int)w.pos[i]];
        else if (a == 2) ys[i] = aux[n + nl + (int)w.pos[i]];
      }
    }
    wave_sync();
  }

  // ---- un-scale and report :544-548 ----
  double *ox = g.x + b * (size_t)n, *oy = g.y + b * (size_t)m;
  for (int j = lane; j < n; j += kWave) {
    const double v = w.sx[j] * xs[j];
    ox[j] = v;
    t[j]  = v;
  }
  for (int i = lane; i < m; i += kWave) oy[i] = w.sy[i] * ys[i] / c;
  wave_sync();
  if (g.obj != nullptr) {
    for (int i = lane; i < n; i += kWave) {
      double s = 0.0;
      for (int j = 0; j < n; ++j) s = fma(0.5 * PM(i, j), t[j], s);
      temp[i] = s + q[i];
    }
    wave_sync();
    if (lane == 0) {
      double o = 0.0;
      for (int i = 0; i < n; ++i) o = fma(t[i], temp[i], o);
      g.obj[b] = o;
    }
  }
  if (lane == 0) {
    g.code[b] = (ret_code >= 0) ? ret_code : SFB_QP_MAX_ITERATIONS;
    if (g.iter != nullptr) g.iter[b] = iter;
  }
#undef PM
#undef AM
}

hipError_t qp_dense_big_launch(const DenseKernelParams &kp, int64_t batch, const QpBatch &g, double *workspace, hipStream_t stream)
{
  const int k = kp.n + kp.m;
  if (k > kBigMaxK || k <= kDenseMidMaxK) return hipErrorInvalidValue;  // (up to 128: qp_dense_mid.hip, everything on chip)
  const BigLds cfg = big_lds_config(kp.n, kp.m, batch);
  const size_t lds = cfg.bytes;
  const int rcap = cfg.rcap, dck = cfg.diag_cache ? kDiagCacheK : 0;
  const size_t wsd = qp_dense_big_ws_doubles(kp.n, kp.m);
  const dim3 grid((unsigned)batch), block(kWave);
  const int rb = (k + kWave - 1) / kWave;
  if (rb <= 4) hipLaunchKernelGGL((qp_dense_big_kernel<4>), grid, block, lds, stream, kp, g, workspace, wsd, rcap, dck);
  else if (rb <= 8) hipLaunchKernelGGL((qp_dense_big_kernel<8>), grid, block, lds, stream, kp, g, workspace, wsd, rcap, dck);
  else hipLaunchKernelGGL((qp_dense_big_kernel<kRB>), grid, block, lds, stream, kp, g, workspace, wsd, rcap, dck);
  return hipGetLastError();
}

}  // namespace sfb
