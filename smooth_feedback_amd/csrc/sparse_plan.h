// Symbolic analysis ("plan") of a batch of sparse QPs that share one sparsity pattern.
//
// Replaces QPSolver<QuadraticProgramSparse<double>>::analyze (qp_solver.hpp:297-338) and the
// analyzePattern half of Eigen::SimplicialLDLT (qp_solver.hpp:424): builds the permuted KKT
// pattern [P+sI, A'; A, -1/rho] (:382-395), a fill-reducing elimination order (own minimum-degree
// heuristic, or a caller-supplied permutation -- Eigen's AMD is not available), the pattern of L
// (elimination-tree reach) and every index array the HIP kernel needs.  Host code only.
#pragma once
#include <cstdint>
#include <vector>

namespace sfb {

#ifndef SFB_KKT_KINDS
#define SFB_KKT_KINDS
enum { K_P = 0, K_A = 1, K_SIGMA = 2, K_RHO = 3 };
#endif

struct SparsePlanHost {
  int n = 0, m = 0, k = 0, nnzP = 0, nnzA = 0, nnzK = 0, nnzL = 0;
  std::vector<int32_t> Pp, Pi, Pcol;           // P: CSC as stored, + column of every entry
  std::vector<int32_t> Ap, Aj, Arow;           // A: CSR, + row of every entry
  std::vector<int32_t> Acp, Aci, Acpos;        // CSC view of A (rows ascending, position in A values)
  std::vector<int32_t> Prp, Prj, Prpos;        // CSR view of P
  std::vector<int32_t> Sp, Sj, Spos;           // symmetric view of triu(P) (selfadjointView<Upper>)
  std::vector<int32_t> perm, pinv;             // perm[new] = old
  std::vector<int32_t> Kp, Ki, Kkind, Kidx;    // permuted KKT, lower CSC
  std::vector<int32_t> Lp, Li;                 // strictly-lower pattern of L (col-major, rows ascending)
  std::vector<int32_t> Rp, Rk, Rpos;           // row structure of L
  std::vector<int32_t> Rlen;                   // Lp[Rk+1] - Rpos: length of the source-column suffix
  // Packed sweep schedules.  A sweep is a sequence of STEPS (= units) of 128 independent slots; slot = one
  // update   t[tgt] = fma(-L, t[piv], t[tgt]).
  // Every entry of L is one slot.  A slot depends on (i) the last update of its pivot (the pivot must be
  // final) and (ii) the previous update of its target in the sequential sweep order (so that the updates of
  // one target keep that order: two of them never share a step).  The arithmetic and its order per entry
  // are thus exactly those of the column-by-column (row-by-row) loop.  Steps are filled by critical-path
  // list scheduling: among the ready slots those with the longest chain of dependants go first (MPC
  // pattern: 384 + 336 steps for 2 x 41 030 entries; critical paths 285 / 218, width bound 321).
  //   xmap[q]  : position in the column-major values of L feeding slot q, or -1 (padding)
  //   xidx[q]  : (tgt | piv << 16) * idx_scale   (padding: both = k, a scratch slot of the LDS vector);
  //              idx_scale = 8 (byte offsets) when (k+1)*8 < 65536, else 1
  // Storage: unit u, lane l, slot s at [(u * 64 + l) * 2 + s] (one 16-byte value load and one 8-byte index
  // load per lane and unit).  Unit counts are multiples of kSweepPad and the arrays carry kSweepPad extra
  // all-padding units so the kernel prefetches branch-free.
  static constexpr int kSweepPad = 16;
  std::vector<int32_t> fmap, fidx, bmap, bidx;
  int funits = 0, bunits = 0, idx_scale = 1;
  // Right-looking factorisation schedule.  When column kk is final it updates, for every pair of its
  // rows (r_a >= r_b), the accumulator of entry (r_a, r_b) [the diagonal D(r_b) when a == b]:
  //   acc -= L(r_a,kk) * (L(r_b,kk) * D(kk)).
  // All updates issued by one column hit distinct accumulators and columns are processed in order,
  // so every accumulator receives its updates in ascending source order -- the same arithmetic as the
  // left-looking loop of the oracle -- while a step is 64 independent, fully packed slots.
  //   Kmap[p]  : accumulator index of KKT entry p   (accumulators: [L values (nnzL) | D (k) | scratch])
  //   rptr[kk] : first 64-slot step of column kk (rptr[k] = total steps);  per slot:
  //   rtgt[q]  : accumulator index (padding: nnzL + k),  rab[q] = a | b << 16 (entry numbers in column kk)
  std::vector<int32_t> Kmap, rptr, rtgt, rab;
  int rsteps = 0, maxcol = 0;
  // Supernodes: runs of consecutive columns j0..j1 with struct(j) = {j+1} u struct(j+1).  All columns of a
  // supernode update the SAME trailing accumulators (pairs of rows of struct(j1)), so the kernel eliminates
  // the supernode's own (dense) panel in LDS and then applies its w rank-1 updates to every trailing
  // accumulator with ONE read-modify-write (sources still in ascending order -> same bits), using the
  // schedule of column j1.  Widths are capped so that the panel and its multipliers, 2 w (cnt(j0)+1)
  // doubles, fit the kernel's LDS scratch of lds_doubles.
  //   snptr[s] : first column of supernode s (snptr[nsn] = k)
  //   poff[s]  : start of the supernode's panel map;  pmap[poff[s] + jj * R + r] = accumulator index of panel
  //              entry (row r, column jj) for r >= jj (r == jj: D), else the scratch accumulator nnzL + k
  std::vector<int32_t> snptr, poff, pmap;
  int nsn = 0, lds_doubles = 0;
};

// ordering: 0 = natural, 1 = minimum degree (default); user_perm (k entries, new->old) overrides.
// stage (k entries, nullable): constrained minimum degree -- unknowns of a lower stage are
// eliminated before any unknown of a higher stage (e.g. interval separators of an MPC horizon last:
// shorter elimination tree, less fill).  Returns false on malformed input (msg set).
bool build_sparse_plan(int n, int m, const int32_t *Pp, const int32_t *Pi, const int32_t *Ap, const int32_t *Aj,
                       int ordering, const int32_t *user_perm, const int32_t *stage, SparsePlanHost &out,
                       const char **msg);

}  // namespace sfb
