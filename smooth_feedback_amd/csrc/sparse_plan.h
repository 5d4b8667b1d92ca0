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
  // load per lane and unit); the c slots of a step occupy the leading ceil(c/2) lanes (slot e: lane e % lanes, slot e / lanes).  Unit counts are multiples of kSweepPad and the arrays carry kSweepPad extra
  // all-padding units so the kernel prefetches branch-free.
  static constexpr int kSweepPad = 16;
  //   xmask[u] : 64 - (leading lanes of unit u's value load that carry slots);
  //              [full0, full1) : run of completely filled units (multiples of 8), loaded unmasked
  std::vector<int32_t> fmap, fidx, bmap, bidx, fmask, bmask;
  int funits = 0, bunits = 0, idx_scale = 1, ffull0 = 0, ffull1 = 0, bfull0 = 0, bfull1 = 0;
  // Right-looking, supernodal factorisation schedule.  Accumulators: [L values (nnzL) | D (k) | scratch | zero];
  // Kmap[p] = accumulator of KKT entry p.  When a column j is final it updates, for every pair of its rows
  // (r_a >= r_b), the accumulator of entry (r_a, r_b) [the diagonal D(r_b) when a == b]:
  //   acc = fma(-L(r_a,j), L(r_b,j) * D(j), acc),
  // and every accumulator receives these updates in ascending order of j -- the arithmetic of the oracle's
  // left-looking loop.
  // RELAXED SUPERNODES: consecutive columns j0 .. j0+w-1 are grouped (greedily, see sparse_plan.cpp); the panel
  // of a group has R = w + |U| rows, the columns themselves and the union U of their remaining row structures
  // (entries of a member outside its own structure are explicit zeros: they stay exactly zero and contribute
  // exact zeros).  The kernel eliminates the panel in LDS and then gives every TRAILING accumulator (a pair of
  // rows of U) the w updates of the group with ONE read-modify-write, in ascending member order.  MPC pattern:
  // 207 groups and 81 k accumulator touches per factorisation instead of 1 480 columns and 711 k touches.
  //   snptr[s] : first column of group s (snptr[nsn] = k);   snR[s] : panel rows R
  //   poff[s]  : start of the panel map;  pmap[poff[s] + jj * R + r] = accumulator of panel entry (row r, member jj)
  //              for r >= jj (r == jj: D; "zero" where L has no entry), else the scratch accumulator
  //   rptr[s]  : first 64-slot step of the trailing schedule of group s (rptr[nsn] = rsteps);  per slot:
  //   rtgt[q]  : accumulator index (padding: scratch),  rab[q] = a | b << 16 (positions in U)
  // Group widths are capped so that the panel and its multipliers, 2 w R doubles, fit the kernel's LDS
  // scratch of lds_doubles.
  std::vector<int32_t> Kmap, rptr, rtgt, rab, snptr, snR, poff, pmap;
  std::vector<int32_t> Kdesc;  // per KKT entry {kind, idx, row, col of the source entry}; Kdesc and Kmap are padded by 512 entries
  int rsteps = 0, maxcol = 0, nsn = 0, lds_doubles = 0;
  // FACTORISATION NUMBERING.  All of the above (accumulator positions, supernodes, K entries) is in F, a postorder
  // of the elimination tree of `perm` (children ascending): subtrees are contiguous column ranges, consecutive
  // columns share structure (wider supernodes), and an accumulator receives its updates in ascending F order.
  // The sweeps stay in S (= perm): f2s[F column] = S column (Dinv is stored in S), the sweep maps point at F positions.
  std::vector<int32_t> f2s;
  // LDS-RESIDENT SUBTREES (sparse_plan.cpp): the columns are cut into segments, each either a subtree whose
  // accumulators fit the item's LDS ("lds": factorised on chip) or a run of top columns (accumulators in HBM).
  //   seg[10 s ..] = {first supernode, last + 1, first column, last + 1, lds, nL, accN = nL + columns,
  //                  first K entry, last + 1, first L entry}; LDS layout of a segment: [L entries | D | sink | zero | panel scratch]
  //   pmapL  : LDS offsets of the panel entries (parallel to pmap; 0 outside LDS segments)
  //   rsplit[s] : first HBM step of supernode s's trailing schedule; steps [rptr[s], rsplit[s]) target LDS offsets
  //   KmapL  : LDS offset of every K entry of an LDS segment;  KdescT / KmapT (nnzKT entries, padded): the K
  //            entries of the top columns;  ztop: {start, length} accumulator ranges the kernel zeroes in HBM
  static constexpr int kSegStride = 12;  // seg also carries {outside accumulators, start of their map} (unit engine)
  std::vector<int32_t> seg, pmapL, rsplit, KmapL, KdescT, KmapT, ztop;
  int nseg = 0, nnzKT = 0, nztop = 0;
  // UNIT ENGINE of the numeric factorisation (round 5; sparse_plan.cpp).  Instead of supernodal panels, the work of a
  // segment [c0, c1) of columns is a static schedule of UNITS of 128 independent slots of one kind,
  //   DM  (division):  L(r, j) = acc(r, j) / D(j)                                  words {t | d << 16, all ones}
  //   F   (update):    acc(a, b) = fma(-L(a, j), L(b, j) * D(j), acc(a, b))          words {t | a << 16, b | d << 16}
  // all operands being byte offsets into the segment's LDS working set
  //   [own L entries (nL) | own D (c1 - c0) | outside accumulators (nout) | sink | zero | one]:
  // the accumulators of later columns that the segment updates are fetched from the workspace when it starts
  // (uomap[omap0 + e] = accumulator index) and returned when it ends.  Every accumulator still receives its updates
  // in ascending order of the source column, each one the oracle's fma -- bit for bit the left-looking loop.
  // A closed segment (lds == 1: a whole subtree, nobody outside updates its accumulators) fills its KKT entries on
  // chip; an open one (a run of top columns) finds its accumulators, filled and partly updated, in the workspace.
  //   ustream : unit u, lane l, half h (slot h * 64 + l) at [(u * 64 + l) * 4 + 2 h ..+1]; padded by kSweepPad units
  //   utype[u]: 1 = DM, 0 = F;  seg[..] = {u0, u1, ...} the units of a segment
  // units == 0: some column does not fit the LDS on its own (or SFB_PLAN_UNITS=0): the supernodal engine runs.
  std::vector<int32_t> ustream, utype, uomap;
  int units = 0, nunits = 0;
};

// ordering: 0 = natural, 1 = minimum degree (default); user_perm (k entries, new->old) overrides.
// stage (k entries, nullable): constrained minimum degree -- unknowns of a lower stage are
// eliminated before any unknown of a higher stage (e.g. interval separators of an MPC horizon last:
// shorter elimination tree, less fill).  Returns false on malformed input (msg set).
// lds_hint > 0: the LDS (doubles) of the plan this one accompanies as the whole-pattern fallback (sparse_plan.cpp).
bool build_sparse_plan(int n, int m, const int32_t *Pp, const int32_t *Pi, const int32_t *Ap, const int32_t *Aj,
                       int ordering, const int32_t *user_perm, const int32_t *stage, SparsePlanHost &out,
                       const char **msg, int lds_hint = 0);

}  // namespace sfb
