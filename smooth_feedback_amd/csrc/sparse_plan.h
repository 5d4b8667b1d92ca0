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
  // Sweep schedules: chunks of <= 64 entries, descriptor = {start, pivot | count << 24}, padded with
  // empty chunks to a multiple of 64 plus one extra block (the kernel prefetches ahead branch-free).
  std::vector<int32_t> fdesc, bdesc;           // forward (columns ascending) / backward (rows descending)
  int fblocks = 0, bblocks = 0;                // number of 64-chunk blocks (without the extra block)
};

// ordering: 0 = natural, 1 = minimum degree (default); user_perm (k entries, new->old) overrides.
// Returns false on malformed input (msg set).
bool build_sparse_plan(int n, int m, const int32_t *Pp, const int32_t *Pi, const int32_t *Ap, const int32_t *Aj,
                       int ordering, const int32_t *user_perm, SparsePlanHost &out, const char **msg);

}  // namespace sfb
