#include "sparse_plan.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <iterator>
#include <array>
#include <numeric>
#include <tuple>

namespace sfb {

namespace {

// transpose a compressed pattern: `outer_ptr` over `nouter` slices with `inner` indices < ninner.
// Output compressed by inner index, entries ordered by ascending outer index, with positions.
void transpose_pattern(int nouter, int ninner, const std::vector<int32_t> &outer_ptr,
                       const std::vector<int32_t> &inner, std::vector<int32_t> &tp, std::vector<int32_t> &ti,
                       std::vector<int32_t> &tpos)
{
  const int nnz = outer_ptr[nouter];
  tp.assign(ninner + 1, 0);
  ti.resize(nnz);
  tpos.resize(nnz);
  for (int p = 0; p < nnz; ++p) tp[inner[p] + 1]++;
  for (int i = 0; i < ninner; ++i) tp[i + 1] += tp[i];
  std::vector<int32_t> fill(ninner, 0);
  for (int o = 0; o < nouter; ++o)
    for (int p = outer_ptr[o]; p < outer_ptr[o + 1]; ++p) {
      const int i       = inner[p];
      ti[tp[i] + fill[i]]   = o;
      tpos[tp[i] + fill[i]] = p;
      fill[i]++;
    }
}

// Minimum-degree ordering on the symmetric KKT graph with explicit fill (bitset adjacency).
// k is at most a few thousand for the problems of this path; O(k^2 * k/64) is fine on the host.
std::vector<int32_t> min_degree_order(int k, const std::vector<std::pair<int, int>> &edges, const int32_t *stage)
{
  const int W = (k + 63) / 64;
  std::vector<uint64_t> adj((size_t)k * W, 0);
  auto set  = [&](int a, int b) { adj[(size_t)a * W + (b >> 6)] |= (1ull << (b & 63)); };
  for (auto [a, b] : edges)
    if (a != b) { set(a, b); set(b, a); }
  std::vector<int> deg(k);
  std::vector<char> done(k, 0);
  auto degree = [&](int a) {
    int d = 0;
    for (int w = 0; w < W; ++w) d += __builtin_popcountll(adj[(size_t)a * W + w]);
    return d;
  };
  for (int a = 0; a < k; ++a) deg[a] = degree(a);
  std::vector<int32_t> order;
  order.reserve(k);
  std::vector<int> nb;
  for (int step = 0; step < k; ++step) {
    int best = -1;  // lowest stage first, then minimum degree, ties: lowest index
    for (int a = 0; a < k; ++a) {
      if (done[a]) continue;
      if (best < 0) { best = a; continue; }
      const int sa = stage ? stage[a] : 0, sb = stage ? stage[best] : 0;
      if (sa < sb || (sa == sb && deg[a] < deg[best])) best = a;
    }
    order.push_back(best);
    done[best] = 1;
    nb.clear();
    for (int w = 0; w < W; ++w) {
      uint64_t bits = adj[(size_t)best * W + w];
      while (bits) {
        const int b = (w << 6) + __builtin_ctzll(bits);
        bits &= bits - 1;
        nb.push_back(b);
      }
    }
    // eliminate: neighbours become a clique, `best` leaves the graph
    for (int a : nb) {
      uint64_t *ra = &adj[(size_t)a * W];
      const uint64_t *rb = &adj[(size_t)best * W];
      for (int w = 0; w < W; ++w) ra[w] |= rb[w];
      ra[a >> 6] &= ~(1ull << (a & 63));
      ra[best >> 6] &= ~(1ull << (best & 63));
    }
    for (int a : nb) deg[a] = degree(a);
  }
  return order;
}

}  // namespace

bool build_sparse_plan(int n, int m, const int32_t *Pp, const int32_t *Pi, const int32_t *Ap, const int32_t *Aj,
                       int ordering, const int32_t *user_perm, const int32_t *stage, SparsePlanHost &o,
                       const char **msg)
{
  static const char *ok = "";
  *msg = ok;
  if (n < 1 || m < 1 || !Pp || !Ap) { *msg = "bad sizes / NULL pattern"; return false; }
  const int k = n + m;
  o = SparsePlanHost();
  o.n = n; o.m = m; o.k = k;
  o.nnzP = Pp[n]; o.nnzA = Ap[m];
  if (Pp[0] != 0 || Ap[0] != 0 || o.nnzP < 0 || o.nnzA < 0) { *msg = "pattern pointers must start at 0"; return false; }
  if ((o.nnzP > 0 && !Pi) || (o.nnzA > 0 && !Aj)) { *msg = "NULL index array"; return false; }
  o.Pp.assign(Pp, Pp + n + 1); o.Pi.assign(Pi, Pi + o.nnzP);
  o.Ap.assign(Ap, Ap + m + 1); o.Aj.assign(Aj, Aj + o.nnzA);
  o.Pcol.resize(o.nnzP); o.Arow.resize(o.nnzA);
  for (int c = 0; c < n; ++c) {
    if (Pp[c + 1] < Pp[c]) { *msg = "P column pointers not monotone"; return false; }
    for (int p = Pp[c]; p < Pp[c + 1]; ++p) {
      if (Pi[p] < 0 || Pi[p] >= n) { *msg = "P row index out of range"; return false; }
      if (p > Pp[c] && Pi[p] <= Pi[p - 1]) { *msg = "P row indices must be strictly ascending per column"; return false; }
      o.Pcol[p] = c;
    }
  }
  for (int r = 0; r < m; ++r) {
    if (Ap[r + 1] < Ap[r]) { *msg = "A row pointers not monotone"; return false; }
    for (int p = Ap[r]; p < Ap[r + 1]; ++p) {
      if (Aj[p] < 0 || Aj[p] >= n) { *msg = "A column index out of range"; return false; }
      if (p > Ap[r] && Aj[p] <= Aj[p - 1]) { *msg = "A column indices must be strictly ascending per row"; return false; }
      o.Arow[p] = r;
    }
  }
  transpose_pattern(m, n, o.Ap, o.Aj, o.Acp, o.Aci, o.Acpos);
  transpose_pattern(n, n, o.Pp, o.Pi, o.Prp, o.Prj, o.Prpos);

  // symmetric view of the upper-stored part of P: row i -> mirrored entries (cols < i), then upper ones
  {
    o.Sp.assign(n + 1, 0);
    for (int c = 0; c < n; ++c)
      for (int p = Pp[c]; p < Pp[c + 1]; ++p) {
        const int r = Pi[p];
        if (c >= r) { o.Sp[r + 1]++; if (r != c) o.Sp[c + 1]++; }
      }
    for (int i = 0; i < n; ++i) o.Sp[i + 1] += o.Sp[i];
    o.Sj.resize(o.Sp[n]); o.Spos.resize(o.Sp[n]);
    std::vector<int32_t> fill(n, 0);
    for (int c = 0; c < n; ++c)
      for (int p = Pp[c]; p < Pp[c + 1]; ++p)
        if (Pi[p] < c) { o.Sj[o.Sp[c] + fill[c]] = Pi[p]; o.Spos[o.Sp[c] + fill[c]] = p; fill[c]++; }
    for (int c = 0; c < n; ++c)
      for (int p = Pp[c]; p < Pp[c + 1]; ++p) {
        const int r = Pi[p];
        if (c >= r) { o.Sj[o.Sp[r] + fill[r]] = c; o.Spos[o.Sp[r] + fill[r]] = p; fill[r]++; }
      }
  }

  // KKT entries in ORIGINAL indices (upper triangle, qp_solver.hpp:382-395)
  struct Ent { int r, c, kind, idx; };
  std::vector<Ent> ents;
  ents.reserve(o.nnzP + o.nnzA + k);
  std::vector<char> hasdiag(n, 0);
  for (int c = 0; c < n; ++c)
    for (int p = Pp[c]; p < Pp[c + 1]; ++p)
      if (c >= Pi[p]) {
        ents.push_back({Pi[p], c, K_P, p});
        if (Pi[p] == c) hasdiag[c] = 1;
      }
  for (int v = 0; v < n; ++v)
    if (!hasdiag[v]) ents.push_back({v, v, K_SIGMA, v});
  for (int r = 0; r < m; ++r) {
    for (int p = Ap[r]; p < Ap[r + 1]; ++p) ents.push_back({Aj[p], n + r, K_A, p});
    ents.push_back({n + r, n + r, K_RHO, r});
  }

  // elimination order
  o.perm.resize(k);
  if (user_perm) {
    std::vector<char> seen(k, 0);
    for (int i = 0; i < k; ++i) {
      if (user_perm[i] < 0 || user_perm[i] >= k || seen[user_perm[i]]) { *msg = "user_perm is not a permutation"; return false; }
      seen[user_perm[i]] = 1;
      o.perm[i] = user_perm[i];
    }
  } else if (ordering == 0) {
    std::iota(o.perm.begin(), o.perm.end(), 0);
  } else {
    std::vector<std::pair<int, int>> edges;
    edges.reserve(ents.size());
    for (const Ent &e : ents) edges.emplace_back(e.r, e.c);
    o.perm = min_degree_order(k, edges, stage);
  }
  o.pinv.resize(k);
  for (int i = 0; i < k; ++i) o.pinv[o.perm[i]] = i;

  // permuted lower CSC
  struct PE { int i, j, kind, idx; };
  std::vector<PE> pe;
  pe.reserve(ents.size());
  for (const Ent &e : ents) {
    const int a = o.pinv[e.r], b = o.pinv[e.c];
    pe.push_back({std::max(a, b), std::min(a, b), e.kind, e.idx});
  }
  std::sort(pe.begin(), pe.end(), [](const PE &x, const PE &y) { return std::tie(x.j, x.i) < std::tie(y.j, y.i); });
  o.nnzK = (int)pe.size();
  o.Kp.assign(k + 1, 0); o.Ki.resize(o.nnzK); o.Kkind.resize(o.nnzK); o.Kidx.resize(o.nnzK);
  for (int t = 0; t < o.nnzK; ++t) {
    o.Kp[pe[t].j + 1]++;
    o.Ki[t] = pe[t].i; o.Kkind[t] = pe[t].kind; o.Kidx[t] = pe[t].idx;
  }
  for (int j = 0; j < k; ++j) o.Kp[j + 1] += o.Kp[j];

  // rows of the lower form (strict): row r -> columns j < r
  std::vector<int32_t> rp(k + 1, 0), rj;
  for (const PE &e : pe)
    if (e.i != e.j) rp[e.i + 1]++;
  for (int r = 0; r < k; ++r) rp[r + 1] += rp[r];
  rj.resize(rp[k]);
  {
    std::vector<int32_t> fill(k, 0);
    for (const PE &e : pe)
      if (e.i != e.j) rj[rp[e.i] + fill[e.i]++] = e.j;
  }
  // elimination tree + pattern of L by row reach
  std::vector<int32_t> parent(k, -1), flag(k, -1), lnz(k, 0);
  for (int r = 0; r < k; ++r) {
    flag[r] = r;
    for (int p = rp[r]; p < rp[r + 1]; ++p)
      for (int i = rj[p]; flag[i] != r; i = parent[i]) {
        if (parent[i] == -1) parent[i] = r;
        lnz[i]++;
        flag[i] = r;
      }
  }
  o.Lp.assign(k + 1, 0);
  for (int j = 0; j < k; ++j) o.Lp[j + 1] = o.Lp[j] + lnz[j];
  o.nnzL = o.Lp[k];
  o.Li.resize(o.nnzL);
  {
    std::vector<int32_t> fill(k, 0);
    std::fill(flag.begin(), flag.end(), -1);
    for (int r = 0; r < k; ++r) {
      flag[r] = r;
      for (int p = rp[r]; p < rp[r + 1]; ++p)
        for (int i = rj[p]; flag[i] != r; i = parent[i]) {
          o.Li[o.Lp[i] + fill[i]++] = r;
          flag[i] = r;
        }
    }
  }
  transpose_pattern(k, k, o.Lp, o.Li, o.Rp, o.Rk, o.Rpos);
  o.Rlen.resize(o.nnzL);
  for (int t = 0; t < o.nnzL; ++t) o.Rlen[t] = o.Lp[o.Rk[t] + 1] - o.Rpos[t];

  // right-looking factorisation schedule (see sparse_plan.h)
  {
    std::vector<int32_t> where(k, -1);  // row -> position in the column currently scattered
    auto pos_in_col = [&](int col, int row) {  // binary search: rows ascending
      const int32_t *b = o.Li.data() + o.Lp[col], *e = o.Li.data() + o.Lp[col + 1];
      const int32_t *it = std::lower_bound(b, e, row);
      return (it != e && *it == row) ? (int)(it - o.Li.data()) : -1;
    };
    o.Kmap.resize(o.nnzK);
    for (int j = 0; j < k; ++j)
      for (int p = o.Kp[j]; p < o.Kp[j + 1]; ++p) {
        const int i = o.Ki[p];
        o.Kmap[p]   = (i == j) ? o.nnzL + j : pos_in_col(j, i);
        if (o.Kmap[p] < 0) { *msg = "internal: KKT entry outside the pattern of L"; return false; }
      }
    // descriptor of every KKT entry for the kernel's batched fill: {kind, idx, r, c} with the row / column of the
    // source entry resolved here (pattern only), + the accumulator it goes to; padded for branch-free batches
    o.Kdesc.assign((size_t)(o.nnzK + 64 * 8) * 4, 0);
    for (int p = 0; p < o.nnzK; ++p) {
      const int kind = o.Kkind[p], idx = o.Kidx[p];
      int r = 0, c = 0;
      if (kind == K_P) { r = o.Pi[idx]; c = o.Pcol[idx]; }
      else if (kind == K_A) { r = o.Arow[idx]; c = o.Aj[idx]; }
      o.Kdesc[4 * (size_t)p] = kind; o.Kdesc[4 * (size_t)p + 1] = idx; o.Kdesc[4 * (size_t)p + 2] = r; o.Kdesc[4 * (size_t)p + 3] = c;
    }
    for (size_t p = o.nnzK; p < (size_t)o.nnzK + 64 * 8; ++p) o.Kdesc[4 * p] = K_SIGMA;  // padding: constant, to scratch
    o.Kmap.resize((size_t)o.nnzK + 64 * 8, o.nnzL + k);
    o.maxcol = 0;
    for (int kk = 0; kk < k; ++kk) o.maxcol = std::max(o.maxcol, o.Lp[kk + 1] - o.Lp[kk]);
    if (o.maxcol >= (1 << 16)) { *msg = "column of L too long for the update encoding"; return false; }
    o.lds_doubles = std::max(k + 2, 2 * o.maxcol + 4);

    // ---- relaxed supernodes (see sparse_plan.h) ----
    // Greedy grouping of consecutive columns: the panel rows are the columns themselves plus the union U of
    // their remaining row structures.  A column joins while the panel and its multipliers (2 w R doubles)
    // fit the LDS scratch, w <= 16, and the union grows by at most kRelax rows over the larger of the two
    // structures (explicit zeros cost LDS work, not HBM traffic).
    constexpr int kRelax = 4, kMaxWidth = 16;
    const int ZERO = o.nnzL + k + 1, SCRATCH = o.nnzL + k;  // accumulator indices: always-zero entry, padding sink
    o.snptr.clear(); o.snR.clear(); o.poff.clear(); o.pmap.clear(); o.rptr.clear(); o.rtgt.clear(); o.rab.clear();
    o.rptr.push_back(0);
    int j0 = 0;
    while (j0 < k) {
      std::vector<int32_t> U(o.Li.begin() + o.Lp[j0], o.Li.begin() + o.Lp[j0 + 1]);  // sorted rows outside the panel
      int w = 1;
      while (j0 + w < k && w < kMaxWidth) {
        const int jn = j0 + w;
        std::vector<int32_t> Un;
        std::set_union(U.begin(), U.end(), o.Li.begin() + o.Lp[jn], o.Li.begin() + o.Lp[jn + 1], std::back_inserter(Un));
        Un.erase(std::remove_if(Un.begin(), Un.end(), [&](int32_t r) { return r <= jn; }), Un.end());
        const int Rn = (w + 1) + (int)Un.size();
        if (2 * (w + 1) * Rn > o.lds_doubles) break;
        int ubase = 0;
        for (int32_t r : U) ubase += (r != jn);
        const int base = std::max(ubase, o.Lp[jn + 1] - o.Lp[jn]);
        if ((int)Un.size() - base > kRelax) break;
        U.swap(Un);
        ++w;
      }
      const int nu = (int)U.size(), R = w + nu;
      o.snptr.push_back(j0);
      o.snR.push_back(R);
      // panel map: entry (row r, member jj), r >= jj; rows of L that are not in the member's structure are zeros
      o.poff.push_back((int)o.pmap.size());
      for (int jj = 0; jj < w; ++jj)
        for (int r = 0; r < R; ++r) {
          int src = SCRATCH;
          if (r == jj) src = o.nnzL + j0 + jj;
          else if (r > jj) {
            const int g = (r < w) ? j0 + r : U[r - w];
            const int pos = pos_in_col(j0 + jj, g);
            src = (pos >= 0) ? pos : ZERO;
          }
          o.pmap.push_back(src);
        }
      // trailing schedule: pairs (a >= b) of U whose accumulator exists and is touched by some member
      std::vector<std::array<int32_t, 2>> slots;
      for (int bq = 0; bq < nu; ++bq)
        for (int aq = bq; aq < nu; ++aq) {
          const int rb = U[bq], ra = U[aq];
          int tgt;
          if (aq == bq) tgt = o.nnzL + rb;
          else {
            tgt = pos_in_col(rb, ra);
            if (tgt < 0) continue;
          }
          bool touched = false;
          for (int jj = 0; jj < w && !touched; ++jj)
            touched = pos_in_col(j0 + jj, ra) >= 0 && pos_in_col(j0 + jj, rb) >= 0;
          if (!touched) continue;
          slots.push_back({tgt, aq | (bq << 16)});
        }
      const int steps = (int)((slots.size() + 63) / 64);
      const size_t q0 = o.rtgt.size();
      o.rtgt.resize(q0 + (size_t)steps * 64, SCRATCH);
      o.rab.resize(q0 + (size_t)steps * 64, 0);
      for (size_t e = 0; e < slots.size(); ++e) { o.rtgt[q0 + e] = slots[e][0]; o.rab[q0 + e] = slots[e][1]; }
      o.rptr.push_back(o.rptr.back() + steps);
      j0 += w;
    }
    o.nsn = (int)o.snptr.size();
    if (const char *dbg = getenv("SFB_PLAN_DEBUG"); dbg && dbg[0] == '1') {  // diagnostics: supernode shapes
      int maxR = 0, over64 = 0;
      long long panel = 0;
      for (int sn = 0; sn < o.nsn; ++sn) {
        const int w = (sn + 1 < o.nsn ? o.snptr[sn + 1] : k) - o.snptr[sn], R = o.snR[sn];
        maxR = std::max(maxR, R);
        over64 += R > 64;
        panel += (long long)w * R;
        fprintf(stderr, "[sfb plan] supernode %d: columns %d..%d (w=%d) R=%d steps=%d\n", sn, o.snptr[sn], o.snptr[sn] + w - 1, w, R,
                o.rptr[sn + 1] - o.rptr[sn]);
      }
      fprintf(stderr, "[sfb plan] %d supernodes, max R %d, %d with R > 64, panel entries %lld, trailing steps %d\n", o.nsn, maxR,
              over64, panel, o.rptr.back());
    }
    o.snptr.push_back(k);
    o.poff.push_back((int)o.pmap.size());
    o.rsteps = o.rptr.back();
    for (int pad = 0; pad < 64 * SparsePlanHost::kSweepPad; ++pad) {  // branch-free batched reads past the end
      o.pmap.push_back(SCRATCH);
      o.rtgt.push_back(SCRATCH);
      o.rab.push_back(0);
    }
  }

  // packed sweep schedules (see sparse_plan.h)
  if (k + 1 >= (1 << 16)) { *msg = "n+m too large for the packed sweep encoding (max 65534)"; return false; }
  o.idx_scale = ((k + 1) * 8 < (1 << 16)) ? 8 : 1;
  auto build = [&](bool forward, std::vector<int32_t> &xmap, std::vector<int32_t> &xidx, int &units,
                   std::vector<int32_t> &xmask, int &full0, int &full1) {
    const int cap = 128;  // slots per dependent step
    // Critical-path list scheduling.  Slot = one entry of L, enumerated in the sequential sweep order.  It
    // depends on (a) the previous update of its target (per-target order = the sequential one) and (b) the
    // last update of its pivot (the pivot must be final).  Steps are filled with the ready slots of greatest
    // height (longest chain of dependants), which keeps the schedule close to max(critical path, slots/cap).
    struct Slot { int32_t pos, tgt, piv; };
    std::vector<Slot> sl;
    sl.reserve(o.nnzL);
    std::vector<int32_t> col_first(k + 1, 0);  // slots with pivot j (in order of t): [col_first[t], col_first[t+1])
    for (int t = 0; t < k; ++t) {
      const int j = forward ? t : k - 1 - t;
      col_first[t] = (int32_t)sl.size();
      const int p0 = forward ? o.Lp[j] : o.Rp[j], p1 = forward ? o.Lp[j + 1] : o.Rp[j + 1];
      for (int p = p0; p < p1; ++p) sl.push_back({forward ? p : o.Rpos[p], forward ? o.Li[p] : o.Rk[p], j});
    }
    col_first[k] = (int32_t)sl.size();
    const int ns = (int)sl.size();
    auto tof = [&](int j) { return forward ? j : k - 1 - j; };  // position of pivot j in the processing order
    std::vector<int32_t> prev_same(ns, -1), next_same(ns, -1), last_upd(k, -1);
    for (int q = 0; q < ns; ++q) {
      const int tg = sl[q].tgt;
      if (last_upd[tg] >= 0) { prev_same[q] = last_upd[tg]; next_same[last_upd[tg]] = q; }
      last_upd[tg] = q;
    }
    // heights, reverse order (every dependant of a slot has a larger number)
    std::vector<int32_t> height(ns, 1), colh(k, 0);
    for (int q = ns - 1; q >= 0; --q) {
      int h = 0;
      if (next_same[q] >= 0) h = height[next_same[q]];
      else h = std::max(h, (int)colh[sl[q].tgt]);  // last update of its target: the target's own slots wait for it
      height[q] = h + 1;
      colh[sl[q].piv] = std::max(colh[sl[q].piv], height[q]);
    }
    std::vector<int32_t> npred(ns, 0);
    for (int q = 0; q < ns; ++q) npred[q] = (prev_same[q] >= 0 ? 1 : 0) + (last_upd[sl[q].piv] >= 0 ? 1 : 0);
    std::vector<std::vector<std::array<int32_t, 3>>> slots;  // per step: (pos, tgt, piv)
    std::vector<std::pair<int32_t, int32_t>> heap;          // (height, -slot): ready slots
    std::vector<int32_t> arriving;                           // become ready for the NEXT step
    for (int q = 0; q < ns; ++q)
      if (npred[q] == 0) heap.emplace_back(height[q], -q);
    std::make_heap(heap.begin(), heap.end());
    int done = 0;
    while (done < ns) {
      slots.emplace_back();
      arriving.clear();
      for (int c = 0; c < cap && !heap.empty(); ++c) {
        std::pop_heap(heap.begin(), heap.end());
        const int q = -heap.back().second;
        heap.pop_back();
        slots.back().push_back({sl[q].pos, sl[q].tgt, sl[q].piv});
        ++done;
        if (next_same[q] >= 0) {
          if (--npred[next_same[q]] == 0) arriving.push_back(next_same[q]);
        } else {  // target final: release its own slots
          const int tt = tof(sl[q].tgt);
          for (int r = col_first[tt]; r < col_first[tt + 1]; ++r)
            if (--npred[r] == 0) arriving.push_back(r);
        }
      }
      for (int q : arriving) {
        heap.emplace_back(height[q], -q);
        std::push_heap(heap.begin(), heap.end());
      }
    }
    const int steps = (int)slots.size();
    if (const char *dbg = getenv("SFB_PLAN_DEBUG"); dbg && dbg[0] == '1') {  // diagnostics: fill of the sweep units
      fprintf(stderr, "[sfb plan] %s sweep: %d units, slots per unit:", forward ? "forward" : "backward", steps);
      for (int s = 0; s < steps; ++s) fprintf(stderr, " %d", (int)slots[s].size());
      fprintf(stderr, "\n");
    }
    units           = steps;
    units = ((units + SparsePlanHost::kSweepPad - 1) / SparsePlanHost::kSweepPad) * SparsePlanHost::kSweepPad;
    const size_t total = (size_t)(units + SparsePlanHost::kSweepPad) * 128;
    xmap.assign(total, -1);
    // indices are stored as BYTE offsets into the work vector when they fit 16 bits (saves the kernel four
    // shifts per unit on the dependent chain of a lone wave), else as element indices
    const int sc = o.idx_scale;
    xidx.assign(total, (k * sc) | ((k * sc) << 16));
    // lane masks of the units' VALUE loads (stored as the shift 64 - lanes): the slots of a step occupy the leading
    // ceil(c/2) lanes, two per lane, so a unit with c slots touches only those lanes' 16 bytes -- the
    // other lanes are masked off and their cache lines never leave HBM (their slots are padding: target = pivot =
    // the scratch entry k, whatever value the register holds).  full0/full1: the longest run of completely filled
    // units, cut to multiples of 8; the kernel uses unmasked loads there.
    xmask.assign((size_t)units + 2 * SparsePlanHost::kSweepPad, 63);  // padding units: lane 0 only
    for (int s = 0; s < steps; ++s) {
      const int lanes = std::max(1, ((int)slots[s].size() + 1) / 2);
      xmask[s]        = 64 - lanes;  // exec = all ones >> (64 - lanes)
      for (size_t e = 0; e < slots[s].size(); ++e) {
        // step s -> unit s, lane e % lanes, slot e / lanes: the leading `lanes` lanes carry everything, and slots
        // that follow each other in the sweep order (neighbouring rows of one column) sit in neighbouring lanes
        // (LDS banks); a full unit is lane e % 64, slot e / 64
        const size_t q = ((size_t)s * 64 + (e % (size_t)lanes)) * 2 + (e / (size_t)lanes);
        xmap[q] = slots[s][e][0];
        xidx[q] = (slots[s][e][1] * sc) | ((slots[s][e][2] * sc) << 16);
      }
    }
    full0 = full1 = 0;
    for (int s = 0; s < steps;) {
      if ((int)slots[s].size() != cap) { ++s; continue; }
      int e = s;
      while (e < steps && (int)slots[e].size() == cap) ++e;
      const int a = ((s + 7) / 8) * 8, b = (e / 8) * 8;
      if (b - a > full1 - full0) { full0 = a; full1 = b; }
      s = e;
    }
  };
  build(true, o.fmap, o.fidx, o.funits, o.fmask, o.ffull0, o.ffull1);
  build(false, o.bmap, o.bidx, o.bunits, o.bmask, o.bfull0, o.bfull1);

  return true;
}

}  // namespace sfb
