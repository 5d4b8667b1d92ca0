#include "knobs.h"
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

// ---------------------------------------------------------------------------------------------------------------
// BANK-AWARE PLACEMENT of the slots of one sweep unit.  The slots of a unit are independent (two of them never share a
// target, and no pivot is a target of the same unit), so any slot may sit in any of the unit's 2 x lanes positions
// (lane, half) without changing a bit.  What the position does change is which LDS accesses are served together: per
// half the kernel issues ds_read_b64 t[piv], ds_read_b64 t[tgt] and ds_write_b64 t[tgt] for all lanes.  gfx950 serves
// an 8-byte read in two groups of 32 lanes with 64 four-byte banks -- two doubles whose indices agree mod 32 collide
// unless they are the same double (broadcast) -- and an 8-byte write in four groups of 16 lanes with 32 banks (indices
// mod 16).  Every extra distinct address on a bank costs the group one more LDS cycle, on the dependent chain of a lone
// wave.  Modelled cost of a unit = sum over the read groups of [max bank load of the targets + max bank load of the
// distinct pivots] + sum over the write groups of [max bank load of the targets].
struct UnitPlacer {
  struct S { int32_t tgt, piv; };
  int lanes = 0;
  std::vector<S> sl;
  std::vector<int> where;  // slot -> position (lane * 2 + half)

  // write group g = half * 4 + lane / 16 (8 of them), read group r = g / 2 = half * 2 + lane / 32
  int capacity(int wg) const
  {
    const int l0 = (wg & 3) * 16;
    return std::max(0, std::min(lanes, l0 + 16) - l0);
  }
  // bank occupancy of a placement, kept incrementally
  std::vector<int> pid;      // slot -> index of its pivot among the unit's distinct pivots
  std::vector<int> pbank;    // distinct pivot -> bank
  int np = 0;
  // bank loads with their maximum kept up to date in O(1) (histogram of the loads)
  template<int NBANK>
  struct Banks {
    int n[NBANK], hist[130], mx;
    void clear() { std::fill(n, n + NBANK, 0); std::fill(hist, hist + 130, 0); hist[0] = NBANK; mx = 0; }
    void inc(int b) { hist[n[b]]--; hist[++n[b]]++; if (n[b] > mx) mx = n[b]; }
    void dec(int b) { hist[n[b]]--; hist[--n[b]]++; while (mx > 0 && hist[mx] == 0) --mx; }
  };
  Banks<32> tc[4], pb[4];
  Banks<16> wc[8];
  std::vector<int> pc[4];    // per read group: slots per distinct pivot
  void reset()
  {
    for (auto &b : tc) b.clear();
    for (auto &b : pb) b.clear();
    for (auto &b : wc) b.clear();
    for (auto &v : pc) v.assign((size_t)np, 0);
  }
  void add(int e, int g, int d)  // d = +1 / -1
  {
    const int r = g >> 1;
    int &n = pc[r][pid[e]];
    if (d > 0) {
      tc[r].inc(sl[e].tgt & 31);
      wc[g].inc(sl[e].tgt & 15);
      if (n++ == 0) pb[r].inc(pbank[pid[e]]);
    } else {
      tc[r].dec(sl[e].tgt & 31);
      wc[g].dec(sl[e].tgt & 15);
      if (--n == 0) pb[r].dec(pbank[pid[e]]);
    }
  }
  int rcost(int r) const { return tc[r].mx + pb[r].mx; }
  int wcost(int g) const { return wc[g].mx; }
  int cost_of(int g, int h) const  // everything the write groups g and h take part in
  {
    int c = wcost(g) + wcost(h) + rcost(g >> 1);
    if ((h >> 1) != (g >> 1)) c += rcost(h >> 1);
    return c;
  }
  int total() const
  {
    int c = 0;
    for (int g = 0; g < 8; ++g) c += wcost(g);
    for (int r = 0; r < 4; ++r) c += rcost(r);
    return c;
  }
  bool bad(int e, int g) const  // on a bank that sets the cost of its group
  {
    const int r = g >> 1;
    const int t = tc[r].n[sl[e].tgt & 31], w = wc[g].n[sl[e].tgt & 15], p = pb[r].n[pbank[pid[e]]];
    return (t > 1 && t == tc[r].mx) || (w > 1 && w == wc[g].mx) || (p > 1 && p == pb[r].mx);
  }
  // returns {modelled cycles of the natural placement, of the chosen one}; `where` filled.  search: local search passes
  std::pair<int, int> place(const int search)
  {
    const int c = (int)sl.size();
    // distinct pivots
    pid.assign((size_t)c, 0);
    pbank.clear();
    {
      std::vector<std::pair<int32_t, int>> pv;
      for (int e = 0; e < c; ++e) pv.emplace_back(sl[e].piv, e);
      std::sort(pv.begin(), pv.end());
      np = 0;
      for (int i = 0; i < c; ++i) {
        if (i == 0 || pv[i].first != pv[i - 1].first) { pbank.push_back(pv[i].first & 31); ++np; }
        pid[pv[i].second] = np - 1;
      }
    }
    std::vector<int> grp((size_t)c);
    // natural placement: slot e at lane e % lanes, half e / lanes
    reset();
    for (int e = 0; e < c; ++e) {
      const int lane = e % lanes, half = e / lanes;
      grp[e] = half * 4 + (lane >> 4);
      add(e, grp[e], +1);
    }
    const int before = total();
    std::vector<int> natural(grp);
    // greedy: the slots of a pivot together (largest columns first), each into the write group where it adds the fewest
    // collisions; ties go to a group that already reads the pivot, then to the emptier one
    std::vector<int> cnt((size_t)np, 0), order((size_t)c), fill(8, 0);
    for (int e = 0; e < c; ++e) cnt[pid[e]]++;
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
      if (cnt[pid[x]] != cnt[pid[y]]) return cnt[pid[x]] > cnt[pid[y]];
      if (pid[x] != pid[y]) return pid[x] < pid[y];
      return sl[x].tgt < sl[y].tgt;
    });
    reset();
    for (int e : order) {
      int bg = -1;
      long bs = 0;
      for (int g = 0; g < 8; ++g) {
        if (fill[g] >= capacity(g)) continue;
        const int r = g >> 1;
        const bool has = pc[r][pid[e]] > 0;
        const int col = tc[r].n[sl[e].tgt & 31] + wc[g].n[sl[e].tgt & 15] + (has ? 0 : pb[r].n[pbank[pid[e]]]);
        const long score = 1000L * col + (has ? 0 : 100) + fill[g];
        if (bg < 0 || score < bs) { bg = g; bs = score; }
      }
      grp[e] = bg;
      fill[bg]++;
      add(e, bg, +1);
    }
    // local search: a slot on an over-subscribed bank swaps with a slot of another write group (or moves into free
    // capacity there) whenever that lowers the modelled cycles of the groups involved
    int nrg = 0, nwg = 0;  // groups in use: their count is the floor of the model
    for (int g = 0; g < 8; ++g) nwg += capacity(g) > 0;
    for (int r = 0; r < 4; ++r) nrg += capacity(2 * r) > 0;
    for (int pass = 0; pass < search && total() > 2 * nrg + nwg; ++pass) {
      bool improved = false;
      for (int a = 0; a < c; ++a) {
        if (!bad(a, grp[a])) continue;
        const int g = grp[a];
        bool moved = false;
        for (int h = 0; h < 8 && !moved; ++h) {  // free capacity first
          if (h == g || fill[h] >= capacity(h)) continue;
          const int c0 = cost_of(g, h);
          add(a, g, -1); add(a, h, +1);
          if (cost_of(g, h) < c0) { grp[a] = h; fill[g]--; fill[h]++; moved = true; }
          else { add(a, h, -1); add(a, g, +1); }
        }
        for (int b = 0; b < c && !moved; ++b) {
          const int h = grp[b];
          if (h == g) continue;
          const int c0 = cost_of(g, h);
          add(a, g, -1); add(b, h, -1); add(a, h, +1); add(b, g, +1);
          if (cost_of(g, h) < c0) { grp[a] = h; grp[b] = g; moved = true; }
          else { add(a, h, -1); add(b, g, -1); add(a, g, +1); add(b, h, +1); }
        }
        improved = improved || moved;
      }
      if (!improved) break;
    }
    int after = total();
    if (after >= before) { grp = natural; after = before; }
    where.assign((size_t)c, -1);
    if (grp == natural) {
      for (int e = 0; e < c; ++e) where[e] = (e % lanes) * 2 + e / lanes;
    } else {
      int nxt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int e = 0; e < c; ++e) {
        const int g = grp[e], half = g >> 2, l0 = (g & 3) * 16;
        where[e] = (l0 + nxt[g]++) * 2 + half;
      }
    }
    return {before, after};
  }
};

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
                       const char **msg, int lds_hint)
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
  // permuted lower CSC + elimination tree (numbering S: the elimination order `perm`)
  struct PE { int i, j, kind, idx; };
  std::vector<PE> pe;
  std::vector<int32_t> rp, rj, parent, flag(k, -1), lnz;
  {
    o.pinv.resize(k);
    for (int i = 0; i < k; ++i) o.pinv[o.perm[i]] = i;
    pe.reserve(ents.size());
    for (const Ent &e : ents) {
      const int a = o.pinv[e.r], b = o.pinv[e.c];
      pe.push_back({std::max(a, b), std::min(a, b), e.kind, e.idx});
    }
    std::sort(pe.begin(), pe.end(), [](const PE &x, const PE &y) { return std::tie(x.j, x.i) < std::tie(y.j, y.i); });
    // rows of the lower form (strict): row r -> columns j < r
    rp.assign(k + 1, 0);
    for (const PE &e : pe)
      if (e.i != e.j) rp[e.i + 1]++;
    for (int r = 0; r < k; ++r) rp[r + 1] += rp[r];
    rj.resize(rp[k]);
    {
      std::vector<int32_t> fill(k, 0);
      for (const PE &e : pe)
        if (e.i != e.j) rj[rp[e.i] + fill[e.i]++] = e.j;
    }
    // elimination tree + column counts by row reach
    parent.assign(k, -1);
    lnz.assign(k, 0);
    for (int r = 0; r < k; ++r) {
      flag[r] = r;
      for (int p = rp[r]; p < rp[r + 1]; ++p)
        for (int i = rj[p]; flag[i] != r; i = parent[i]) {
          if (parent[i] == -1) parent[i] = r;
          lnz[i]++;
          flag[i] = r;
        }
    }
  }
  o.nnzK = (int)pe.size();
  o.Kp.assign(k + 1, 0); o.Ki.resize(o.nnzK); o.Kkind.resize(o.nnzK); o.Kidx.resize(o.nnzK);
  for (int t = 0; t < o.nnzK; ++t) {
    o.Kp[pe[t].j + 1]++;
    o.Ki[t] = pe[t].i; o.Kkind[t] = pe[t].kind; o.Kidx[t] = pe[t].idx;
  }
  for (int j = 0; j < k; ++j) o.Kp[j + 1] += o.Kp[j];
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

  // ---- factorisation numbering F: a POSTORDER of the elimination tree (children in ascending order) ----
  // Same tree, same pattern of L; every subtree becomes a contiguous range of columns.  The numeric factorisation
  // (accumulators, supernodes, the order in which an accumulator receives its updates) lives in F; the triangular
  // sweeps keep the numbering S of `perm`.  rankF[S column] = F column, f2s = inverse.
  std::vector<int32_t> rankF(k), parF(k, -1);
  {
    std::vector<int32_t> head(k, -1), next(k, -1), stack;
    for (int j = k - 1; j >= 0; --j)
      if (parent[j] >= 0) { next[j] = head[parent[j]]; head[parent[j]] = j; }
    o.f2s.clear();
    o.f2s.reserve(k);
    for (int r = 0; r < k; ++r) {
      if (parent[r] >= 0) continue;
      stack.push_back(r);
      while (!stack.empty()) {
        const int v = stack.back(), c = head[v];
        if (c >= 0) { head[v] = next[c]; stack.push_back(c); }
        else { rankF[v] = (int32_t)o.f2s.size(); o.f2s.push_back(v); stack.pop_back(); }
      }
    }
    for (int j = 0; j < k; ++j)
      if (parent[j] >= 0) parF[rankF[j]] = rankF[parent[j]];
  }
  // pattern of L in F (column-major, rows ascending) and the position of every S entry in it
  std::vector<int32_t> FLp(k + 1, 0), FLi(o.nnzL), posF(o.nnzL);
  {
    std::vector<std::pair<int32_t, int32_t>> tmp;
    for (int jF = 0; jF < k; ++jF) {
      const int jS = o.f2s[jF];
      tmp.clear();
      for (int p = o.Lp[jS]; p < o.Lp[jS + 1]; ++p) tmp.emplace_back(rankF[o.Li[p]], p);
      std::sort(tmp.begin(), tmp.end());
      FLp[jF + 1] = FLp[jF] + (int)tmp.size();
      for (size_t e = 0; e < tmp.size(); ++e) {
        if (tmp[e].first <= jF) { *msg = "internal: postorder does not keep ancestors behind"; return false; }
        FLi[FLp[jF] + e]   = tmp[e].first;
        posF[tmp[e].second] = FLp[jF] + (int)e;
      }
    }
  }
  // KKT entries in F (lower CSC): {row, kind, idx}
  std::vector<int32_t> FKp(k + 1, 0), FKi(o.nnzK), FKkind(o.nnzK), FKidx(o.nnzK);
  {
    std::vector<std::array<int32_t, 4>> fe;
    fe.reserve(o.nnzK);
    for (int jS = 0; jS < k; ++jS)
      for (int p = o.Kp[jS]; p < o.Kp[jS + 1]; ++p) fe.push_back({rankF[jS], rankF[o.Ki[p]], o.Kkind[p], o.Kidx[p]});
    std::sort(fe.begin(), fe.end());
    for (int t = 0; t < o.nnzK; ++t) {
      if (fe[t][1] < fe[t][0]) { *msg = "internal: KKT entry above the diagonal in the factorisation numbering"; return false; }
      FKp[fe[t][0] + 1]++;
      FKi[t] = fe[t][1]; FKkind[t] = fe[t][2]; FKidx[t] = fe[t][3];
    }
    for (int j = 0; j < k; ++j) FKp[j + 1] += FKp[j];
  }

  // right-looking factorisation schedule (see sparse_plan.h), everything in F
  {
    auto pos_in_col = [&](int col, int row) {  // binary search: rows ascending
      const int32_t *b = FLi.data() + FLp[col], *e = FLi.data() + FLp[col + 1];
      const int32_t *it = std::lower_bound(b, e, row);
      return (it != e && *it == row) ? (int)(it - FLi.data()) : -1;
    };
    o.Kmap.resize(o.nnzK);
    for (int j = 0; j < k; ++j)
      for (int p = FKp[j]; p < FKp[j + 1]; ++p) {
        const int i = FKi[p];
        o.Kmap[p]   = (i == j) ? o.nnzL + j : pos_in_col(j, i);
        if (o.Kmap[p] < 0) { *msg = "internal: KKT entry outside the pattern of L"; return false; }
      }
    o.maxcol = 0;
    for (int kk = 0; kk < k; ++kk) o.maxcol = std::max(o.maxcol, FLp[kk + 1] - FLp[kk]);
    if (o.maxcol >= (1 << 16)) { *msg = "column of L too long for the update encoding"; return false; }
    // LDS per item: the work vector of the sweeps (k + 1), the panel scratch of the widest column, and room for the
    // accumulators of a subtree (below)
    // Unit engine (below): as little as holds the whole factor -- or, when that is more than kLdsSmall doubles, the work
    // vector and kLdsSmall (twelve waves per CU instead of ten for the MPC-sized plans; the segments shrink with it).
    // When some column does not fit that on its own, kLdsTarget is tried, then the supernodal engine with kLdsTarget.
    // lds_hint > 0 (the whole-pattern fallback of a pruned plan): the LDS of the plan it accompanies -- the kernel is
    // launched with the larger of the two, and the fallback solves a handful of items.
    constexpr int kLdsTarget = 2048, kLdsSmall = 1536;
    const int lds_floor = std::max(k + 2, 2 * o.maxcol + 4);
    const int lds_small = lds_hint > 0 ? std::max(lds_floor, lds_hint)
                                       : (std::max(lds_floor, std::min(o.nnzL + k + 3, kLdsSmall)) + 63) / 64 * 64;
    const int lds_large = std::max(lds_small, lds_hint > 0 ? lds_hint : kLdsTarget);
    o.lds_doubles = lds_small;

    // ---- LDS-resident subtrees ----
    // A subtree of the elimination tree (a contiguous column range [c0, c1) in F) whose accumulators -- its
    // nL entries of L and its c1 - c0 diagonal entries -- fit the item's LDS is factorised ON CHIP: KKT fill,
    // panel gathers and the updates between its columns never touch HBM (they are the dependent memory round trips
    // that bound the factorisation of a lone wave); only the final values and the updates to ancestors outside
    // the subtree go to the workspace.  Updates reach an accumulator only from descendants of its column, so a
    // subtree's accumulators are touched by its own columns alone, and processing the segments in ascending column
    // order keeps the ascending-source order of every update sequence.  Maximal fitting subtrees are chosen
    // top-down; columns outside them ("top") keep their accumulators in HBM.
    std::vector<int32_t> fdesc(k);  // first descendant: subtree(j) = [fdesc[j], j]
    for (int j = 0; j < k; ++j) fdesc[j] = j;
    for (int j = 0; j < k; ++j)
      if (parF[j] >= 0) fdesc[parF[j]] = std::min(fdesc[parF[j]], fdesc[j]);
    std::vector<int32_t> maxR1(k);  // largest single-column panel (1 + column count) inside subtree(j)
    for (int j = 0; j < k; ++j) maxR1[j] = 1 + FLp[j + 1] - FLp[j];
    for (int j = 0; j < k; ++j)
      if (parF[j] >= 0) maxR1[parF[j]] = std::max(maxR1[parF[j]], maxR1[j]);
    struct Seg { int c0, c1, lds, nL, accN, sn0, sn1, nout, omap0; };
    std::vector<Seg> segs;
    // `units`: segments for the unit engine (below) -- a subtree fits when its accumulators and the accumulators of
    // its ancestors that it updates (at most all pairs of the root column's structure) do
    auto choose_segments = [&](const bool units) {
      segs.clear();
      constexpr int kMinCols = 8;
      std::vector<char> in_lds(k, 0);
      for (int j = k - 1; j >= 0;) {
        const int c0 = fdesc[j], nc = j - c0 + 1, nL = FLp[j + 1] - FLp[c0];
        const int reserve = std::max(2 * maxR1[j], 384);  // panel + multipliers of the supernodes formed below
        const int ccj = FLp[j + 1] - FLp[j];
        const bool fits = units ? nL + nc + ccj * (ccj + 1) / 2 + 3 <= o.lds_doubles
                                : (maxR1[j] <= 64 && nL + nc + 2 + reserve <= o.lds_doubles);
        if (nc >= kMinCols && fits) {
          for (int c = c0; c <= j; ++c) in_lds[c] = 1;
          j = c0 - 1;
        } else {
          --j;
        }
      }
      for (int c = 0; c < k;) {  // maximal runs: LDS subtrees (one per root found above) and top columns
        int e = c + 1;
        if (in_lds[c]) {
          // the run [c, e) must be ONE subtree: extend to the root whose first descendant is c
          while (e < k && in_lds[e] && fdesc[e] >= c) ++e;
          // e - 1 is the root only if fdesc[e - 1] == c; a run of several sibling subtrees is split at roots
          int root = e - 1;
          while (fdesc[root] != c) --root;
          e = root + 1;
          segs.push_back({c, e, 1, FLp[e] - FLp[c], FLp[e] - FLp[c] + (e - c), 0, 0, 0, 0});
        } else {
          while (e < k && !in_lds[e]) ++e;
          segs.push_back({c, e, 0, 0, 0, 0, 0, 0, 0});
        }
        c = e;
      }
    };

    // ---- UNIT ENGINE of the numeric factorisation (see sparse_plan.h) ----
    // Every segment's work -- the divisions L(r, j) = acc(r, j) / D(j) ("DM" slots) and the updates
    // acc(a, b) = fma(-L(a, j), L(b, j) D(j), acc(a, b)) ("F" slots) of its columns -- is list-scheduled into units of
    // 128 independent slots of ONE kind, executed on accumulators that all live in LDS: the segment's own ones and the
    // ("outside") accumulators of later columns that it updates, fetched from the workspace when the segment starts
    // and returned when it ends.  A slot waits for the previous writer of its target (the updates of one accumulator
    // keep the ascending order of their source columns: the oracle's order) and for the slots that finalise what it
    // reads.  Applicable when every column fits the LDS on its own; otherwise the supernodal engine below runs.
    o.units = 0;
    o.ustream.clear(); o.utype.clear(); o.uomap.clear();
    {
      const char *uk = sfb::knob("SFB_PLAN_UNITS");  // A/B knob: 0 = the supernodal engine for every plan
      bool ok = false;
      for (int attempt = 0; attempt < 2 && !ok; ++attempt) {
      if (attempt == 1 && lds_large == lds_small) break;
      o.lds_doubles = attempt == 0 ? lds_small : lds_large;
      ok = !(uk && uk[0] == '0') && (size_t)o.lds_doubles * 8 < (1u << 16);
      if (ok) {
        choose_segments(true);
        // top runs are cut greedily into ranges whose own + outside accumulators fit
        std::vector<Seg> cut;
        std::vector<int32_t> seen(o.nnzL + k, -1), percol(k + 1, 0);
        for (const Seg &sg : segs) {
          if (sg.lds) { cut.push_back(sg); continue; }
          int c0 = sg.c0;
          while (c0 < sg.c1 && ok) {
            // grow [c0, c1): own = entries + diagonals of the columns, outside = distinct accumulators (a, b), b >= c1
            int own = 0, nout = 0, best = -1;
            std::vector<int32_t> touched;
            for (int j = c0; j < sg.c1; ++j) {
              own += FLp[j + 1] - FLp[j] + 1;
              nout -= percol[j];  // accumulators of column j itself are own from now on
              for (int bp = FLp[j]; bp < FLp[j + 1]; ++bp)
                for (int ap = bp; ap < FLp[j + 1]; ++ap) {
                  const int b = FLi[bp], a = FLi[ap];
                  const int g = (a == b) ? o.nnzL + b : pos_in_col(b, a);
                  if (g < 0) { *msg = "internal: update outside the pattern of L"; return false; }
                  if (seen[g] != c0) { seen[g] = c0; touched.push_back(g); percol[b]++; ++nout; }
                }
              if (own + nout + 3 <= o.lds_doubles) best = j + 1;
              else if (best >= 0) break;
              else if (j == c0) break;
            }
            for (int g : touched) seen[g] = -1;
            std::fill(percol.begin(), percol.end(), 0);
            if (best < 0) { ok = false; break; }
            cut.push_back({c0, best, 0, 0, 0, 0, 0, 0, 0});
            c0 = best;
          }
        }
        if (ok) segs.swap(cut);
      }
      }
      if (ok) {
        struct Op { int32_t t, a, b, d; };  // LDS offsets (DM: a = b = -1)
        std::vector<int32_t> outidx(o.nnzL + k, -1);
        for (Seg &sg : segs) {
          const int c0 = sg.c0, c1 = sg.c1;
          sg.nL   = FLp[c1] - FLp[c0];
          sg.accN = sg.nL + (c1 - c0);
          sg.omap0 = (int)o.uomap.size();
          sg.sn0   = (int)o.utype.size();
          // outside accumulators, in order of first touch
          int nout = 0;
          auto off_of = [&](int g) {
            if (g < o.nnzL) { if (g >= FLp[c0] && g < FLp[c1]) return g - FLp[c0]; }
            else if (g - o.nnzL >= c0 && g - o.nnzL < c1) return sg.nL + (g - o.nnzL - c0);
            if (outidx[g] < 0) { outidx[g] = nout++; o.uomap.push_back(g); }
            return sg.accN + outidx[g];
          };
          std::vector<Op> ops;
          std::vector<std::array<int32_t, 3>> deps;  // up to three predecessors (-1 = none)
          std::vector<int32_t> dm_of(sg.nL, -1);
          std::vector<int32_t> writer;               // LDS offset -> last op that wrote it
          for (int j = c0; j < c1; ++j) {
            const int dOff = sg.nL + (j - c0);
            if ((int)writer.size() < sg.accN) writer.resize(sg.accN, -1);
            for (int p = FLp[j]; p < FLp[j + 1]; ++p) {
              const int tOff = p - FLp[c0];
              deps.push_back({writer[tOff], writer[dOff], -1});
              dm_of[tOff] = (int)ops.size();
              writer[tOff] = (int)ops.size();
              ops.push_back({tOff, -1, -1, dOff});
            }
            for (int bp = FLp[j]; bp < FLp[j + 1]; ++bp)
              for (int ap = bp; ap < FLp[j + 1]; ++ap) {
                const int b = FLi[bp], a = FLi[ap];
                const int g = (a == b) ? o.nnzL + b : pos_in_col(b, a);
                if (g < 0) { *msg = "internal: update outside the pattern of L"; return false; }
                const int T = off_of(g);
                if ((int)writer.size() <= T) writer.resize(T + 1, -1);
                deps.push_back({dm_of[ap - FLp[c0]], dm_of[bp - FLp[c0]], writer[T]});
                writer[T] = (int)ops.size();
                ops.push_back({T, ap - FLp[c0], bp - FLp[c0], dOff});
              }
          }
          for (int e = sg.omap0; e < (int)o.uomap.size(); ++e) outidx[o.uomap[e]] = -1;
          sg.nout = nout;
          if (sg.accN + nout + 3 > o.lds_doubles) { *msg = "internal: segment does not fit the LDS"; return false; }
          const int SINK = sg.accN + nout, ZEROL = SINK + 1, ONE = SINK + 2;
          // list scheduling on the critical path, units of one kind
          const int nops = (int)ops.size();
          std::vector<int32_t> nsucc(nops + 1, 0), succ, height(nops, 1), npred(nops, 0);
          for (int i = 0; i < nops; ++i)
            for (int d : deps[i])
              if (d >= 0) { nsucc[d + 1]++; npred[i]++; }
          for (int i = 0; i < nops; ++i) nsucc[i + 1] += nsucc[i];
          succ.resize(nsucc[nops]);
          {
            std::vector<int32_t> fill(nops, 0);
            for (int i = 0; i < nops; ++i)
              for (int d : deps[i])
                if (d >= 0) succ[nsucc[d] + fill[d]++] = i;
          }
          for (int i = nops - 1; i >= 0; --i)
            for (int d : deps[i])
              if (d >= 0) height[d] = std::max(height[d], height[i] + 1);
          std::vector<std::pair<int32_t, int32_t>> ready[2];  // heaps of (height, -op): [0] F, [1] DM
          for (int i = 0; i < nops; ++i)
            if (npred[i] == 0) ready[ops[i].a < 0].emplace_back(height[i], -i);
          for (auto &h : ready) std::make_heap(h.begin(), h.end());
          int done = 0;
          std::vector<int32_t> cur, arriving;
          while (done < nops) {
            // the kind whose best ready slot has the longer chain of dependants; ties: divisions (they release updates)
            int kind;
            if (ready[0].empty()) kind = 1;
            else if (ready[1].empty()) kind = 0;
            else kind = ready[1].front().first >= ready[0].front().first;
            cur.clear();
            arriving.clear();
            while ((int)cur.size() < 128 && !ready[kind].empty()) {
              std::pop_heap(ready[kind].begin(), ready[kind].end());
              cur.push_back(-ready[kind].back().second);
              ready[kind].pop_back();
            }
            done += (int)cur.size();
            const size_t base = o.ustream.size();
            o.ustream.resize(base + 256);
            o.utype.push_back(kind);
            for (int e = 0; e < 128; ++e) {
              const int lane = e % 64, half = e / 64;
              int32_t w0, w1;
              if (e < (int)cur.size()) {
                const Op &op = ops[cur[e]];
                if (kind) { w0 = (op.t * 8) | ((op.d * 8) << 16); w1 = -1; }
                else { w0 = (op.t * 8) | ((op.a * 8) << 16); w1 = (op.b * 8) | ((op.d * 8) << 16); }
              } else if (kind) { w0 = (SINK * 8) | ((ONE * 8) << 16); w1 = -1; }
              else { w0 = (SINK * 8) | ((ZEROL * 8) << 16); w1 = (ZEROL * 8) | ((ZEROL * 8) << 16); }
              o.ustream[base + (size_t)lane * 4 + 2 * half]     = w0;
              o.ustream[base + (size_t)lane * 4 + 2 * half + 1] = w1;
            }
            for (int i : cur)
              for (int q = nsucc[i]; q < nsucc[i + 1]; ++q)
                if (--npred[succ[q]] == 0) arriving.push_back(succ[q]);
            for (int i : arriving) {
              auto &h = ready[ops[i].a < 0];
              h.emplace_back(height[i], -i);
              std::push_heap(h.begin(), h.end());
            }
          }
          sg.sn1 = (int)o.utype.size();
        }
        o.units  = 1;
        o.nunits = (int)o.utype.size();
        o.ustream.resize(o.ustream.size() + (size_t)256 * 2 * SparsePlanHost::kSweepPad, 0);  // prefetched past the end (two blocks), never executed
        o.utype.resize(o.utype.size() + SparsePlanHost::kSweepPad + 32, 0);
        o.uomap.resize(o.uomap.size() + 64 * 8, o.nnzL + k);  // padding: the scratch accumulator
      } else {
        o.ustream.clear(); o.utype.clear(); o.uomap.clear();
        o.lds_doubles = lds_large;
        choose_segments(false);
      }
    }

    // ---- relaxed supernodes (see sparse_plan.h) ----
    // Greedy grouping of consecutive columns of one segment: the panel rows are the columns themselves plus the
    // union U of their remaining row structures.  A column joins while the panel and its multipliers (2 w R
    // doubles) fit the LDS scratch (what the segment's accumulators leave of it), w <= 16, and the union grows by
    // at most kRelax rows over the larger of the two structures (explicit zeros cost LDS work, not HBM traffic).
    constexpr int kRelax = 4, kMaxWidth = 16;
    const int ZERO = o.nnzL + k + 1, SCRATCH = o.nnzL + k;  // accumulator indices: always-zero entry, padding sink
    o.snptr.clear(); o.snR.clear(); o.poff.clear(); o.pmap.clear(); o.pmapL.clear(); o.rptr.clear(); o.rsplit.clear();
    o.rtgt.clear(); o.rab.clear();
    o.rptr.push_back(0);
    for (Seg &sg : segs) {
      if (o.units) break;
      sg.sn0 = (int)o.snptr.size();
      const int scratch = sg.lds ? o.lds_doubles - ((sg.accN + 2 + 1) & ~1) : o.lds_doubles;
      // accumulator -> LDS offset inside an LDS segment: [L entries | D | sink | zero]
      auto lds_off = [&](int g) {
        if (g == SCRATCH) return sg.accN;
        if (g == ZERO) return sg.accN + 1;
        if (g < o.nnzL) return (g >= FLp[sg.c0] && g < FLp[sg.c1]) ? g - FLp[sg.c0] : -1;
        const int j = g - o.nnzL;
        return (j >= sg.c0 && j < sg.c1) ? sg.nL + (j - sg.c0) : -1;
      };
      int j0 = sg.c0;
      while (j0 < sg.c1) {
        std::vector<int32_t> U(FLi.begin() + FLp[j0], FLi.begin() + FLp[j0 + 1]);  // sorted rows outside the panel
        int w = 1;
        while (j0 + w < sg.c1 && w < kMaxWidth) {
          const int jn = j0 + w;
          std::vector<int32_t> Un;
          std::set_union(U.begin(), U.end(), FLi.begin() + FLp[jn], FLi.begin() + FLp[jn + 1], std::back_inserter(Un));
          Un.erase(std::remove_if(Un.begin(), Un.end(), [&](int32_t r) { return r <= jn; }), Un.end());
          const int Rn = (w + 1) + (int)Un.size();
          if (2 * (w + 1) * Rn > scratch) break;
          if (sg.lds && Rn > 64) break;  // on-chip segments eliminate their panels in registers (lane = row)
          int ubase = 0;
          for (int32_t r : U) ubase += (r != jn);
          const int base = std::max(ubase, FLp[jn + 1] - FLp[jn]);
          if ((int)Un.size() - base > kRelax) break;
          U.swap(Un);
          ++w;
        }
        const int nu = (int)U.size(), R = w + nu;
        if (2 * w * R > scratch) { *msg = "internal: panel does not fit the LDS scratch"; return false; }
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
            o.pmapL.push_back(sg.lds ? lds_off(src) : 0);
            if (sg.lds && o.pmapL.back() < 0) { *msg = "internal: panel entry outside its LDS segment"; return false; }
          }
        // trailing schedule: pairs (a >= b) of U whose accumulator exists and is touched by some member; inside an
        // LDS segment first the slots whose accumulator is on chip (target = LDS offset), then the others
        std::vector<std::array<int32_t, 2>> slots[2];
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
            const int off = sg.lds ? lds_off(tgt) : -1;
            if (off >= 0) slots[0].push_back({off, aq | (bq << 16)});
            else slots[1].push_back({tgt, aq | (bq << 16)});
          }
        int total = 0;
        for (int part = 0; part < 2; ++part) {
          const int steps = (int)((slots[part].size() + 63) / 64);
          const size_t q0 = o.rtgt.size();
          o.rtgt.resize(q0 + (size_t)steps * 64, part == 0 ? sg.accN : SCRATCH);  // padding: the sink of that space
          o.rab.resize(q0 + (size_t)steps * 64, 0);
          for (size_t e = 0; e < slots[part].size(); ++e) { o.rtgt[q0 + e] = slots[part][e][0]; o.rab[q0 + e] = slots[part][e][1]; }
          total += steps;
          if (part == 0) o.rsplit.push_back(o.rptr.back() + steps);
        }
        o.rptr.push_back(o.rptr.back() + total);
        j0 += w;
      }
      sg.sn1 = (int)o.snptr.size();
    }
    o.nsn = (int)o.snptr.size();
    // segment table {sn0, sn1, c0, c1, lds, nL, accN, first K entry, last K entry + 1, first L entry, outside accumulators,
    // start of their map} (unit engine: sn0, sn1 = the segment's units)
    o.nseg = (int)segs.size();
    o.seg.clear();
    for (const Seg &sg : segs)
      for (int v : {sg.sn0, sg.sn1, sg.c0, sg.c1, sg.lds, sg.nL, sg.accN, (int)FKp[sg.c0], (int)FKp[sg.c1], (int)FLp[sg.c0], sg.nout, sg.omap0})
        o.seg.push_back(v);
    static_assert(SparsePlanHost::kSegStride == 12, "segment table layout");
    // KKT fill.  Kdesc: {kind, idx, r, c} of every entry in F order (row / column of the source entry resolved
    // here), Kmap its accumulator, KmapL its LDS offset inside an LDS segment; KdescT / KmapT: the entries of the
    // TOP columns only (their accumulators live in HBM), padded for branch-free batches.
    std::vector<char> col_lds(k, 0);
    for (const Seg &sg : segs)
      if (sg.lds) std::fill(col_lds.begin() + sg.c0, col_lds.begin() + sg.c1, 1);
    o.Kdesc.assign((size_t)(o.nnzK + 64 * 8) * 4, 0);
    o.KmapL.assign((size_t)o.nnzK + 64 * 8, 0);
    o.KdescT.clear(); o.KmapT.clear();
    {
      size_t si = 0;
      for (int j = 0; j < k; ++j) {
        while (segs[si].c1 <= j) ++si;
        const Seg &sg = segs[si];
        for (int p = FKp[j]; p < FKp[j + 1]; ++p) {
          const int kind = FKkind[p], idx = FKidx[p];
          int r = 0, c = 0;
          if (kind == K_P) { r = o.Pi[idx]; c = o.Pcol[idx]; }
          else if (kind == K_A) { r = o.Arow[idx]; c = o.Aj[idx]; }
          const int32_t d4[4] = {kind, idx, r, c};
          std::copy(d4, d4 + 4, o.Kdesc.begin() + 4 * (size_t)p);
          if (sg.lds) {
            const int g = o.Kmap[p];
            o.KmapL[p]  = (g < o.nnzL) ? g - FLp[sg.c0] : sg.nL + (g - o.nnzL - sg.c0);
          } else {
            o.KdescT.insert(o.KdescT.end(), d4, d4 + 4);
            o.KmapT.push_back(o.Kmap[p]);
          }
        }
      }
    }
    o.nnzKT = (int)o.KmapT.size();
    for (int pad = 0; pad < 64 * 8; ++pad) {
      const int32_t d4[4] = {K_SIGMA, 0, 0, 0};  // padding: a constant, to the scratch accumulator
      o.KdescT.insert(o.KdescT.end(), d4, d4 + 4);
      o.KmapT.push_back(SCRATCH);
    }
    for (size_t p = o.nnzK; p < (size_t)o.nnzK + 64 * 8; ++p) o.Kdesc[4 * p] = K_SIGMA;
    o.Kmap.resize((size_t)o.nnzK + 64 * 8, SCRATCH);
    // accumulator ranges of the top columns (what the kernel zeroes in HBM before the fill): {start, length} pairs
    o.ztop.clear();
    for (const Seg &sg : segs)
      if (!sg.lds) {
        o.ztop.push_back(FLp[sg.c0]); o.ztop.push_back(FLp[sg.c1] - FLp[sg.c0]);
        o.ztop.push_back(o.nnzL + sg.c0); o.ztop.push_back(sg.c1 - sg.c0);
      }
    o.ztop.push_back(SCRATCH); o.ztop.push_back(2);
    o.nztop = (int)o.ztop.size() / 2;
    if (const char *dbg = sfb::knob("SFB_PLAN_DEBUG"); dbg && dbg[0] == '1') {  // diagnostics: segments and supernodes
      int maxR = 0, over64 = 0, nlds = 0, cols_lds = 0;
      long long panel = 0, acc_lds = 0, ext = 0, internal = 0;
      for (const Seg &sg : segs) {
        nlds += sg.lds; cols_lds += sg.lds ? sg.c1 - sg.c0 : 0; acc_lds += sg.lds ? sg.accN : 0;
        fprintf(stderr, "[sfb plan] segment columns %d..%d %s accumulators %d supernodes %d\n", sg.c0, sg.c1 - 1, sg.lds ? "LDS" : "top",
                sg.lds ? sg.accN : FLp[sg.c1] - FLp[sg.c0] + sg.c1 - sg.c0, sg.sn1 - sg.sn0);
      }
      for (int sn = 0; sn < o.nsn; ++sn) {
        const int w = (sn + 1 < o.nsn ? o.snptr[sn + 1] : k) - o.snptr[sn], R = o.snR[sn];
        maxR = std::max(maxR, R);
        over64 += R > 64;
        panel += (long long)w * R;
        internal += o.rsplit[sn] - o.rptr[sn];
        ext += o.rptr[sn + 1] - o.rsplit[sn];
      }
      if (o.units) {
        int nf = 0, nd = 0, maxw = 0;
        for (int u = 0; u < o.nunits; ++u) (o.utype[u] ? nd : nf)++;
        for (const Seg &sg : segs) maxw = std::max(maxw, sg.accN + sg.nout + 3);
        fprintf(stderr, "[sfb plan] unit engine: %d units (%d update + %d division), %d segments, largest working set %d of %d doubles\n",
                o.nunits, nf, nd, (int)segs.size(), maxw, o.lds_doubles);
      }
      fprintf(stderr, "[sfb plan] %d supernodes, max R %d, %d with R > 64, panel entries %lld, trailing steps %d on chip + %d in HBM; "
              "%d segments, %d on chip with %d columns and %lld accumulators of %d; lds_doubles %d\n", o.nsn, maxR, over64, panel,
              (int)internal, (int)ext, (int)segs.size(), nlds, cols_lds, acc_lds, o.nnzL + k, o.lds_doubles);
    }
    o.snptr.push_back(k);
    o.poff.push_back((int)o.pmap.size());
    o.rsplit.push_back(o.rptr.back());
    o.rsteps = o.rptr.back();
    for (int pad = 0; pad < 64 * SparsePlanHost::kSweepPad; ++pad) {  // branch-free batched reads past the end
      o.pmap.push_back(SCRATCH);
      o.pmapL.push_back(0);
      o.rtgt.push_back(SCRATCH);
      o.rab.push_back(0);
    }
  }
  // the sweeps enumerate the entries of L in S; their values are gathered from the accumulators (F positions)
  auto acc_of = [&](int posS) { return posF[posS]; };

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
    if (const char *dbg = sfb::knob("SFB_PLAN_DEBUG"); dbg && dbg[0] == '1') {  // diagnostics: fill of the sweep units
      fprintf(stderr, "[sfb plan] %s sweep: %d units, slots per unit:", forward ? "forward" : "backward", steps);
      for (int s = 0; s < steps; ++s) fprintf(stderr, " %d", (int)slots[s].size());
      fprintf(stderr, "\n");
    }
    units           = steps;
    units = ((units + 7) / 8) * 8;  // whole prefetch blocks of the kernel (8 units); an empty unit still costs a lone wave 80 ns
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
    constexpr bool bank_aware = true;  // (the natural placement -- slot e at lane e % lanes, half e / lanes -- is what the model is compared with)
    long cyc_before = 0, cyc_after = 0;
    UnitPlacer up;
    const int search_passes = ns <= 20000 ? 1 : 0;  // (one pass does nearly everything; the large whole-pattern fallback plans get the greedy placement only)
    for (int s = 0; s < steps; ++s) {
      const int lanes = std::max(1, ((int)slots[s].size() + 1) / 2);
      xmask[s]        = 64 - lanes;  // exec = all ones >> (64 - lanes)
      // step s -> unit s; the leading `lanes` lanes carry everything, two slots per lane (UnitPlacer: which slot where)
      up.lanes = lanes;
      up.sl.clear();
      for (const auto &e : slots[s]) up.sl.push_back({e[1], e[2]});
      const auto cyc = up.place(search_passes);
      cyc_before += cyc.first;
      cyc_after += bank_aware ? cyc.second : cyc.first;
      for (size_t e = 0; e < slots[s].size(); ++e) {
        const int pos  = bank_aware ? up.where[e] : (int)((e % (size_t)lanes) * 2 + e / (size_t)lanes);
        const size_t q = (size_t)s * 128 + (size_t)pos;
        xmap[q] = acc_of(slots[s][e][0]);
        xidx[q] = (slots[s][e][1] * sc) | ((slots[s][e][2] * sc) << 16);
      }
    }
    if (const char *dbg = sfb::knob("SFB_PLAN_DEBUG"); dbg && dbg[0] == '1')
      fprintf(stderr, "[sfb plan] %s sweep: modelled LDS cycles of the slot accesses per sweep: natural placement %ld, bank-aware %ld (floor %d)\n",
              forward ? "forward" : "backward", cyc_before, cyc_after, 16 * steps);
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
