// C-ABI of the device-side MPC assembly and of the device-resident swarm (include/sfb.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "knobs.h"
#include "../../include/sfb.h"
#include "capi_common.h"
#include "mpc_kernel.h"
#include "qp_sparse_kernel.h"
#include "sparse_plan.h"

namespace sfb {
// order[] = the indices 0 .. B-1 sorted by count DESCENDING, equal counts in index order (what std::stable_sort gives): a counting
// sort when the counts are small numbers (iteration counts: they are), two passes over B entries instead of B log B comparisons
static void order_by_count_descending(const std::vector<uint32_t> &count, std::vector<int32_t> &order)
{
  const size_t B = count.size();
  order.resize(B);
  uint32_t top = 0;
  for (size_t b = 0; b < B; ++b) top = std::max(top, count[b]);
  if ((size_t)top > 4 * B + 65536) {  // (sparse keys: the general sort)
    for (size_t b = 0; b < B; ++b) order[b] = (int32_t)b;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t c) { return count[a] > count[c]; });
    return;
  }
  std::vector<uint32_t> start((size_t)top + 2, 0);  // start[v] = number of entries with a count > v, after the prefix pass
  for (size_t b = 0; b < B; ++b) ++start[count[b]];
  uint32_t run = 0;
  for (size_t v = (size_t)top + 1; v-- > 0;) {
    const uint32_t c = start[v];
    start[v]         = run;
    run += c;
  }
  for (size_t b = 0; b < B; ++b) order[start[count[b]]++] = (int32_t)b;
}
}  // namespace sfb

namespace {

struct Sizes {
  int64_t N, dyn_f, dyn_jx, dyn_ju, cr_c, cr_jx, cr_ju, ce_e, ce_J;
};
Sizes sizes_of(const sfb_mpc_layout *L)
{
  Sizes s{};
  s.N      = (int64_t)L->kmesh * L->nivals;
  s.dyn_f  = s.N * L->nx;
  s.dyn_jx = s.N * L->nx * L->nx;
  s.dyn_ju = s.N * L->nx * L->nu;
  s.cr_c   = s.N * L->ncr;
  s.cr_jx  = s.N * L->ncr * L->nx;
  s.cr_ju  = s.N * L->ncr * L->nu;
  s.ce_e   = L->nx;
  s.ce_J   = (int64_t)L->nx * L->nx;
  return s;
}

sfb_status check_layout(const sfb_mpc_layout *L)
{
  if (!L) return sfb::fail(SFB_ERR_INVALID_ARG, "layout is NULL");
  if (L->nx < 1 || L->nu < 1 || L->ncr < 0 || L->kmesh < 1 || L->nivals < 1)
    return sfb::fail(SFB_ERR_INVALID_ARG, "layout: nx, nu, kmesh, nivals must be >= 1 and ncr >= 0");
  if (!L->alpha || !L->D || (L->ncr > 0 && (!L->crl || !L->cru)))
    return sfb::fail(SFB_ERR_INVALID_ARG, "layout: alpha, D (and crl, cru when ncr > 0) must be given");
  if (L->nparts < 0 || (L->nparts > 0 && (!L->part_kind || !L->part_dof)))
    return sfb::fail(SFB_ERR_INVALID_ARG, "layout: part_kind / part_dof missing");
  if (L->nivals > sfb::kMpcMaxIvals || L->kmesh > sfb::kMpcMaxKmesh || L->nx > sfb::kMpcMaxNx || L->ncr > sfb::kMpcMaxNcr)
    return sfb::fail(SFB_ERR_UNSUPPORTED, "layout: supported up to 128 intervals, 8 nodes per interval, nx 24, ncr 16");
  if (L->jac_keep && L->nu > 32) return sfb::fail(SFB_ERR_UNSUPPORTED, "layout: jac_keep needs nu <= 32");
  int dof = 0;
  for (int g = 0; g < L->nparts; ++g) {
    const int k = L->part_kind[g], d = L->part_dof[g];
    if (k != SFB_LIE_RN && k != SFB_LIE_SE2 && k != SFB_LIE_SO3) return sfb::fail(SFB_ERR_INVALID_ARG, "layout: unknown part kind");
    if (d < 1 || ((k == SFB_LIE_SE2 || k == SFB_LIE_SO3) && d != 3)) return sfb::fail(SFB_ERR_INVALID_ARG, "layout: bad part dof");
    dof += d;
  }
  if (L->nparts > 0 && dof != L->nx) return sfb::fail(SFB_ERR_INVALID_ARG, "layout: part dofs do not sum to nx");
  return SFB_OK;
}

// offsets of the record's fields; keep != nullptr: per-agent Jacobians packed by the flags (sfb.h, jac_keep)
void set_packing(sfb::MpcAsmParams &p, bool shared, const uint8_t *keep)
{
  const int64_t N = p.N, nx = p.nx, nu = p.nu, ncr = p.ncr;
  int64_t njx = N * nx * nx, nju = N * nx * nu, ncx = N * ncr * nx, ncu = N * ncr * nu, nJ = nx * nx;
  p.packed = 0;
  if (keep && !shared) {
    p.packed = 1;
    auto masks = [&](const uint8_t *f, int rows, int cols, uint32_t *km, uint16_t *kp) {
      int cnt = 0;
      for (int d = 0; d < rows; ++d) {
        kp[d] = (uint16_t)cnt;
        uint32_t mk = 0;
        for (int c = 0; c < cols; ++c)
          if (f[d * cols + c]) { mk |= 1u << c; ++cnt; }
        km[d] = mk;
      }
      return cnt;
    };
    const uint8_t *f = keep;
    p.n_fx = masks(f, (int)nx, (int)nx, p.km_fx, p.kp_fx); f += nx * nx;
    p.n_fu = masks(f, (int)nx, (int)nu, p.km_fu, p.kp_fu); f += nx * nu;
    p.n_cx = masks(f, (int)ncr, (int)nx, p.km_cx, p.kp_cx); f += ncr * nx;
    p.n_cu = masks(f, (int)ncr, (int)nu, p.km_cu, p.kp_cu); f += ncr * nu;
    p.n_J  = masks(f, (int)nx, (int)nx, p.km_J, p.kp_J);
    njx = N * p.n_fx; nju = N * p.n_fu; ncx = N * p.n_cx; ncu = N * p.n_cu; nJ = p.n_J;
  }
  int64_t o = 0;
  p.o_f = (int)o; o += N * nx;
  p.o_dx = (int)o; o += N * nx;
  if (!shared) {
    p.o_dfdx = (int)o; o += njx;
    p.o_dfdu = (int)o; o += nju;
  }
  p.o_c = (int)o; o += N * ncr;
  if (!shared) {
    p.o_dcdx = (int)o; o += ncx;
    p.o_dcdu = (int)o; o += ncu;
  }
  p.o_e = (int)o; o += nx;
  p.o_J = (int)o; o += nJ;
  p.rec_doubles = o;
}

// ad(a) of the bundle as a sign/index table (lie.hpp: SE2::ad, SO3::ad = hat)
void fill_params(const sfb_mpc_layout *L, bool shared, sfb::MpcAsmParams &p)
{
  std::memset(&p, 0, sizeof(p));
  const Sizes s = sizes_of(L);
  p.nx = L->nx; p.nu = L->nu; p.ncr = L->ncr; p.kmesh = L->kmesh; p.nivals = L->nivals; p.N = (int)s.N;
  p.rowlen_dyn = L->kmesh + L->nx + L->nu;
  p.nnz_dyn    = (int)(s.dyn_f * p.rowlen_dyn);
  p.nnz_cr     = (int)(s.cr_c * (L->nx + L->nu));
  p.nnzA       = p.nnz_dyn + p.nnz_cr + L->nx * L->nx;
  p.m          = (int)(s.dyn_f + s.cr_c + L->nx);
  p.tf         = L->tf;
  set_packing(p, shared, shared ? nullptr : L->jac_keep);
  if (shared) {
    p.o_dfdx = 0;
    p.o_dfdu = (int)s.dyn_jx;
    p.o_dcdx = (int)(s.dyn_jx + s.dyn_ju);
    p.o_dcdu = (int)(s.dyn_jx + s.dyn_ju + s.cr_jx);
  }
  for (int i = 0; i < L->nivals; ++i) p.alpha[i] = L->alpha[i];
  for (int i = 0; i < (L->kmesh + 1) * L->kmesh; ++i) p.D[i] = L->D[i];
  for (int i = 0; i < L->ncr; ++i) { p.crl[i] = L->crl[i]; p.cru[i] = L->cru[i]; }
  int off = 0;
  for (int g = 0; g < L->nparts; ++g) {
    const int k = L->part_kind[g];
    auto set = [&](int r, int c, int src, int sign) { p.adsrc[(off + r) * L->nx + (off + c)] = (int8_t)(sign * (off + src + 1)); };
    if (k == SFB_LIE_SE2 || k == SFB_LIE_SO3) {
      p.has_ad = 1;
      set(0, 1, 2, -1); set(0, 2, 1, +1);
      set(1, 0, 2, +1); set(1, 2, 0, -1);
      if (k == SFB_LIE_SO3) { set(2, 0, 1, -1); set(2, 1, 0, +1); }
    }
    off += L->part_dof[g];
  }
  // a bundle with a non-commutative part takes the ad branch for all of its entries (the zero ones add -0.0)
}

}  // namespace

struct sfb_mpc_swarm {
  sfb_sparse_qp_plan *plan = nullptr;
  sfb::MpcAsmParams rec_own{}, rec_shared{};
  int64_t rec_full_doubles = 0;  // per-agent record without packing: what the record buffers are sized for
  int64_t agents = 0, shared_doubles = 0;
  int n = 0, m = 0, nnzP = 0, nnzA = 0, nu = 0, uoff = 0, devid = 0;
  size_t wsd = 0;  // solver workspace of the whole swarm, in doubles
  char *mem = nullptr;
  double *Px = nullptr, *q = nullptr, *Ax = nullptr, *l = nullptr, *u = nullptr, *x = nullptr, *y = nullptr, *wx = nullptr,
         *wy = nullptr, *rec = nullptr, *shared = nullptr, *du0 = nullptr, *ws = nullptr;
  uint32_t *iter = nullptr;
  int32_t *code = nullptr, *order = nullptr;
  sfb::MpcAsmDesc *table_own = nullptr, *table_shared = nullptr;  // the assembly kernel's tables (one allocation; mpc_kernel.h)
  char *h_out = nullptr;          // pinned: [du0 | iter | code] of a tick come back in ONE copy (they are adjacent on the device)
  std::vector<uint32_t> h_iter;   // iteration counts of the last tick (host), for the launch order of the next
  std::vector<int32_t> h_order;
  bool have_order = false;
  bool order_sorted = false;      // h_order has been made from h_iter
  // pipelined upload (sfb_mpc_swarm_host_records / _upload): pinned host records, a copy stream, and which agents'
  // records of the CURRENT tick are already on their way
  double *pinned = nullptr;
  hipStream_t up_stream = nullptr;
  std::vector<uint8_t> uploaded;
  std::mutex mu;
};

extern "C" {

int64_t sfb_mpc_record_doubles(const sfb_mpc_layout *layout, int shared_jac)
{
  if (check_layout(layout) != SFB_OK) return -1;
  sfb::MpcAsmParams p;
  fill_params(layout, shared_jac != 0, p);
  return p.rec_doubles;
}

int64_t sfb_mpc_shared_jac_doubles(const sfb_mpc_layout *layout)
{
  if (check_layout(layout) != SFB_OK) return -1;
  const Sizes s = sizes_of(layout);
  return s.dyn_jx + s.dyn_ju + s.cr_jx + s.cr_ju;
}

int64_t sfb_mpc_nnzA(const sfb_mpc_layout *layout)
{
  if (check_layout(layout) != SFB_OK) return -1;
  sfb::MpcAsmParams p;
  fill_params(layout, false, p);
  return p.nnzA;
}

sfb_status sfb_mpc_assemble_batch(const sfb_mpc_layout *layout, int64_t batch, const double *records,
                                  const double *shared_jac, double *Ax, double *l, double *u, void *stream)
{
  sfb_status st = check_layout(layout);
  if (st != SFB_OK) return st;
  if (batch < 0) return sfb::fail(SFB_ERR_INVALID_ARG, "batch < 0");
  if (batch > 0 && (!records || !Ax || !l || !u)) return sfb::fail(SFB_ERR_INVALID_ARG, "NULL record / output pointer");
  st = sfb::require_device();
  if (st != SFB_OK) return st;
  if (batch == 0) return SFB_OK;
  sfb::MpcAsmParams p;
  fill_params(layout, shared_jac != nullptr, p);
  hipError_t e = sfb::mpc_assemble_launch(p, batch, records, shared_jac, Ax, l, u, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return sfb::hip_fail(e, "mpc_assemble_kernel launch");
  return SFB_OK;
}

sfb_status sfb_mpc_swarm_create(sfb_sparse_qp_plan *plan, const sfb_mpc_layout *layout, const double *Px,
                                const double *q, int64_t agents, sfb_mpc_swarm **swarm)
{
  if (!swarm) return sfb::fail(SFB_ERR_INVALID_ARG, "swarm is NULL");
  *swarm = nullptr;
  if (!plan) return sfb::fail(SFB_ERR_INVALID_ARG, "plan is NULL");
  sfb_status st = check_layout(layout);
  if (st != SFB_OK) return st;
  if (agents < 1 || agents > 0x7FFFFFFFll) return sfb::fail(SFB_ERR_INVALID_ARG, "agents must be in [1, 2^31-1]");
  const sfb::SparsePlanHost &h = sfb::plan_io(plan);
  sfb::MpcAsmParams p, ps, pfull;
  fill_params(layout, false, p);
  fill_params(layout, true, ps);
  pfull = p;
  set_packing(pfull, false, nullptr);  // buffers are sized for unpacked records: the packing may be switched later
  const int n = layout->nx * (p.N + 1) + layout->nu * p.N;
  if (h.n != n || h.m != p.m || h.nnzA != p.nnzA) return sfb::fail(SFB_ERR_INVALID_ARG, "plan does not have the sizes of the layout's QP");
  {  // the pattern of A must be the one the kernel writes (ocp_to_qp_allocate :56-69)
    std::vector<int32_t> Aj;
    Aj.reserve(p.nnzA);
    const int nx = layout->nx, nu = layout->nu, km = layout->kmesh, uB = nx * (p.N + 1);
    for (int node = 0; node < p.N; ++node)
      for (int d = 0; d < nx; ++d) {
        const int M = (node / km) * km, i = node - M;
        for (int j = 0; j <= km; ++j) {
          if (j == i)
            for (int c = 0; c < nx; ++c) Aj.push_back((M + j) * nx + c);
          else
            Aj.push_back((M + j) * nx + d);
        }
        for (int c = 0; c < nu; ++c) Aj.push_back(uB + node * nu + c);
        if (h.Ap[node * nx + d + 1] != (int)Aj.size()) return sfb::fail(SFB_ERR_INVALID_ARG, "plan: A row pointers differ from the MPC layout");
      }
    for (int node = 0; node < p.N; ++node)
      for (int d = 0; d < layout->ncr; ++d) {
        for (int c = 0; c < nx; ++c) Aj.push_back(node * nx + c);
        for (int c = 0; c < nu; ++c) Aj.push_back(uB + node * nu + c);
      }
    for (int d = 0; d < nx; ++d)
      for (int c = 0; c < nx; ++c) Aj.push_back(c);
    if (Aj != h.Aj) return sfb::fail(SFB_ERR_INVALID_ARG, "plan: A pattern differs from the MPC layout");
  }
  if ((h.nnzP > 0 && !Px) || !q) return sfb::fail(SFB_ERR_INVALID_ARG, "Px / q is NULL");
  st = sfb::require_device();
  if (st != SFB_OK) return st;
  auto *S = new sfb_mpc_swarm;
  S->plan = plan; S->rec_own = p; S->rec_shared = ps; S->agents = agents; S->rec_full_doubles = pfull.rec_doubles;
  S->n = n; S->m = p.m; S->nnzP = h.nnzP; S->nnzA = p.nnzA; S->nu = layout->nu; S->uoff = layout->nx * (p.N + 1);
  S->shared_doubles = sfb_mpc_shared_jac_doubles(layout);
  {
    int64_t wsb = 0;
    st = sfb_sparse_qp_plan_workspace_bytes(plan, agents, &wsb);
    if (st != SFB_OK) { delete S; return st; }
    S->wsd = ((size_t)wsb + 7) / 8;  // doubles, whole batch
  }
  hipError_t e = hipGetDevice(&S->devid);
  const size_t B = (size_t)agents, N = (size_t)n, M = (size_t)p.m;
  const size_t doubles = B * ((size_t)h.nnzP + N + (size_t)p.nnzA + 2 * M + 2 * (N + M) + (size_t)pfull.rec_doubles + (size_t)layout->nu) + S->wsd +
                         (size_t)S->shared_doubles + (size_t)h.nnzP + N;
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&S->mem), doubles * sizeof(double) + B * 12);
  if (e != hipSuccess) {
    delete S;
    return sfb::hip_fail(e, "hipMalloc(swarm)");
  }
  double *d = reinterpret_cast<double *>(S->mem);
  S->Px = d; d += B * h.nnzP;
  S->q = d; d += B * N;
  S->Ax = d; d += B * p.nnzA;
  S->l = d; d += B * M;
  S->u = d; d += B * M;
  S->x = d; d += B * N;
  S->y = d; d += B * M;
  S->wx = d; d += B * N;
  S->wy = d; d += B * M;
  S->rec = d; d += B * pfull.rec_doubles;
  S->shared = d; d += S->shared_doubles;
  double *stage = d; d += h.nnzP + N;  // one copy of Px and q, replicated below
  S->ws = d; d += S->wsd;
  S->du0 = d; d += B * layout->nu;     // du0 | iter | code one behind the other: a tick brings them back in one copy
  S->iter = reinterpret_cast<uint32_t *>(d);
  S->code  = reinterpret_cast<int32_t *>(S->iter + B);
  S->order = S->code + B;
  S->h_iter.assign(B, 0);
  S->h_order.resize(B);
  do {
    {  // the assembly tables of both record forms
      std::vector<sfb::MpcAsmDesc> tb(2 * (size_t)p.nnzA);
      sfb::mpc_build_table(S->rec_own, false, tb.data());
      sfb::mpc_build_table(S->rec_shared, true, tb.data() + p.nnzA);
      if ((e = hipMalloc(reinterpret_cast<void **>(&S->table_own), tb.size() * sizeof(sfb::MpcAsmDesc))) != hipSuccess) break;
      S->table_shared = S->table_own + p.nnzA;
      if ((e = hipMemcpy(S->table_own, tb.data(), tb.size() * sizeof(sfb::MpcAsmDesc), hipMemcpyHostToDevice)) != hipSuccess) break;
    }
    if (h.nnzP > 0 && (e = hipMemcpy(stage, Px, (size_t)h.nnzP * 8, hipMemcpyHostToDevice)) != hipSuccess) break;
    if ((e = hipMemcpy(stage + h.nnzP, q, N * 8, hipMemcpyHostToDevice)) != hipSuccess) break;
    if ((e = sfb::mpc_replicate_launch(stage, h.nnzP, agents, S->Px, nullptr)) != hipSuccess) break;
    if ((e = sfb::mpc_replicate_launch(stage + h.nnzP, (int64_t)N, agents, S->q, nullptr)) != hipSuccess) break;
    if ((e = hipMemset(S->wx, 0, B * (N + M) * 8)) != hipSuccess) break;  // wx and wy are adjacent
    if ((e = hipHostMalloc(reinterpret_cast<void **>(&S->h_out), B * ((size_t)layout->nu * 8 + 8), hipHostMallocDefault)) != hipSuccess) break;
    e = hipDeviceSynchronize();
  } while (false);
  if (e != hipSuccess) {
    if (S->h_out) (void)hipHostFree(S->h_out);
    (void)hipFree(S->mem);
    if (S->table_own) (void)hipFree(S->table_own);
    delete S;
    return sfb::hip_fail(e, "sfb_mpc_swarm_create");
  }
  *swarm = S;
  return SFB_OK;
}

void sfb_mpc_swarm_destroy(sfb_mpc_swarm *swarm)
{
  if (!swarm) return;
  if (swarm->up_stream) { (void)hipStreamSynchronize(swarm->up_stream); (void)hipStreamDestroy(swarm->up_stream); }
  if (swarm->pinned) (void)hipHostFree(swarm->pinned);
  if (swarm->h_out) (void)hipHostFree(swarm->h_out);
  if (swarm->mem) (void)hipFree(swarm->mem);
  if (swarm->table_own) (void)hipFree(swarm->table_own);
  delete swarm;
}

namespace {
// the swarm's memory lives on the device that was current when it was created
sfb_status check_swarm_device(const sfb_mpc_swarm *S)
{
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return sfb::hip_fail(e, "hipGetDevice");
  if (dev != S->devid) return sfb::fail(SFB_ERR_INVALID_ARG, "swarm was created on another HIP device than the current one");
  return SFB_OK;
}
}  // namespace

sfb_status sfb_mpc_swarm_host_records(sfb_mpc_swarm *S, double **records)
{
  if (!S || !records) return sfb::fail(SFB_ERR_INVALID_ARG, "swarm / records is NULL");
  if (sfb_status sd = check_swarm_device(S); sd != SFB_OK) return sd;
  std::lock_guard<std::mutex> lk(S->mu);
  if (!S->pinned) {
    hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&S->pinned), (size_t)S->agents * (size_t)S->rec_full_doubles * 8, hipHostMallocDefault);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&S->up_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      if (S->pinned) (void)hipHostFree(S->pinned);
      S->pinned = nullptr;
      return sfb::hip_fail(e, "hipHostMalloc(swarm records)");
    }
    S->uploaded.assign((size_t)S->agents, 0);
  }
  *records = S->pinned;
  return SFB_OK;
}

sfb_status sfb_mpc_swarm_upload(sfb_mpc_swarm *S, int64_t first, int64_t count)
{
  if (!S) return sfb::fail(SFB_ERR_INVALID_ARG, "swarm is NULL");
  if (!S->pinned) return sfb::fail(SFB_ERR_INVALID_ARG, "sfb_mpc_swarm_host_records first");
  if (first < 0 || count < 0 || first + count > S->agents) return sfb::fail(SFB_ERR_INVALID_ARG, "range outside the swarm");
  // This entry point is meant to be called from the caller's linearisation threads, and a new host thread starts on
  // device 0 whatever the thread that created the swarm had selected: the swarm's device is made current for the copy
  // (it goes to the swarm's own stream and memory) and the caller's device is restored on the way out.
  struct DeviceGuard {
    int prev = -1;
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
  } guard;
  {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess && dev != S->devid) {
      e = hipSetDevice(S->devid);
      if (e == hipSuccess) guard.prev = dev;
    }
    if (e != hipSuccess) return sfb::hip_fail(e, "hipSetDevice(swarm device)");
  }
  if (count == 0) return SFB_OK;
  std::lock_guard<std::mutex> lk(S->mu);
  const size_t rd = (size_t)S->rec_own.rec_doubles;
  hipError_t e = hipMemcpyAsync(S->rec + (size_t)first * rd, S->pinned + (size_t)first * rd, (size_t)count * rd * 8, hipMemcpyHostToDevice,
                                S->up_stream);
  if (e != hipSuccess) return sfb::hip_fail(e, "hipMemcpyAsync(swarm records)");
  (void)hipStreamQuery(S->up_stream);  // flush: the copy must start now, not at the next synchronisation
  std::fill(S->uploaded.begin() + first, S->uploaded.begin() + first + count, (uint8_t)1);
  return SFB_OK;
}

sfb_status sfb_mpc_swarm_set_jac_keep(sfb_mpc_swarm *S, const uint8_t *jac_keep, int64_t *record_doubles)
{
  if (!S) return sfb::fail(SFB_ERR_INVALID_ARG, "swarm is NULL");
  if (jac_keep && S->rec_own.nu > 32) return sfb::fail(SFB_ERR_UNSUPPORTED, "jac_keep needs nu <= 32");
  std::lock_guard<std::mutex> lk(S->mu);
  if (S->up_stream) {  // nothing of the old layout may be in flight
    hipError_t e = hipStreamSynchronize(S->up_stream);
    if (e != hipSuccess) return sfb::hip_fail(e, "hipStreamSynchronize");
    std::fill(S->uploaded.begin(), S->uploaded.end(), (uint8_t)0);
  }
  set_packing(S->rec_own, false, jac_keep);
  if (record_doubles) *record_doubles = S->rec_own.rec_doubles;
  {  // the assembly table of the new record form (a synchronous copy on the null stream: behind the ticks launched so far)
    std::vector<sfb::MpcAsmDesc> tb((size_t)S->rec_own.nnzA);
    sfb::mpc_build_table(S->rec_own, false, tb.data());
    const hipError_t e = hipMemcpy(S->table_own, tb.data(), tb.size() * sizeof(sfb::MpcAsmDesc), hipMemcpyHostToDevice);
    if (e != hipSuccess) return sfb::hip_fail(e, "hipMemcpy(assembly table)");
  }
  return SFB_OK;
}

sfb_status sfb_mpc_swarm_reset_warmstart(sfb_mpc_swarm *S)
{
  if (!S) return sfb::fail(SFB_ERR_INVALID_ARG, "swarm is NULL");
  if (sfb_status sd = check_swarm_device(S); sd != SFB_OK) return sd;
  std::lock_guard<std::mutex> lk(S->mu);
  // an all-zero warm start IS the cold start of qp_solver.hpp:436-445 (x = y = 0 and z = A x = 0)
  hipError_t e = hipMemset(S->wx, 0, (size_t)S->agents * ((size_t)S->n + S->m) * 8);
  if (e != hipSuccess) return sfb::hip_fail(e, "hipMemset");
  S->have_order = false;
  return SFB_OK;
}

static sfb_status swarm_step_impl(sfb_mpc_swarm *S, const sfb_qp_params *prm, const double *records, const double *shared_jac,
                                  int warmstart, double *du0, uint32_t *iter, int32_t *code, double *primal, double *dual,
                                  const bool resident)
{
  if (!S) return sfb::fail(SFB_ERR_INVALID_ARG, "swarm is NULL");
  if (!prm || (!records && !resident) || !du0 || !code) return sfb::fail(SFB_ERR_INVALID_ARG, "NULL params / records / output pointer");
  if (sfb_status sd = check_swarm_device(S); sd != SFB_OK) return sd;
  std::lock_guard<std::mutex> lk(S->mu);
  const size_t B = (size_t)S->agents;
  const sfb::MpcAsmParams &p = shared_jac ? S->rec_shared : S->rec_own;
  hipError_t e = hipSuccess;
  sfb_status st = SFB_OK;
  // SFB_MPC_TIMING=1: synchronise after every stage and print the wall time of each (diagnostics only)
  const char *const tk = sfb::knob("SFB_MPC_TIMING");
  const bool timing    = tk && tk[0] == '1';
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto t0 = now();
  auto lap = [&](const char *what) {
    if (!timing) return;
    (void)hipDeviceSynchronize();
    const auto t1 = now();
    fprintf(stderr, "[sfb_mpc_swarm] %-10s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  };
  do {
    if (resident) {
      // the records are in the swarm's device buffer already (written there by the caller's kernel, on the null stream
      // or synchronised with it)
    } else if (records == S->pinned && !shared_jac) {
      // pipelined upload: ranges announced with sfb_mpc_swarm_upload are in flight on the copy stream; the rest go now
      const size_t rd = (size_t)p.rec_doubles;
      for (size_t b = 0; b < B && e == hipSuccess;) {
        if (S->uploaded[b]) { ++b; continue; }
        size_t b1 = b;
        while (b1 < B && !S->uploaded[b1]) ++b1;
        e = hipMemcpyAsync(S->rec + b * rd, S->pinned + b * rd, (b1 - b) * rd * 8, hipMemcpyHostToDevice, S->up_stream);
        b = b1;
      }
      if (e != hipSuccess) break;
      if ((e = hipStreamSynchronize(S->up_stream)) != hipSuccess) break;
      std::fill(S->uploaded.begin(), S->uploaded.end(), (uint8_t)0);
    } else if ((e = hipMemcpy(S->rec, records, B * (size_t)p.rec_doubles * 8, hipMemcpyHostToDevice)) != hipSuccess) break;
    if (shared_jac && (e = hipMemcpy(S->shared, shared_jac, (size_t)S->shared_doubles * 8, hipMemcpyHostToDevice)) != hipSuccess) break;
    lap("H2D");
    if ((e = sfb::mpc_assemble_launch(p, S->agents, S->rec, shared_jac ? S->shared : nullptr, S->Ax, S->l, S->u, nullptr,
                                      shared_jac ? S->table_shared : S->table_own)) != hipSuccess) break;
    lap("assemble");
    // Warm-started ticks: the agents that iterated longest in the previous tick are launched first (their counts
    // change little from tick to tick), so the stragglers overlap with the bulk of the batch.
    // The order is made HERE, from the counts the previous tick brought back, while the assembly kernel (and a device-side
    // linearisation in front of it) runs: at the end of the previous tick the same sort was 0.2 ms of a tick's wall time.
    const bool ordered = warmstart && S->have_order;
    if (ordered && !S->order_sorted) {
      sfb::order_by_count_descending(S->h_iter, S->h_order);
      S->order_sorted = true;
    }
    if (ordered && (e = hipMemcpy(S->order, S->h_order.data(), B * 4, hipMemcpyHostToDevice)) != hipSuccess) break;
    st = sfb_sparse_qp_solve_batch_ordered(S->plan, prm, S->agents, S->Px, S->q, S->Ax, S->l, S->u,
                                           warmstart ? S->wx : nullptr, warmstart ? S->wy : nullptr, S->x, S->y, nullptr,
                                           S->iter, S->code, S->ws, ordered ? S->order : nullptr, nullptr);
    if (st != SFB_OK) break;
    lap("solve");
    if ((e = sfb::mpc_store_launch(S->agents, S->n, S->m, S->uoff, S->nu, warmstart != 0, S->x, S->y, S->code, S->wx, S->wy,
                                   S->du0, nullptr)) != hipSuccess) break;
    {  // du0, iter and code lie one behind the other on the device: one copy into pinned memory instead of three into pageable
      const size_t nb_u = B * (size_t)S->nu * 8;
      if ((e = hipMemcpy(S->h_out, S->du0, nb_u + B * 8, hipMemcpyDeviceToHost)) != hipSuccess) break;
      std::memcpy(du0, S->h_out, nb_u);
      std::memcpy(S->h_iter.data(), S->h_out + nb_u, B * 4);
      std::memcpy(code, S->h_out + nb_u + B * 4, B * 4);
    }
    if (iter) std::memcpy(iter, S->h_iter.data(), B * 4);
    S->have_order   = true;  // (sorted at the start of the next tick, behind its first kernels)
    S->order_sorted = false;
    if (primal && (e = hipMemcpy(primal, S->x, B * (size_t)S->n * 8, hipMemcpyDeviceToHost)) != hipSuccess) break;
    if (dual && (e = hipMemcpy(dual, S->y, B * (size_t)S->m * 8, hipMemcpyDeviceToHost)) != hipSuccess) break;
    lap("store+D2H");
  } while (false);
  if (e != hipSuccess) st = sfb::hip_fail(e, "sfb_mpc_swarm_step_host");
  return st;
}

sfb_status sfb_mpc_swarm_step_host(sfb_mpc_swarm *S, const sfb_qp_params *prm, const double *records,
                                   const double *shared_jac, int warmstart, double *du0, uint32_t *iter,
                                   int32_t *code, double *primal, double *dual)
{
  return swarm_step_impl(S, prm, records, shared_jac, warmstart, du0, iter, code, primal, dual, false);
}

sfb_status sfb_mpc_swarm_step_resident(sfb_mpc_swarm *S, const sfb_qp_params *prm, int warmstart, double *du0, uint32_t *iter,
                                       int32_t *code, double *primal, double *dual)
{
  return swarm_step_impl(S, prm, nullptr, nullptr, warmstart, du0, iter, code, primal, dual, true);
}

sfb_status sfb_mpc_swarm_device_records(sfb_mpc_swarm *S, double **records, int64_t *record_doubles)
{
  if (!S || !records) return sfb::fail(SFB_ERR_INVALID_ARG, "swarm / records is NULL");
  *records = S->rec;
  if (record_doubles) *record_doubles = S->rec_own.rec_doubles;
  return SFB_OK;
}

sfb_status sfb_mpc_swarm_debug_buffers(sfb_mpc_swarm *S, const double **Ax, const double **l, const double **u)
{
  if (!S) return sfb::fail(SFB_ERR_INVALID_ARG, "swarm is NULL");
  if (Ax) *Ax = S->Ax;
  if (l) *l = S->l;
  if (u) *u = S->u;
  return SFB_OK;
}

}  // extern "C"
