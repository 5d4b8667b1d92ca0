// C-ABI for the shared-pattern sparse QP path (include/sfb.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/sfb.h"
#include "capi_common.h"
#include "qp_sparse_kernel.h"
#include "sparse_plan.h"

struct sfb_sparse_qp_plan {
  sfb::SparsePlanHost host;  // what the kernel works on: the caller's pattern, or (pruned plan) its kept entries only
  // Pruned plans (sfb_sparse_qp_plan_create_pruned): `full` is the caller's whole pattern analysed with the SAME
  // elimination order -- the plan of the fallback launch for items whose masked entries are not all zero.
  bool pruned = false;
  sfb::SparsePlanHost full;
  std::vector<int32_t> Aorig, Amasked;  // kept entry -> position in the caller's value array; masked positions
  std::mutex mu;
  struct DevCopy {
    int32_t *blob = nullptr, *blob_full = nullptr;
    sfb::SparsePlanDev dev{}, dev_full{};
  };
  std::map<int, DevCopy> per_device;  // device ordinal -> uploaded index arrays
  // Device memory of the host-pointer entry point, kept between calls like the working memory a
  // smooth::feedback::QPSolver object keeps between solves (qp_solver.hpp:242-338): a swarm that ticks at
  // a fixed batch size allocates its ~1.3 MB per agent once, not every tick.  Guarded by host_mu (calls
  // with host pointers on one plan AND one device are serialised; calls on different devices -- the shards of
  // sfb_sparse_qp_solve_batch_host_multi -- run side by side; device-pointer calls bring their own workspace).
  struct HostDev {
    std::mutex mu;
    std::pair<char *, size_t> ws{nullptr, 0};  // (buffer, bytes)
    int64_t batch = -1;                        // batch of the last call whose workspace content is still in the buffer
    uint64_t origin = 0;                       // who made that call: 0 = a direct host call, else the signature of a *_multi call
    // The stream the host entry launches on: its own, not synchronising with the null stream.  (Measured, round 5: once the
    // application has recorded timing events on the null stream, a solve launched there from this entry takes 60 instead of
    // 43 ms -- the runtime then tracks every dispatch of that queue; bench.py's event-timed loop did exactly that.)
    hipStream_t stream = nullptr;
    hipEvent_t uploaded = nullptr;  // recorded on the null stream behind the uploads; `stream` waits for it before the first launch
  };
  std::map<int, HostDev> host_dev;  // device ordinal -> state; entries are created under `mu` and never move
};

namespace sfb {
const SparsePlanHost &plan_host(const sfb_sparse_qp_plan *plan) { return plan->host; }
const SparsePlanHost &plan_io(const sfb_sparse_qp_plan *plan) { return plan->pruned ? plan->full : plan->host; }
}  // namespace sfb

namespace {

// set by sfb_sparse_qp_solve_batch_host_multi around its per-shard calls: signature of (device list, total batch)
thread_local uint64_t tl_multi_origin = 0;

// upload all index arrays of one host plan as one blob
sfb_status upload_plan(const sfb::SparsePlanHost &h, const std::vector<int32_t> *Aorig, const std::vector<int32_t> *Amasked,
                       int nnzA_io, int32_t **blob_out, sfb::SparsePlanDev &d)
{
  static const std::vector<int32_t> none;
  const std::vector<int32_t> *arrs[] = {&h.Pp, &h.Pi, &h.Pcol, &h.Ap, &h.Aj, &h.Arow, &h.Acp, &h.Aci, &h.Acpos,
                                        &h.Prp, &h.Prj, &h.Prpos, &h.Sp, &h.Sj, &h.Spos, &h.perm, &h.pinv,
                                        &h.Kp, &h.Ki, &h.Kdesc, &h.Lp, &h.Li, &h.Rp, &h.Rk, &h.Rpos, &h.Rlen,
                                        &h.fmap, &h.fidx, &h.bmap, &h.bidx, &h.Kmap, &h.rptr, &h.rtgt, &h.rab, &h.snptr, &h.snR, &h.poff, &h.pmap, &h.fmask, &h.bmask,
                                        &h.f2s, &h.seg, &h.pmapL, &h.rsplit, &h.KmapL, &h.KdescT, &h.KmapT, &h.ztop, &h.ustream, &h.utype, &h.uomap,
                                        Aorig ? Aorig : &none, Amasked ? Amasked : &none};
  constexpr int NA = sizeof(arrs) / sizeof(arrs[0]);
  size_t off[NA + 1];
  off[0] = 0;
  for (int a = 0; a < NA; ++a) off[a + 1] = off[a] + ((arrs[a]->size() + 3) / 4) * 4 + 4;  // 16-B aligned, never empty
  const size_t self_words = (sizeof(sfb::SparsePlanDev) + 15) / 16 * 4;               // the struct itself, at the end
  std::vector<int32_t> blob(off[NA] + self_words, 0);
  for (int a = 0; a < NA; ++a) std::copy(arrs[a]->begin(), arrs[a]->end(), blob.begin() + off[a]);
  int32_t *dblob = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&dblob), blob.size() * sizeof(int32_t));
  if (e != hipSuccess) return sfb::hip_fail(e, "hipMalloc(plan)");
  e = hipMemcpy(dblob, blob.data(), blob.size() * sizeof(int32_t), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(dblob);
    return sfb::hip_fail(e, "hipMemcpy(plan)");
  }
  d.n = h.n; d.m = h.m; d.k = h.k; d.nnzP = h.nnzP; d.nnzA = h.nnzA; d.nnzK = h.nnzK; d.nnzL = h.nnzL;
  const int32_t **ptrs[] = {&d.Pp, &d.Pi, &d.Pcol, &d.Ap, &d.Aj, &d.Arow, &d.Acp, &d.Aci, &d.Acpos,
                            &d.Prp, &d.Prj, &d.Prpos, &d.Sp, &d.Sj, &d.Spos, &d.perm, &d.pinv,
                            &d.Kp, &d.Ki, &d.Kdesc, &d.Lp, &d.Li, &d.Rp, &d.Rk, &d.Rpos, &d.Rlen,
                            &d.fmap, &d.fidx, &d.bmap, &d.bidx, &d.Kmap, &d.rptr, &d.rtgt, &d.rab, &d.snptr, &d.snR, &d.poff, &d.pmap, &d.fmask, &d.bmask,
                            &d.f2s, &d.seg, &d.pmapL, &d.rsplit, &d.KmapL, &d.KdescT, &d.KmapT, &d.ztop, &d.ustream, &d.utype, &d.uomap,
                            &d.Aorig, &d.Amasked};
  d.funits = h.funits; d.bunits = h.bunits; d.ffull0 = h.ffull0; d.ffull1 = h.ffull1; d.bfull0 = h.bfull0; d.bfull1 = h.bfull1; d.idx_scale = h.idx_scale; d.rsteps = h.rsteps; d.maxcol = h.maxcol; d.nsn = h.nsn; d.lds_doubles = h.lds_doubles; d.nseg = h.nseg; d.nnzKT = h.nnzKT; d.nztop = h.nztop; d.units = h.units; d.nunits = h.nunits;
  for (int a = 0; a < NA; ++a) *ptrs[a] = dblob + off[a];
  d.nnzA_io = nnzA_io;
  d.nmasked = Amasked ? (int)Amasked->size() - 512 : 0;  // without the padding
  if (!Aorig) d.Aorig = d.Amasked = nullptr;
  if ((e = sfb::sparse_device_words(&d.dev_active, &d.dev_busy)) != hipSuccess) {
    (void)hipFree(dblob);
    return sfb::hip_fail(e, "launch bookkeeping of the device");
  }
  d.self = reinterpret_cast<const sfb::SparsePlanDev *>(dblob + off[NA]);
  e      = hipMemcpy(dblob + off[NA], &d, sizeof(d), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(dblob);
    return sfb::hip_fail(e, "hipMemcpy(plan struct)");
  }
  *blob_out = dblob;
  return SFB_OK;
}

// device copies of the plan's index arrays (once per device)
sfb_status plan_on_device(sfb_sparse_qp_plan *plan, const sfb_sparse_qp_plan::DevCopy **out)
{
  int devid = 0;
  hipError_t e = hipGetDevice(&devid);
  if (e != hipSuccess) return sfb::hip_fail(e, "hipGetDevice");
  std::lock_guard<std::mutex> lk(plan->mu);
  auto it = plan->per_device.find(devid);
  if (it != plan->per_device.end()) {
    *out = &it->second;
    return SFB_OK;
  }
  sfb_sparse_qp_plan::DevCopy dc;
  sfb_status st;
  if (plan->pruned) {
    st = upload_plan(plan->host, &plan->Aorig, &plan->Amasked, plan->full.nnzA, &dc.blob, dc.dev);
    if (st != SFB_OK) return st;
    st = upload_plan(plan->full, nullptr, nullptr, plan->full.nnzA, &dc.blob_full, dc.dev_full);
    if (st != SFB_OK) {
      (void)hipFree(dc.blob);
      return st;
    }
  } else {
    st = upload_plan(plan->host, nullptr, nullptr, plan->host.nnzA, &dc.blob, dc.dev);
    if (st != SFB_OK) return st;
  }
  auto ins = plan->per_device.emplace(devid, dc);
  *out     = &ins.first->second;
  return SFB_OK;
}

// Workspace of one call: batch items of the plan's per-item block; for a pruned plan a pool of whole-pattern slots
// (items that violate the mask are solved there, inside the same launch); the launch's auxiliary memory (queue of a
// time-sliced launch, pool flags) at the end.
struct WsLayout {
  size_t item_bytes, main_bytes, pool_off, aux_off, total;
};
WsLayout ws_layout(const sfb_sparse_qp_plan *plan, int64_t batch)
{
  const sfb::SparsePlanHost &h = plan->host;
  WsLayout L{};
  L.item_bytes = sfb::qp_sparse_ws_doubles(h.n, h.m, h.nnzL, h.funits, h.bunits, plan->pruned ? h.nnzA : 0) * sizeof(double);
  L.main_bytes = (size_t)batch * L.item_bytes;
  L.pool_off   = L.main_bytes;
  size_t pool  = 0;
  if (plan->pruned) {
    const sfb::SparsePlanHost &f = plan->full;
    // pool of whole-pattern slots: min(batch, 64); the kernel is told the same count and probes the flags modulo it
    pool = (size_t)std::min<int64_t>(batch, sfb::qp_sparse_fallback_slots()) *
           sfb::qp_sparse_ws_doubles(f.n, f.m, f.nnzL, f.funits, f.bunits, 0) * sizeof(double);
  }
  L.aux_off = L.pool_off + pool;
  L.total   = L.aux_off + (sfb::qp_sparse_aux_bytes(batch) + 15) / 16 * 16;
  return L;
}

sfb_status check_sparse_args(const sfb_sparse_qp_plan *plan, const sfb_qp_params *prm, int64_t batch, const void *Px,
                             const void *q, const void *Ax, const void *l, const void *u, const void *wx,
                             const void *wy, const void *x, const void *y, const void *code)
{
  if (!plan) return sfb::fail(SFB_ERR_INVALID_ARG, "plan is NULL");
  if (!prm) return sfb::fail(SFB_ERR_INVALID_ARG, "prm is NULL");
  if (batch < 0) return sfb::fail(SFB_ERR_INVALID_ARG, "batch < 0");
  if (batch > 0 && (!q || !l || !u || !x || !y || !code || (plan->host.nnzP > 0 && !Px) || (sfb::plan_io(plan).nnzA > 0 && !Ax)))
    return sfb::fail(SFB_ERR_INVALID_ARG, "NULL problem / solution pointer");
  if ((wx == nullptr) != (wy == nullptr))
    return sfb::fail(SFB_ERR_INVALID_ARG, "warm_x and warm_y must both be given or both be NULL");
  if (prm->max_iter > 0xFFFFFFFFll) return sfb::fail(SFB_ERR_INVALID_ARG, "max_iter exceeds uint32");
  if (batch > 0x7FFFFFFFll) return sfb::fail(SFB_ERR_UNSUPPORTED, "batch exceeds 2^31-1 per call");
  if ((size_t)std::max(plan->host.lds_doubles, plan->pruned ? plan->full.lds_doubles : 0) * sizeof(double) > 150 * 1024)
    return sfb::fail(SFB_ERR_UNSUPPORTED, "n+m too large for the LDS-resident work vector (max 19200)");
  return SFB_OK;
}

}  // namespace

extern "C" {

sfb_status sfb_sparse_qp_plan_create(int n, int m, const int32_t *P_colptr, const int32_t *P_rowind,
                                     const int32_t *A_rowptr, const int32_t *A_colind, int ordering,
                                     const int32_t *user_perm, sfb_sparse_qp_plan **plan)
{
  return sfb_sparse_qp_plan_create_staged(n, m, P_colptr, P_rowind, A_rowptr, A_colind, ordering, user_perm, nullptr,
                                          plan);
}

sfb_status sfb_sparse_qp_plan_create_staged(int n, int m, const int32_t *P_colptr, const int32_t *P_rowind,
                                            const int32_t *A_rowptr, const int32_t *A_colind, int ordering,
                                            const int32_t *user_perm, const int32_t *stage,
                                            sfb_sparse_qp_plan **plan)
{
  return sfb_sparse_qp_plan_create_pruned(n, m, P_colptr, P_rowind, A_rowptr, A_colind, ordering, user_perm, stage, nullptr,
                                          plan);
}

sfb_status sfb_sparse_qp_plan_create_pruned(int n, int m, const int32_t *P_colptr, const int32_t *P_rowind,
                                            const int32_t *A_rowptr, const int32_t *A_colind, int ordering,
                                            const int32_t *user_perm, const int32_t *stage, const uint8_t *A_keep,
                                            sfb_sparse_qp_plan **plan)
{
  if (!plan) return sfb::fail(SFB_ERR_INVALID_ARG, "plan out-pointer is NULL");
  *plan = nullptr;
  auto *p = new (std::nothrow) sfb_sparse_qp_plan();
  if (!p) return sfb::fail(SFB_ERR_INVALID_ARG, "out of memory");
  const char *msg = "";
  bool any_masked = false;
  if (A_keep && n >= 1 && m >= 1 && A_rowptr && A_rowptr[0] == 0 && A_rowptr[m] >= 0) {
    bool monotone = true;  // A_keep has A_rowptr[m] entries only if the row pointers are what they claim to be
    for (int r = 0; r < m && monotone; ++r) monotone = A_rowptr[r + 1] >= A_rowptr[r] && A_rowptr[r + 1] <= A_rowptr[m];
    for (int e = 0; monotone && e < A_rowptr[m] && !any_masked; ++e) any_masked = A_keep[e] == 0;
  }
  if (!any_masked) {
    if (!sfb::build_sparse_plan(n, m, P_colptr, P_rowind, A_rowptr, A_colind, ordering, user_perm, stage, p->host, &msg)) {
      delete p;
      return sfb::fail(SFB_ERR_INVALID_ARG, msg);
    }
    *plan = p;
    return SFB_OK;
  }
  // compressed pattern of A: the kept entries, in storage order.  (The pattern is read here BEFORE build_sparse_plan
  // validates it, so the checks the loop below relies on come first.)
  const int nnzA = A_rowptr[m];
  if (nnzA > 0 && !A_colind) {
    delete p;
    return sfb::fail(SFB_ERR_INVALID_ARG, "A_colind is NULL");
  }
  std::vector<int32_t> Ap(m + 1, 0), Aj;
  Aj.reserve(nnzA);
  for (int r = 0; r < m; ++r) {
    if (A_rowptr[r + 1] < A_rowptr[r] || A_rowptr[r + 1] > nnzA) {
      delete p;
      return sfb::fail(SFB_ERR_INVALID_ARG, "A row pointers not monotone");
    }
    for (int e = A_rowptr[r]; e < A_rowptr[r + 1]; ++e)
      if (A_colind[e] < 0 || A_colind[e] >= n) {
        delete p;
        return sfb::fail(SFB_ERR_INVALID_ARG, "A column index out of range");
      } else if (A_keep[e]) {
        Aj.push_back(A_colind[e]);
        p->Aorig.push_back(e);
      } else {
        p->Amasked.push_back(e);
      }
    Ap[r + 1] = (int32_t)Aj.size();
  }
  // kernel plan on the kept entries; fallback plan on the whole pattern with the SAME elimination order, so that
  // an item's result does not depend on which of the two solved it (up to the sign of zeros)
  bool ok = sfb::build_sparse_plan(n, m, P_colptr, P_rowind, Ap.data(), Aj.data(), ordering, user_perm, stage, p->host, &msg);
  if (ok) ok = sfb::build_sparse_plan(n, m, P_colptr, P_rowind, A_rowptr, A_colind, ordering, p->host.perm.data(), nullptr, p->full, &msg, p->host.lds_doubles);
  if (!ok) {
    delete p;
    return sfb::fail(SFB_ERR_INVALID_ARG, msg);
  }
  p->pruned = true;
  // padding for the kernel's branch-free batches: repeat a valid position
  const int32_t lastk = p->Aorig.empty() ? 0 : p->Aorig.back(), lastm = p->Amasked.back();
  p->Aorig.resize(p->Aorig.size() + 512, lastk);
  p->Amasked.resize(p->Amasked.size() + 512, lastm);
  *plan = p;
  return SFB_OK;
}

void sfb_sparse_qp_plan_destroy(sfb_sparse_qp_plan *plan)
{
  if (!plan) return;
  for (auto &kv : plan->per_device) {
    if (kv.second.blob) (void)hipFree(kv.second.blob);
    if (kv.second.blob_full) (void)hipFree(kv.second.blob_full);
  }
  for (auto &kv : plan->host_dev) {
    if (kv.second.ws.first) (void)hipFree(kv.second.ws.first);
    if (kv.second.stream) (void)hipStreamDestroy(kv.second.stream);
    if (kv.second.uploaded) (void)hipEventDestroy(kv.second.uploaded);
  }
  delete plan;
}

sfb_status sfb_sparse_qp_plan_info(const sfb_sparse_qp_plan *plan, int64_t *nnzK, int64_t *nnzL,
                                   int64_t *workspace_bytes_per_item)
{
  if (!plan) return sfb::fail(SFB_ERR_INVALID_ARG, "plan is NULL");
  if (nnzK) *nnzK = plan->host.nnzK;
  if (nnzL) *nnzL = plan->host.nnzL;
  if (workspace_bytes_per_item) {
    // batch * this many bytes always suffice; sfb_sparse_qp_plan_workspace_bytes is exact (a pruned plan's pool of
    // whole-pattern slots and the launch's auxiliary memory do not grow with the batch beyond 64 items)
    const WsLayout L1 = ws_layout(plan, 1);
    *workspace_bytes_per_item = (int64_t)L1.total;
  }
  return SFB_OK;
}

sfb_status sfb_sparse_qp_plan_workspace_bytes(const sfb_sparse_qp_plan *plan, int64_t batch, int64_t *bytes)
{
  if (!plan || !bytes) return sfb::fail(SFB_ERR_INVALID_ARG, "NULL argument");
  if (batch < 0) return sfb::fail(SFB_ERR_INVALID_ARG, "batch < 0");
  *bytes = (int64_t)ws_layout(plan, batch).total;
  return SFB_OK;
}

sfb_status sfb_sparse_qp_plan_pruned_info(const sfb_sparse_qp_plan *plan, int64_t *nnzA_kept, int64_t *nnzL_fallback)
{
  if (!plan) return sfb::fail(SFB_ERR_INVALID_ARG, "plan is NULL");
  if (nnzA_kept) *nnzA_kept = plan->host.nnzA;
  if (nnzL_fallback) *nnzL_fallback = plan->pruned ? plan->full.nnzL : plan->host.nnzL;
  return SFB_OK;
}

sfb_status sfb_sparse_qp_plan_get_perm(const sfb_sparse_qp_plan *plan, int32_t *perm)
{
  if (!plan || !perm) return sfb::fail(SFB_ERR_INVALID_ARG, "NULL argument");
  std::copy(plan->host.perm.begin(), plan->host.perm.end(), perm);
  return SFB_OK;
}

sfb_status sfb_sparse_qp_plan_get_factor_order(const sfb_sparse_qp_plan *plan, int fallback, int32_t *rank)
{
  if (!plan || !rank) return sfb::fail(SFB_ERR_INVALID_ARG, "NULL argument");
  const sfb::SparsePlanHost &h = (fallback && plan->pruned) ? plan->full : plan->host;
  for (int f = 0; f < h.k; ++f) rank[h.f2s[f]] = f;
  return SFB_OK;
}

namespace {
sfb_status solve_batch_impl(sfb_sparse_qp_plan *plan, const sfb_qp_params *prm, int64_t batch,
                                     const double *Px, const double *q, const double *Ax, const double *l,
                                     const double *u, const double *warm_x, const double *warm_y, double *x,
                                     double *y, double *obj, uint32_t *iter, int32_t *code, void *workspace,
                                             const int32_t *order, void *stream, double *trace, int32_t trace_rows,
                                             double *phase_us = nullptr)
{
  sfb_status st = check_sparse_args(plan, prm, batch, Px, q, Ax, l, u, warm_x, warm_y, x, y, code);
  if (st != SFB_OK) return st;
  if (batch > 0 && !workspace) return sfb::fail(SFB_ERR_INVALID_ARG, "workspace is NULL");
  st = sfb::require_device();
  if (st != SFB_OK) return st;
  if (batch == 0) return SFB_OK;
  const sfb_sparse_qp_plan::DevCopy *dc = nullptr;
  st = plan_on_device(plan, &dc);
  if (st != SFB_OK) return st;
  const sfb::DenseKernelParams kp = sfb::make_kernel_params(prm, plan->host.n, plan->host.m);
  hipStream_t hs   = static_cast<hipStream_t>(stream);
  const WsLayout L = ws_layout(plan, batch);
  char *wsc        = static_cast<char *>(workspace);
  hipError_t e     = sfb::qp_sparse_launch(dc->dev, kp, batch, Px, q, Ax, l, u, warm_x, warm_y, x, y, obj, iter, code,
                                           reinterpret_cast<double *>(wsc), hs, order, reinterpret_cast<int32_t *>(wsc + L.aux_off),
                                           plan->pruned ? &dc->dev_full : nullptr, reinterpret_cast<double *>(wsc + L.pool_off),
                                           trace, (int)trace_rows, phase_us);
  if (e != hipSuccess) return sfb::hip_fail(e, "qp_sparse_kernel launch");
  return SFB_OK;
}
}  // namespace

sfb_status sfb_sparse_qp_solve_batch_ordered(sfb_sparse_qp_plan *plan, const sfb_qp_params *prm, int64_t batch,
                                             const double *Px, const double *q, const double *Ax, const double *l,
                                             const double *u, const double *warm_x, const double *warm_y, double *x,
                                             double *y, double *obj, uint32_t *iter, int32_t *code, void *workspace,
                                             const int32_t *order, void *stream)
{
  return solve_batch_impl(plan, prm, batch, Px, q, Ax, l, u, warm_x, warm_y, x, y, obj, iter, code, workspace, order, stream,
                          nullptr, 0);
}

sfb_status sfb_sparse_qp_solve_batch_trace(sfb_sparse_qp_plan *plan, const sfb_qp_params *prm, int64_t batch,
                                           const double *Px, const double *q, const double *Ax, const double *l,
                                           const double *u, const double *warm_x, const double *warm_y, double *x,
                                           double *y, double *obj, uint32_t *iter, int32_t *code, void *workspace,
                                           double *trace, int32_t trace_rows, void *stream)
{
  if (!trace || trace_rows <= 0) return sfb::fail(SFB_ERR_INVALID_ARG, "trace is NULL or trace_rows <= 0");
  return solve_batch_impl(plan, prm, batch, Px, q, Ax, l, u, warm_x, warm_y, x, y, obj, iter, code, workspace, nullptr, stream,
                          trace, trace_rows);
}

sfb_status sfb_sparse_qp_solve_batch_phases(sfb_sparse_qp_plan *plan, const sfb_qp_params *prm, int64_t batch,
                                            const double *Px, const double *q, const double *Ax, const double *l,
                                            const double *u, const double *warm_x, const double *warm_y, double *x,
                                            double *y, double *obj, uint32_t *iter, int32_t *code, void *workspace,
                                            double *trace, int32_t trace_rows, double *phase_us, void *stream)
{
  if (!phase_us) return sfb::fail(SFB_ERR_INVALID_ARG, "phase_us is NULL");
  if (trace != nullptr && trace_rows <= 0) return sfb::fail(SFB_ERR_INVALID_ARG, "trace_rows <= 0");
  return solve_batch_impl(plan, prm, batch, Px, q, Ax, l, u, warm_x, warm_y, x, y, obj, iter, code, workspace, nullptr, stream,
                          trace, trace ? trace_rows : 0, phase_us);
}

sfb_status sfb_sparse_qp_solve_batch(sfb_sparse_qp_plan *plan, const sfb_qp_params *prm, int64_t batch,
                                     const double *Px, const double *q, const double *Ax, const double *l,
                                     const double *u, const double *warm_x, const double *warm_y, double *x,
                                     double *y, double *obj, uint32_t *iter, int32_t *code, void *workspace,
                                     void *stream)
{
  return sfb_sparse_qp_solve_batch_ordered(plan, prm, batch, Px, q, Ax, l, u, warm_x, warm_y, x, y, obj, iter, code,
                                           workspace, nullptr, stream);
}

sfb_status sfb_sparse_qp_solve_batch_host(sfb_sparse_qp_plan *plan, const sfb_qp_params *prm, int64_t batch,
                                          const double *Px, const double *q, const double *Ax, const double *l,
                                          const double *u, const double *warm_x, const double *warm_y, double *x,
                                          double *y, double *obj, uint32_t *iter, int32_t *code)
{
  return sfb_sparse_qp_solve_batch_host_trace(plan, prm, batch, Px, q, Ax, l, u, warm_x, warm_y, x, y, obj, iter, code, nullptr, 0);
}

sfb_status sfb_sparse_qp_solve_batch_host_trace(sfb_sparse_qp_plan *plan, const sfb_qp_params *prm, int64_t batch,
                                                const double *Px, const double *q, const double *Ax, const double *l,
                                                const double *u, const double *warm_x, const double *warm_y, double *x,
                                                double *y, double *obj, uint32_t *iter, int32_t *code, double *trace,
                                                int32_t trace_rows)
{
  return sfb_sparse_qp_solve_batch_host_phases(plan, prm, batch, Px, q, Ax, l, u, warm_x, warm_y, x, y, obj, iter, code, trace,
                                               trace_rows, nullptr);
}

sfb_status sfb_sparse_qp_solve_batch_host_phases(sfb_sparse_qp_plan *plan, const sfb_qp_params *prm, int64_t batch,
                                                 const double *Px, const double *q, const double *Ax, const double *l,
                                                 const double *u, const double *warm_x, const double *warm_y, double *x,
                                                 double *y, double *obj, uint32_t *iter, int32_t *code, double *trace,
                                                 int32_t trace_rows, double *phase_us)
{
  sfb_status st = check_sparse_args(plan, prm, batch, Px, q, Ax, l, u, warm_x, warm_y, x, y, code);
  if (st == SFB_OK && trace != nullptr && trace_rows <= 0) st = sfb::fail(SFB_ERR_INVALID_ARG, "trace_rows <= 0");
  // verbose for ONE problem (the reference's use of the flag: one QPSolver object, qp_solver.hpp:409-420, :490-501, :550-565):
  // the per-iteration table and the per-phase times are collected on the device and printed below
  std::vector<double> vtrace, vphase;
  if (st == SFB_OK && prm->verbose && batch == 1 && trace == nullptr) {
    trace_rows = sfb::verbose_table_rows(prm);
    vtrace.assign((size_t)trace_rows * 5, 0.0);
    trace = vtrace.data();
  }
  if (st == SFB_OK && prm->verbose && batch == 1 && phase_us == nullptr) {
    vphase.assign(6, 0.0);
    phase_us = vphase.data();
  }
  if (st != SFB_OK) return st;
  st = sfb::require_device();
  if (st != SFB_OK) return st;
  if (batch == 0) return SFB_OK;
  const sfb::SparsePlanHost &h = sfb::plan_io(plan);  // the caller's pattern (strides of the value arrays)
  const size_t B = (size_t)batch, N = (size_t)h.n, M = (size_t)h.m, NP = (size_t)h.nnzP, NA = (size_t)h.nnzA;
  const size_t wsb  = ws_layout(plan, batch).total;  // multiple of 8
  const size_t TR = (trace ? B * (size_t)trace_rows * 5 : 0) + (phase_us ? B * 6 : 0);  // the table, then the phase times
  const size_t TR0 = trace ? B * (size_t)trace_rows * 5 : 0;
  const size_t in_d = B * (NP + N + NA + 2 * M) + (warm_x ? B * (N + M) : 0), out_d = B * (N + M + 1) + TR;
  const size_t bytes = (in_d + out_d) * sizeof(double) + wsb + B * 8;
  int devid    = 0;
  hipError_t e = hipGetDevice(&devid);
  if (e != hipSuccess) return sfb::hip_fail(e, "hipGetDevice");
  sfb_sparse_qp_plan::HostDev *hd = nullptr;
  {
    std::lock_guard<std::mutex> lk(plan->mu);
    hd = &plan->host_dev[devid];
  }
  std::lock_guard<std::mutex> host_lock(hd->mu);
  auto &cache = hd->ws;
  if (cache.second < bytes) {  // grow-only
    if (cache.first) (void)hipFree(cache.first);
    cache = {nullptr, 0};
    e     = hipMalloc(reinterpret_cast<void **>(&cache.first), bytes);
    if (e != hipSuccess) {
      cache = {nullptr, 0};
      return sfb::hip_fail(e, "hipMalloc");
    }
    cache.second = bytes;
    hd->batch    = -1;
  }
  if (hd->stream == nullptr && (e = hipStreamCreateWithFlags(&hd->stream, hipStreamNonBlocking)) != hipSuccess) {
    hd->stream = nullptr;
    return sfb::hip_fail(e, "hipStreamCreateWithFlags");
  }
  if (hd->uploaded == nullptr && (e = hipEventCreateWithFlags(&hd->uploaded, hipEventDisableTiming)) != hipSuccess) {
    hd->uploaded = nullptr;
    return sfb::hip_fail(e, "hipEventCreateWithFlags");
  }
  // reuse_factor refers to "the previous call on this workspace": the workspace sits at the start of the cached
  // buffer, so it is the same memory, item for item, exactly when the batch size is that of the previous call
  // -- and the call comes from where the previous one came from: a shard of a *_multi call holds the items its device
  // list and total batch assign to this device, a direct call the caller's; equal shard SIZES alone do not make the
  // items the same (the kernel re-checks c and rho only, which agree across the agents of an MPC swarm).
  sfb_qp_params prm_call = *prm;
  if (hd->batch != batch || hd->origin != tl_multi_origin) prm_call.reuse_factor = 0;
  hd->batch  = batch;
  hd->origin = tl_multi_origin;
  prm = &prm_call;
  char *devmem = cache.first;
  double *dws = reinterpret_cast<double *>(devmem);
  double *dPx = reinterpret_cast<double *>(devmem + wsb);
  double *dq = dPx + B * NP, *dAx = dq + B * N, *dl = dAx + B * NA, *du = dl + B * M;
  double *dwx = nullptr, *dwy = nullptr, *dx = du + B * M;
  if (warm_x) { dwx = dx; dwy = dwx + B * N; dx = dwy + B * M; }
  double *dy = dx + B * N, *dobj = dy + B * M, *dtrace = dobj + B;
  uint32_t *dit  = reinterpret_cast<uint32_t *>(dtrace + TR);
  int32_t *dcode = reinterpret_cast<int32_t *>(dit + B);
  auto H2D = [&](void *d, const void *hh, size_t nb) { return nb ? hipMemcpy(d, hh, nb, hipMemcpyHostToDevice) : hipSuccess; };
  auto D2H = [&](void *hh, const void *d, size_t nb) { return nb ? hipMemcpy(hh, d, nb, hipMemcpyDeviceToHost) : hipSuccess; };
  using clk = std::chrono::steady_clock;
  auto ms   = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto tv0 = clk::now();
  auto tv1 = tv0, tv2 = tv0;
  do {
    if ((e = H2D(dPx, Px, B * NP * 8)) != hipSuccess) break;
    if ((e = H2D(dq, q, B * N * 8)) != hipSuccess) break;
    if ((e = H2D(dAx, Ax, B * NA * 8)) != hipSuccess) break;
    if ((e = H2D(dl, l, B * M * 8)) != hipSuccess) break;
    if ((e = H2D(du, u, B * M * 8)) != hipSuccess) break;
    if (warm_x) {
      if ((e = H2D(dwx, warm_x, B * N * 8)) != hipSuccess) break;
      if ((e = H2D(dwy, warm_y, B * M * 8)) != hipSuccess) break;
    }
    if (trace) {  // unused rows keep ITER = -1
      for (size_t r = 0; r < TR0; ++r) trace[r] = (r % 5 == 0) ? -1.0 : 0.0;
      if ((e = H2D(dtrace, trace, TR0 * 8)) != hipSuccess) break;
    }
    tv1 = clk::now();
    // The uploads are synchronous copies on the null stream, the launches go to a stream that does not synchronise with it:
    // a copy out of pageable memory may return once its data is STAGED, so the launch stream is made to wait for the null
    // stream's work up to here explicitly.
    if ((e = hipEventRecord(hd->uploaded, nullptr)) != hipSuccess) break;
    if ((e = hipStreamWaitEvent(hd->stream, hd->uploaded, 0)) != hipSuccess) break;
    st = solve_batch_impl(plan, prm, batch, dPx, dq, dAx, dl, du, dwx, dwy, dx, dy, dobj, dit, dcode, dws, nullptr, hd->stream,
                          trace ? dtrace : nullptr, trace ? trace_rows : 0, phase_us ? dtrace + TR0 : nullptr);
    if (st != SFB_OK) {
      (void)hipStreamSynchronize(hd->stream);
      break;
    }
    if ((e = hipStreamSynchronize(hd->stream)) != hipSuccess) break;
    tv2 = clk::now();
    if ((e = D2H(x, dx, B * N * 8)) != hipSuccess) break;
    if ((e = D2H(y, dy, B * M * 8)) != hipSuccess) break;
    if (obj && (e = D2H(obj, dobj, B * 8)) != hipSuccess) break;
    if (iter && (e = D2H(iter, dit, B * 4)) != hipSuccess) break;
    if ((e = D2H(code, dcode, B * 4)) != hipSuccess) break;
    if (trace && (e = D2H(trace, dtrace, TR0 * 8)) != hipSuccess) break;
    if (phase_us && (e = D2H(phase_us, dtrace + TR0, B * 6 * 8)) != hipSuccess) break;
  } while (false);
  if (e != hipSuccess) st = sfb::hip_fail(e, "sfb_sparse_qp_solve_batch_host");
  if (st == SFB_OK && !vtrace.empty()) {  // the table of qp_solver.hpp:409-420, :490-501 (TIME: device clock, microseconds)
    sfb::verbose_table("sparse", h.n, h.m, vtrace.data(), trace_rows);
  }
  if (st == SFB_OK && !vphase.empty()) {  // the summary of qp_solver.hpp:550-565
    uint32_t it1 = 0;
    if (iter) it1 = iter[0];
    else if (hipMemcpy(&it1, dit, 4, hipMemcpyDeviceToHost) != hipSuccess) (void)hipGetLastError();
    sfb::verbose_summary(code[0], it1, vphase.data());
  }
  if (st == SFB_OK && prm->verbose) {
    std::vector<uint32_t> itv;
    if (!iter) {
      itv.resize(B);
      if (hipMemcpy(itv.data(), dit, B * 4, hipMemcpyDeviceToHost) != hipSuccess) itv.clear();
    }
    std::printf("[sfb] sparse plan: nnz(K) %d, nnz(L) %d%s\n", plan->host.nnzK, plan->host.nnzL,
                plan->pruned ? " (explicit zeros of A left out; whole-pattern fallback per item)" : "");
    sfb::verbose_report("sparse QP batch", batch, h.n, h.m, ms(tv0, tv1), ms(tv1, tv2), ms(tv2, clk::now()), code,
                        iter ? iter : (itv.empty() ? nullptr : itv.data()));
  }
  return st;
}


sfb_status sfb_sparse_qp_solve_batch_host_multi(sfb_sparse_qp_plan *plan, const sfb_qp_params *prm, int64_t batch,
                                                const double *Px, const double *q, const double *Ax, const double *l,
                                                const double *u, const double *warm_x, const double *warm_y, double *x,
                                                double *y, double *obj, uint32_t *iter, int32_t *code)
{
  sfb_status st = check_sparse_args(plan, prm, batch, Px, q, Ax, l, u, warm_x, warm_y, x, y, code);
  if (st != SFB_OK) return st;
  if (batch == 0) return sfb::require_device();
  const sfb::SparsePlanHost &h = sfb::plan_io(plan);
  const size_t N = (size_t)h.n, M = (size_t)h.m, NP = (size_t)h.nnzP, NA = (size_t)h.nnzA;
  // reuse_factor refers to "the previous call on this device's workspace": with a device listed twice two shards
  // share one workspace, and the claim cannot be kept
  sfb_qp_params prm_call = *prm;
  uint64_t origin = 1469598103934665603ull;  // FNV-1a over (device list, batch): which items a device's shard holds
  {
    std::vector<int> d = sfb::device_list();
    for (int v : d) origin = (origin ^ (uint64_t)(uint32_t)v) * 1099511628211ull;
    origin = (origin ^ (uint64_t)batch) * 1099511628211ull;
    if (origin == 0) origin = 1;
    std::sort(d.begin(), d.end());
    if (std::adjacent_find(d.begin(), d.end()) != d.end()) prm_call.reuse_factor = 0;
  }
  prm_call.verbose = 0;  // (reported once for the whole call below, not once per shard)
  const sfb_qp_params *const prm_shard = &prm_call;
  const auto t_call = std::chrono::steady_clock::now();
  const sfb_status rs = sfb::run_sharded(batch, [&](int, int64_t b0, int64_t cnt) {
    const size_t o = (size_t)b0;  // the plan uploads its tables to a device on first use; each device has its own workspace
    tl_multi_origin = origin;
    const sfb_status s1 = sfb_sparse_qp_solve_batch_host(plan, prm_shard, cnt, Px ? Px + o * NP : nullptr, q + o * N, Ax ? Ax + o * NA : nullptr,
                                          l + o * M, u + o * M, warm_x ? warm_x + o * N : nullptr,
                                          warm_y ? warm_y + o * M : nullptr, x + o * N, y + o * M, obj ? obj + o : nullptr,
                                          iter ? iter + o : nullptr, code + o);
    tl_multi_origin = 0;
    return s1;
  });
  if (rs == SFB_OK && prm->verbose)
    sfb::verbose_report("sparse QP batch (sharded over the device list)", batch, h.n, h.m, -1.0,
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count(), -1.0, code, iter);
  return rs;
}

}  // extern "C"
