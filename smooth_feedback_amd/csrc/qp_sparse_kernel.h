// Host<->kernel interface of the sparse (shared-pattern) QP kernel.  Internal.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "qp_dense_kernel.h"

namespace sfb {

#ifndef SFB_KKT_KINDS
#define SFB_KKT_KINDS
enum { K_P = 0, K_A = 1, K_SIGMA = 2, K_RHO = 3 };  // source of a KKT entry (qp_solver.hpp:385,:392,:389,:395)
#endif

// Device copies of the plan's index arrays (all int32, read-only, shared by the whole batch).
struct SparsePlanDev {
  int n, m, k, nnzP, nnzA, nnzK, nnzL;
  const int32_t *Pp, *Pi, *Pcol, *Ap, *Aj, *Arow;
  const int32_t *Acp, *Aci, *Acpos, *Prp, *Prj, *Prpos, *Sp, *Sj, *Spos;
  const int32_t *perm, *pinv, *Kp, *Ki, *Kdesc, *Lp, *Li, *Rp, *Rk, *Rpos, *Rlen;
  const int32_t *fmap, *fidx, *bmap, *bidx;  // packed sweep schedules, see sparse_plan.h
  const int32_t *fmask, *bmask;              // lane-mask shifts of the units' value loads (one word per unit)
  int funits, bunits, idx_scale, ffull0, ffull1, bfull0, bfull1;
  const int32_t *Kmap, *rptr, *rtgt, *rab;   // right-looking factorisation schedule
  int rsteps, maxcol;
  const int32_t *snptr, *snR, *poff, *pmap;  // (relaxed) supernodes of the factorisation and their panel maps
  int nsn, lds_doubles;
  // Pruned plans (sfb_sparse_qp_plan_create_pruned): the arrays above describe the COMPRESSED pattern of A (kept
  // entries only, nnzA of them); the caller's value array still has nnzA_io entries per item.  Aorig[p] = position
  // of kept entry p in the caller's array, Amasked[0..nmasked) = positions of the entries declared zero (padded by
  // 512 entries that repeat the last one).  Aorig == nullptr: no mask, the kernel reads the caller's array directly.
  // factorisation numbering and LDS-resident subtrees (sparse_plan.h)
  const int32_t *f2s, *seg, *pmapL, *rsplit, *KmapL, *KdescT, *KmapT, *ztop;
  int nseg, nnzKT, nztop;
  const int32_t *ustream, *utype, *uomap;  // unit engine of the factorisation (sparse_plan.h); units == 0: supernodal engine
  int units, nunits;
  int nnzA_io, nmasked;
  const int32_t *Aorig, *Amasked;
  // Launch bookkeeping of the DEVICE the copy lives on (sparse_device_words below; owned by the library per device, shared
  // by every plan): [0] of dev_active = waves of the sparse kernel busy with an item right now (all launches on the device:
  // masked or plain factor loads); *dev_busy = one bit per caller stream that has a solve enqueued and not known finished,
  // kept by the HOST at enqueue time in mapped host memory (the helpers of a loop launch leave when another stream has work).
  int32_t *dev_active;
  const unsigned long long *dev_busy;
  const SparsePlanDev *self;  // device copy of this struct (what the kernel is handed)
};

// The two words above for the current device (created on first use, never freed).
hipError_t sparse_device_words(int32_t **active, const unsigned long long **busy);

// per-item workspace, in doubles
constexpr int kSweepPadDev = 16;  // == SparsePlanHost::kSweepPad
#ifdef __HIPCC__
#define SFB_HD __host__ __device__
#else
#define SFB_HD
#endif
// doubles of one copy of the numeric factor: [L values | D | scratch | zero | forward-sweep copy | backward-sweep copy | 1/D]
SFB_HD inline size_t qp_sparse_factor_doubles(int n, int m, int nnzL, int funits, int bunits)
{
  return (size_t)nnzL + 2 * ((size_t)n + m) + 2 + (size_t)(funits + bunits + 2 * kSweepPadDev) * 128;
}
// offset of the second factor block (even, like the first one's)
SFB_HD inline size_t qp_sparse_polish_offset(int n, int m, int nnzL, int funits, int bunits, int nnzA_compact)
{
  const size_t k   = (size_t)n + m;
  const size_t off = qp_sparse_factor_doubles(n, m, nnzL, funits, bunits) + k + 6 * (size_t)n + 11 * (size_t)m + 16 +
                     (((size_t)nnzA_compact + 1) & ~(size_t)1);
  return (off + 1) & ~(size_t)1;
}
inline size_t qp_sparse_ws_doubles(int n, int m, int nnzL, int funits, int bunits, int nnzA_compact = 0)
{
  // nnzA_compact: kept entries of A of a pruned plan (the kernel compacts the item's values into its workspace)
  // Layout: factor of the ADMM matrix, iterate / scaling / constants, header (16 doubles), compacted A, and a SECOND
  // factor block for the polish system -- used by calls with reuse_factor set (sfb.h), which keep the ADMM factor
  // for the next call.
  const size_t tot = qp_sparse_polish_offset(n, m, nnzL, funits, bunits, nnzA_compact) +
                     qp_sparse_factor_doubles(n, m, nnzL, funits, bunits) + m + 4;
  return (tot + 1) & ~(size_t)1;  // (even: the 16-byte loads of the sweep copies keep their alignment from item to item)
}

inline size_t qp_sparse_ws_doubles(const SparsePlanDev &pl)
{
  return qp_sparse_ws_doubles(pl.n, pl.m, pl.nnzL, pl.funits, pl.bunits, pl.Aorig ? pl.nnzA : 0);
}

// One launch over `batch` items (launch position -> item through `order`, nullable).
//   aux (device, qp_sparse_aux_bytes(batch) bytes, need not be initialised; nullable for plain plans = never
//   time-slice): queue of a time-sliced launch + flags of the fallback pool.
//   fallback / fallback_ws: pruned plans only -- the whole-pattern plan and a pool of qp_sparse_fallback_slots()
//   workspace slots of ITS per-item size: an item whose masked entries are not all zero is solved there, inside
//   the same launch.
//   trace (device, batch x trace_cap x 5 doubles, nullable): the reference's verbose table as data -- per item one row
//   (ITER, OBJ, PRI_RES, DUA_RES, TIME us) per stopping check, rows beyond trace_cap are dropped; the caller presets
//   ITER = -1.  Forces one block per item (no time slicing).
//   phase_us (device, batch x 6 doubles, nullable): per item the microseconds of scaling + pre-check | matrix filling |
//   factorisation | iteration | polish | un-scale and report (the reference's summary, qp_solver.hpp:559-563, as data); also
//   through the TRACE instance of the kernel.
hipError_t qp_sparse_launch(const SparsePlanDev &pl, const DenseKernelParams &kp, int64_t batch, const double *Px,
                            const double *q, const double *Ax, const double *l, const double *u, const double *wx,
                            const double *wy, double *x, double *y, double *obj, uint32_t *iter, int32_t *code,
                            double *workspace, hipStream_t stream, const int32_t *order = nullptr, int32_t *aux = nullptr,
                            const SparsePlanDev *fallback = nullptr, double *fallback_ws = nullptr,
                            double *trace = nullptr, int trace_cap = 0, double *phase_us = nullptr);
size_t qp_sparse_aux_bytes(int64_t batch);
int qp_sparse_fallback_slots();

}  // namespace sfb
