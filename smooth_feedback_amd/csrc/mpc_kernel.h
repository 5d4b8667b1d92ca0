// Host<->kernel interface of the device-side MPC assembly.  Internal.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace sfb {

constexpr int kMpcMaxIvals = 128, kMpcMaxKmesh = 8, kMpcMaxNx = 24, kMpcMaxNcr = 16;

// Everything the kernel needs besides the records travels as kernel argument (< 4 KB).
struct MpcAsmParams {
  int nx, nu, ncr, kmesh, nivals, N;
  int rowlen_dyn, nnz_dyn, nnz_cr, nnzA, m;
  int has_ad;
  double tf;
  // offsets (doubles) inside the per-agent record and inside the record holding the Jacobians (which is the
  // per-agent record unless the Jacobians are shared)
  int o_f, o_dx, o_c, o_e, o_J, o_dfdx, o_dfdu, o_dcdx, o_dcdu;
  int64_t rec_doubles;
  double alpha[kMpcMaxIvals];
  double D[(kMpcMaxKmesh + 1) * kMpcMaxKmesh];
  double crl[kMpcMaxNcr], cru[kMpcMaxNcr];
  // ad(s) of the state group as a table over (d, c): 0 -> 0.0, +(k+1) -> s[k], -(k+1) -> -s[k]
  int8_t adsrc[kMpcMaxNx * kMpcMaxNx];
  // Packed per-agent Jacobians (sfb_mpc_layout::jac_keep): the record holds only the entries whose flag is set, row by
  // row; an entry (d, c) of a block is at [kept entries of the rows before d] + popcount(row mask below bit c), the
  // others are 0.0.  km_*: one 32-bit column mask per row, kp_*: kept entries in the rows before, n_*: kept per block.
  int packed;
  int n_fx, n_fu, n_cx, n_cu, n_J;
  uint32_t km_fx[kMpcMaxNx], km_fu[kMpcMaxNx], km_cx[kMpcMaxNcr], km_cu[kMpcMaxNcr], km_J[kMpcMaxNx];
  uint16_t kp_fx[kMpcMaxNx], kp_fu[kMpcMaxNx], kp_cx[kMpcMaxNcr], kp_cu[kMpcMaxNcr], kp_J[kMpcMaxNx];
};

// One stored entry of A as the assembly kernel's table form sees it (built once per layout / packing by mpc_build_table, the
// same for every agent): where its Jacobian entry sits, which term of ocp_to_qp_update_* it takes part in, its constant.
//   meta bits 0..1: 0 = u block of a dynamics row (0.0 + tf j), 1 = x block of the node itself (0.0 + tf j, ad term, - coef),
//                   2 = another node of the interval (0.0 - coef), 3 = the entry itself (cr and ce rows)
//        bit 2: subtract coef (kind 1: the diagonal), bit 3: src is an offset into the agent's record even when the Jacobians are
//        shared (ce rows), bits 4..31 (signed): the ad term, +-(offset of s_k inside f / dxdes + 1), 0 = none
//   src: offset (doubles) of j inside the record holding the Jacobians, -1 = a structural 0.0 (entry left out by the packing)
struct MpcAsmDesc {
  int32_t src, meta;
  double coef;
};
// the table of a parameter set (nnzA entries); `shared`: the Jacobians come from the shared record
void mpc_build_table(const MpcAsmParams &p, bool shared, MpcAsmDesc *out);

// table (device, nullable): the entries of A through the table instead of recomputing their place from the index
hipError_t mpc_assemble_launch(const MpcAsmParams &p, int64_t batch, const double *records, const double *shared_jac,
                               double *Ax, double *l, double *u, hipStream_t stream, const MpcAsmDesc *table = nullptr);
// out[b][:] = src[:] for every b
hipError_t mpc_replicate_launch(const double *src, int64_t len, int64_t batch, double *out, hipStream_t stream);
// after a solve: agents whose code is Optimal / MaxIterations / MaxTime store (x, y) as their warm start
// (mpc.hpp:510-516); du0[b] = x[b][uoff : uoff + nu]
hipError_t mpc_store_launch(int64_t batch, int n, int m, int uoff, int nu, int store, const double *x, const double *y,
                            const int32_t *code, double *wx, double *wy, double *du0, hipStream_t stream);

}  // namespace sfb
