// Device-side numeric fill of the MPC QP: the arithmetic of ocp_to_qp_update_dyn (ocp_to_qp.hpp:240-275),
// ocp_to_qp_update_cr (:279-323) and ocp_to_qp_update_ce (:326-373) for a batch of agents, from per-node
// linearisation records (include/sfb.h).  Pure streaming work: every thread produces ONE entry of A (or one
// row of l, u) -- consecutive threads write consecutive addresses; the records are read through the cache
// (each Jacobian entry is read once, the small vectors a few times).  HBM-bound: algorithmic bytes per agent =
// record + A values + l + u.
#include <hip/hip_runtime.h>

#include "../../include/sfb.h"
#include "mpc_kernel.h"

namespace sfb {

// entry (d, c) of block number `block` of a PACKED Jacobian array (`kept` entries per block, see MpcAsmParams)
__device__ __forceinline__ double jac_entry(const double *__restrict__ blk, const int kept, const uint32_t *km, const uint16_t *kp,
                                            const int block, const int d, const int c)
{
  const uint32_t mk = km[d];
  if (((mk >> c) & 1u) == 0u) return 0.0;
  return blk[(size_t)block * kept + kp[d] + __popc(mk & ((1u << c) - 1u))];
}

// The place of entry `idx` of A in ocp_to_qp_update_dyn / _cr / _ce as a table row (host; mirrors the index arithmetic of the
// kernel's direct form below, which stays as the form without a table and as the definition the table is tested against).
void mpc_build_table(const MpcAsmParams &p, const bool shared, MpcAsmDesc *out)
{
  const int nx = p.nx, nu = p.nu, ncr = p.ncr, kmesh = p.kmesh;
  const bool pk = p.packed != 0 && !shared;
  // offset of entry (d, c) of block `block` of a Jacobian array at `base`, -1 when the packing leaves it out
  auto jac_off = [&](int base, int kept, const uint32_t *km, const uint16_t *kp, int block, int d, int c, int rows, int cols) -> int {
    if (!pk) return base + (block * rows + d) * cols + c;
    const uint32_t mk = km[d];
    if (((mk >> c) & 1u) == 0u) return -1;
    return base + block * kept + kp[d] + __builtin_popcount(mk & ((1u << c) - 1u));
  };
  for (int idx = 0; idx < p.nnzA; ++idx) {
    MpcAsmDesc e{-1, 0, 0.0};
    if (idx < p.nnz_dyn) {
      const int row = idx / p.rowlen_dyn, pos = idx - row * p.rowlen_dyn;
      const int node = row / nx, d = row - node * nx;
      const int s = node / kmesh, i = node - s * kmesh;
      const double alpha = p.alpha[s];
      if (pos >= kmesh + nx) {
        e.src  = jac_off(p.o_dfdu, p.n_fu, p.km_fu, p.kp_fu, node, d, pos - (kmesh + nx), nx, nu);
        e.meta = 0;
      } else if (pos >= i && pos < i + nx) {
        const int c = pos - i;
        e.src  = jac_off(p.o_dfdx, p.n_fx, p.km_fx, p.kp_fx, node, d, c, nx, nx);
        e.meta = 1;
        if (p.has_ad) {
          const int code = p.adsrc[d * nx + c];
          if (code != 0) {
            const int k = (code > 0 ? code : -code) - 1, off = node * nx + k + 1;
            e.meta |= (code > 0 ? off : -off) * 16;
          }
        }
        if (c == d) {
          e.meta |= 4;
          e.coef = alpha * p.D[i * kmesh + i];
        }
      } else {
        const int j = (pos < i) ? pos : pos - nx + 1;
        e.meta = 2;
        e.coef = alpha * p.D[j * kmesh + i];
      }
    } else if (idx < p.nnz_dyn + p.nnz_cr) {
      const int q = idx - p.nnz_dyn, row = q / (nx + nu), pos = q - row * (nx + nu);
      const int cnode = row / ncr, cd = row - cnode * ncr;
      e.src  = (pos < nx) ? jac_off(p.o_dcdx, p.n_cx, p.km_cx, p.kp_cx, cnode, cd, pos, ncr, nx)
                          : jac_off(p.o_dcdu, p.n_cu, p.km_cu, p.kp_cu, cnode, cd, pos - nx, ncr, nu);
      e.meta = 3;
    } else {
      const int q = idx - p.nnz_dyn - p.nnz_cr;
      // (the terminal Jacobian always travels in the agent's record; packed with it when the record is packed)
      e.src  = (p.packed != 0 && !shared) ? jac_off(p.o_J, p.n_J, p.km_J, p.kp_J, 0, q / nx, q % nx, nx, nx) : p.o_J + q;
      e.meta = 3 | 8;
    }
    out[idx] = e;
  }
}

// bounds of row r of the QP: l = u of a dynamics / initial-state row, crl - c .. cru - c of a running-constraint row
__device__ __forceinline__ void mpc_bounds_entry(const MpcAsmParams &p, const double *const rec, const int r, double &lo, double &hi)
{
  const int nx = p.nx, ncr = p.ncr;
  if (r < p.N * nx) {  // :272-273
    lo = -p.tf * (rec[p.o_f + r] - rec[p.o_dx + r]);
    hi = lo;
  } else if (r < p.N * (nx + ncr)) {  // :321-322
    const int q = r - p.N * nx, d = q % ncr;
    const double cv = rec[p.o_c + q];
    lo = p.crl[d] - cv;
    hi = p.cru[d] - cv;
  } else {  // :371-372
    lo = 0.0 - rec[p.o_e + (r - p.N * (nx + ncr))];
    hi = lo;
  }
}

#ifndef SFB_MPC_ASM_AGENTS
#define SFB_MPC_ASM_AGENTS 4
#endif
constexpr int kMpcAsmAgents = SFB_MPC_ASM_AGENTS;  // agents per thread of the table form

__global__ void __launch_bounds__(256) mpc_assemble_kernel(const MpcAsmParams p, const int64_t agents, const double *__restrict__ records,
                                                           const double *__restrict__ shared_jac,
                                                           double *__restrict__ gAx, double *__restrict__ gl,
                                                           double *__restrict__ gu, const MpcAsmDesc *__restrict__ table)
{
  const int idx    = blockIdx.x * 256 + threadIdx.x;
  const int nx = p.nx, nu = p.nu, ncr = p.ncr, kmesh = p.kmesh;
  const double tf = p.tf;
  const bool pk   = p.packed != 0 && shared_jac == nullptr;
  if (table != nullptr) {
    // TABLE FORM (swarms): one 16-byte row says where the entry's Jacobian value is and which terms it takes -- the same
    // operations in the same order as the direct form below, without its divisions and mask arithmetic (the direct form spent
    // 0.88 ms per 8 192 agents of the headline model on 840 MB of output: integer-bound, not HBM-bound).  A thread takes its
    // entry for kMpcAsmAgents agents in a row: the row of the table is read once and the agents' operands are requested
    // together (one 8-byte result per thread behind two dependent loads left the launch at 1.6 TB/s of writes, latency-bound).
    const int64_t b0 = (int64_t)blockIdx.y * kMpcAsmAgents;
    if (idx < p.nnzA) {
      const MpcAsmDesc e = table[idx];
      const int kind     = e.meta & 3;
      const int code     = (kind == 1) ? (e.meta >> 4) : 0;
      const int k        = (code > 0 ? code : -code) - 1;
      double j[kMpcAsmAgents], sf[kMpcAsmAgents], sd[kMpcAsmAgents];
#pragma unroll
      for (int a = 0; a < kMpcAsmAgents; ++a) {
        const bool on      = b0 + a < agents;
        const double *reca = records + (on ? b0 + a : b0) * p.rec_doubles;
        const double *jaca = shared_jac ? shared_jac : reca;
        j[a]  = e.src >= 0 ? ((e.meta & 8) ? reca : jaca)[e.src] : 0.0;
        sf[a] = code != 0 ? reca[p.o_f + k] : 0.0;
        sd[a] = code != 0 ? reca[p.o_dx + k] : 0.0;
      }
#pragma unroll
      for (int a = 0; a < kMpcAsmAgents; ++a) {
        double v;
        if (kind == 3) {
          v = j[a];
        } else if (kind == 0) {
          v = 0.0 + tf * j[a];
        } else if (kind == 1) {
          v = 0.0;
          v += tf * j[a];
          if (code != 0) {  // (no ad term, or one whose entry is 0.0: adding -tf/2 * 0.0 leaves v as it is)
            const double sk = sf[a] + sd[a];
            v += (-tf / 2) * ((code > 0) ? sk : -sk);
          }
          if (e.meta & 4) v -= e.coef;
        } else {
          v = 0.0;
          v -= e.coef;
        }
        if (b0 + a < agents) gAx[(b0 + a) * p.nnzA + idx] = v;
      }
    } else if (idx < p.nnzA + p.m) {
      const int r = idx - p.nnzA;
#pragma unroll
      for (int a = 0; a < kMpcAsmAgents; ++a) {
        if (b0 + a >= agents) break;
        double lo, hi;
        mpc_bounds_entry(p, records + (b0 + a) * p.rec_doubles, r, lo, hi);
        gl[(b0 + a) * p.m + r] = lo;
        gu[(b0 + a) * p.m + r] = hi;
      }
    }
    return;
  }
  const int64_t b  = blockIdx.y;
  const double *rec = records + b * p.rec_doubles;
  const double *jac = shared_jac ? shared_jac : rec;
  if (idx < p.nnzA) {
    double v;
    if (idx < p.nnz_dyn) {  // ocp_to_qp_update_dyn :240-275
      const int row = idx / p.rowlen_dyn, pos = idx - row * p.rowlen_dyn;
      const int node = row / nx, d = row - node * nx;
      const int s = node / kmesh, i = node - s * kmesh;
      const double alpha = p.alpha[s];
      if (pos >= kmesh + nx) {  // u block :259
        const int c = pos - (kmesh + nx);
        const double ju = pk ? jac_entry(jac + p.o_dfdu, p.n_fu, p.km_fu, p.kp_fu, node, d, c)
                             : jac[p.o_dfdu + (node * nx + d) * nu + c];
        v               = 0.0 + tf * ju;
      } else if (pos >= i && pos < i + nx) {  // x block of the node itself
        const int c = pos - i;
        v           = 0.0;
        v += tf * (pk ? jac_entry(jac + p.o_dfdx, p.n_fx, p.km_fx, p.kp_fx, node, d, c)
                      : jac[p.o_dfdx + (node * nx + d) * nx + c]);  // :258
        if (p.has_ad) {                                      // :262-264   -tf/2 ad(f + dxdes)
          const int code = p.adsrc[d * nx + c];
          double a       = 0.0;
          if (code != 0) {
            const int k     = (code > 0 ? code : -code) - 1;
            const double sk = rec[p.o_f + node * nx + k] + rec[p.o_dx + node * nx + k];
            a               = (code > 0) ? sk : -sk;
          }
          v += (-tf / 2) * a;
        }
        if (c == d) v -= alpha * p.D[i * kmesh + i];  // :266-270
      } else {  // the other nodes of the interval: -alpha D(j, i) on the diagonal
        const int j = (pos < i) ? pos : pos - nx + 1;
        v           = 0.0;
        v -= alpha * p.D[j * kmesh + i];
      }
    } else if (idx < p.nnz_dyn + p.nnz_cr) {  // ocp_to_qp_update_cr :279-323
      const int q = idx - p.nnz_dyn, row = q / (nx + nu), pos = q - row * (nx + nu);
      if (pk) {
        const int cnode = row / ncr, cd = row - cnode * ncr;
        v = (pos < nx) ? jac_entry(jac + p.o_dcdx, p.n_cx, p.km_cx, p.kp_cx, cnode, cd, pos)
                       : jac_entry(jac + p.o_dcdu, p.n_cu, p.km_cu, p.kp_cu, cnode, cd, pos - nx);
      } else {
        v = (pos < nx) ? jac[p.o_dcdx + row * nx + pos] : jac[p.o_dcdu + row * nu + (pos - nx)];
      }
    } else {  // ocp_to_qp_update_ce :326-373
      const int e = idx - p.nnz_dyn - p.nnz_cr;
      v           = pk ? jac_entry(rec + p.o_J, p.n_J, p.km_J, p.kp_J, 0, e / nx, e % nx) : rec[p.o_J + e];
    }
    gAx[b * p.nnzA + idx] = v;
  } else if (idx < p.nnzA + p.m) {
    const int r = idx - p.nnzA;
    double lo, hi;
    mpc_bounds_entry(p, rec, r, lo, hi);
    gl[b * p.m + r] = lo;
    gu[b * p.m + r] = hi;
  }
}

__global__ void __launch_bounds__(256) mpc_replicate_kernel(const double *__restrict__ src, const int64_t len,
                                                            double *__restrict__ out)
{
  const int64_t b = blockIdx.y;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < len; e += (int64_t)gridDim.x * 256) out[b * len + e] = src[e];
}

__global__ void __launch_bounds__(64) mpc_store_kernel(const int n, const int m, const int uoff, const int nu,
                                                       const int store, const double *__restrict__ x,
                                                       const double *__restrict__ y, const int32_t *__restrict__ code,
                                                       double *__restrict__ wx, double *__restrict__ wy,
                                                       double *__restrict__ du0)
{
  const int64_t b = blockIdx.x;
  const int lane  = threadIdx.x;
  const double *xb = x + b * n, *yb = y + b * m;
  for (int c = lane; c < nu; c += 64) du0[b * nu + c] = xb[uoff + c];
  const int32_t cd = code[b];
  if (store && (cd == SFB_QP_OPTIMAL || cd == SFB_QP_MAX_TIME || cd == SFB_QP_MAX_ITERATIONS)) {
    for (int j = lane; j < n; j += 64) wx[b * n + j] = xb[j];
    for (int i = lane; i < m; i += 64) wy[b * m + i] = yb[i];
  }
}

hipError_t mpc_assemble_launch(const MpcAsmParams &p, int64_t batch, const double *records, const double *shared_jac,
                               double *Ax, double *l, double *u, hipStream_t stream, const MpcAsmDesc *table)
{
  const int blocks = (p.nnzA + p.m + 255) / 256;
  const int64_t per = table != nullptr ? kMpcAsmAgents : 1;  // agents per row of the grid
  for (int64_t b0 = 0; b0 < batch; b0 += 65535 * per) {    // gridDim.y limit
    const int64_t nb = (batch - b0 < 65535 * per) ? batch - b0 : 65535 * per;
    hipLaunchKernelGGL(mpc_assemble_kernel, dim3(blocks, (unsigned)((nb + per - 1) / per)), dim3(256), 0, stream, p, nb,
                       records + b0 * p.rec_doubles, shared_jac, Ax + b0 * p.nnzA, l + b0 * p.m, u + b0 * p.m, table);
  }
  return hipGetLastError();
}

hipError_t mpc_replicate_launch(const double *src, int64_t len, int64_t batch, double *out, hipStream_t stream)
{
  if (len == 0) return hipSuccess;
  int64_t blocks = (len + 255) / 256;
  if (blocks > 64) blocks = 64;
  for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
    const int64_t nb = (batch - b0 < 65535) ? batch - b0 : 65535;
    hipLaunchKernelGGL(mpc_replicate_kernel, dim3((unsigned)blocks, (unsigned)nb), dim3(256), 0, stream, src, len,
                       out + b0 * len);
  }
  return hipGetLastError();
}

hipError_t mpc_store_launch(int64_t batch, int n, int m, int uoff, int nu, int store, const double *x, const double *y,
                            const int32_t *code, double *wx, double *wy, double *du0, hipStream_t stream)
{
  hipLaunchKernelGGL(mpc_store_kernel, dim3((unsigned)batch), dim3(64), 0, stream, n, m, uoff, nu, store, x, y, code, wx,
                     wy, du0);
  return hipGetLastError();
}

}  // namespace sfb
