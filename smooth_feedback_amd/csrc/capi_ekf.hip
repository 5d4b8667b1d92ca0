// C-ABI for the batched EKF path (include/sfb.h).
#include <hip/hip_runtime.h>

#include <vector>

#include "../../include/sfb.h"
#include "capi_common.h"
#include "ekf_kernel.h"

namespace {

sfb_status ekf_common(int64_t batch, int dof, int ny, const double *A, const double *Q, int q_shared, const double *dt,
                      int dt_shared, const double *H, const double *R, int r_shared, const double *r, double *P,
                      double *delta, int32_t *info, bool predict, bool update, void *stream)
{
  if (batch < 0) return sfb::fail(SFB_ERR_INVALID_ARG, "batch < 0");
  if (!predict && !update) return sfb::fail(SFB_ERR_INVALID_ARG, "nothing to do");
  if (batch > 0 && !P) return sfb::fail(SFB_ERR_INVALID_ARG, "P is NULL");
  if (predict && batch > 0 && (!A || !Q || !dt)) return sfb::fail(SFB_ERR_INVALID_ARG, "predict needs A, Q, dt");
  if (update && batch > 0 && (!H || !R || !r || !delta)) return sfb::fail(SFB_ERR_INVALID_ARG, "update needs H, R, r, delta");
  if (!sfb::ekf_supported(dof, ny, update))
    return sfb::fail(SFB_ERR_UNSUPPORTED, "EKF kernels support dof and ny up to 16");
  sfb_status st = sfb::require_device();
  if (st != SFB_OK) return st;
  if (batch == 0) return SFB_OK;
  sfb::EkfArgs a{};
  a.batch = batch; a.A = A; a.Q = Q; a.dt = dt; a.q_shared = q_shared; a.dt_shared = dt_shared;
  a.H = H; a.R = R; a.r = r; a.r_shared = r_shared; a.delta = delta; a.info = info; a.P = P;
  hipError_t e = sfb::ekf_launch(a, dof, ny, predict, update, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return sfb::hip_fail(e, "ekf_kernel launch");
  return SFB_OK;
}

}  // namespace

extern "C" {

sfb_status sfb_ekf_predict_batch(int64_t batch, int dof, const double *A, const double *Q, int q_shared,
                                 const double *dt, int dt_shared, double *P, void *stream)
{
  return ekf_common(batch, dof, 1, A, Q, q_shared, dt, dt_shared, nullptr, nullptr, 0, nullptr, P, nullptr, nullptr,
                    true, false, stream);
}

sfb_status sfb_ekf_predict_stepper_batch(int stepper, int64_t batch, int dof, const double *A, const double *Q,
                                         int q_shared, const double *dt, int dt_shared, double *P, void *stream)
{
  if (stepper == SFB_EKF_EULER) return sfb_ekf_predict_batch(batch, dof, A, Q, q_shared, dt, dt_shared, P, stream);
  if (stepper != SFB_EKF_RK4) return sfb::fail(SFB_ERR_INVALID_ARG, "unknown stepper");
  return sfb_ekf_predict_rk4_batch(batch, dof, A, nullptr, nullptr, Q, q_shared, dt, dt_shared, P, stream);
}

sfb_status sfb_ekf_predict_rk4_batch(int64_t batch, int dof, const double *A, const double *A_mid, const double *A_end,
                                     const double *Q, int q_shared, const double *dt, int dt_shared, double *P,
                                     void *stream)
{
  if (batch < 0) return sfb::fail(SFB_ERR_INVALID_ARG, "batch < 0");
  if (batch > 0 && (!A || !Q || !dt || !P)) return sfb::fail(SFB_ERR_INVALID_ARG, "predict needs A, Q, dt, P");
  if ((A_mid == nullptr) != (A_end == nullptr)) return sfb::fail(SFB_ERR_INVALID_ARG, "A_mid and A_end must both be given or both be NULL");
  if (!sfb::ekf_supported(dof, 1, false)) return sfb::fail(SFB_ERR_UNSUPPORTED, "EKF kernels support dof up to 16");
  sfb_status st = sfb::require_device();
  if (st != SFB_OK) return st;
  if (batch == 0) return SFB_OK;
  sfb::EkfArgs a{};
  a.batch = batch; a.A = A; a.A_mid = A_mid; a.A_end = A_end; a.Q = Q; a.dt = dt; a.q_shared = q_shared;
  a.dt_shared = dt_shared; a.P = P;
  hipError_t e = sfb::ekf_rk4_launch(a, dof, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return sfb::hip_fail(e, "ekf_rk4_kernel launch");
  return SFB_OK;
}

sfb_status sfb_ekf_predict_rk4_batch_host(int64_t batch, int dof, const double *A, const double *A_mid, const double *A_end,
                                          const double *Q, int q_shared, const double *dt, int dt_shared, double *P)
{
  if (batch < 0 || (batch > 0 && (!A || !Q || !dt || !P))) return sfb::fail(SFB_ERR_INVALID_ARG, "bad arguments");
  if ((A_mid == nullptr) != (A_end == nullptr)) return sfb::fail(SFB_ERR_INVALID_ARG, "A_mid and A_end must both be given or both be NULL");
  if (!sfb::ekf_supported(dof, 1, false)) return sfb::fail(SFB_ERR_UNSUPPORTED, "EKF kernels support dof up to 16");
  sfb_status st = sfb::require_device();
  if (st != SFB_OK) return st;
  if (batch == 0) return SFB_OK;
  const size_t B = (size_t)batch, nn = (size_t)dof * dof;
  const size_t nQ = q_shared ? nn : B * nn, nT = dt_shared ? 1 : B, nA = A_mid ? 3 : 1;
  double *dev = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&dev), ((1 + nA) * B * nn + nQ + nT) * sizeof(double));
  if (e != hipSuccess) return sfb::hip_fail(e, "hipMalloc");
  double *dP = dev, *dA = dP + B * nn, *dAm = dA + B * nn, *dAe = dAm + B * nn, *dQ = dA + nA * B * nn, *ddt = dQ + nQ;
  e = hipMemcpy(dP, P, B * nn * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dA, A, B * nn * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess && A_mid) e = hipMemcpy(dAm, A_mid, B * nn * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess && A_end) e = hipMemcpy(dAe, A_end, B * nn * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dQ, Q, nQ * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(ddt, dt, nT * 8, hipMemcpyHostToDevice);
  st = SFB_OK;
  if (e == hipSuccess) {
    st = sfb_ekf_predict_rk4_batch(batch, dof, dA, A_mid ? dAm : nullptr, A_end ? dAe : nullptr, dQ, q_shared, ddt,
                                   dt_shared, dP, nullptr);
    if (st == SFB_OK) {
      e = hipDeviceSynchronize();
      if (e == hipSuccess) e = hipMemcpy(P, dP, B * nn * 8, hipMemcpyDeviceToHost);
    }
  }
  (void)hipFree(dev);
  if (e != hipSuccess) return sfb::hip_fail(e, "sfb_ekf_predict_rk4_batch_host");
  return st;
}

sfb_status sfb_ekf_predict_stepper_batch_host(int stepper, int64_t batch, int dof, const double *A, const double *Q,
                                              int q_shared, const double *dt, int dt_shared, double *P)
{
  if (stepper == SFB_EKF_EULER)
    return sfb_ekf_step_batch_host(batch, dof, 1, A, Q, q_shared, dt, dt_shared, nullptr, nullptr, 0, nullptr, P,
                                   nullptr, nullptr);
  if (stepper != SFB_EKF_RK4) return sfb::fail(SFB_ERR_INVALID_ARG, "unknown stepper");
  return sfb_ekf_predict_rk4_batch_host(batch, dof, A, nullptr, nullptr, Q, q_shared, dt, dt_shared, P);
}

sfb_status sfb_ekf_update_batch(int64_t batch, int dof, int ny, const double *H, const double *R, int r_shared,
                                const double *r, double *P, double *delta, int32_t *info, void *stream)
{
  return ekf_common(batch, dof, ny, nullptr, nullptr, 0, nullptr, 0, H, R, r_shared, r, P, delta, info, false, true,
                    stream);
}

sfb_status sfb_ekf_predict_update_batch(int64_t batch, int dof, int ny, const double *A, const double *Q, int q_shared,
                                        const double *dt, int dt_shared, const double *H, const double *R,
                                        int r_shared, const double *r, double *P, double *delta, int32_t *info,
                                        void *stream)
{
  return ekf_common(batch, dof, ny, A, Q, q_shared, dt, dt_shared, H, R, r_shared, r, P, delta, info, true, true,
                    stream);
}

sfb_status sfb_ekf_step_batch_host(int64_t batch, int dof, int ny, const double *A, const double *Q, int q_shared,
                                   const double *dt, int dt_shared, const double *H, const double *R, int r_shared,
                                   const double *r, double *P, double *delta, int32_t *info)
{
  const bool predict = A != nullptr, update = H != nullptr;
  if (batch < 0 || (!predict && !update) || (batch > 0 && !P)) return sfb::fail(SFB_ERR_INVALID_ARG, "bad arguments");
  if (!sfb::ekf_supported(dof, ny, update))
    return sfb::fail(SFB_ERR_UNSUPPORTED, "EKF kernels support dof and ny up to 16");
  sfb_status st = sfb::require_device();
  if (st != SFB_OK) return st;
  if (batch == 0) return SFB_OK;
  const size_t B = (size_t)batch, nn = (size_t)dof * dof, mn = (size_t)ny * dof, mm = (size_t)ny * ny;
  struct Buf { const void *h; size_t bytes; void **d; };
  double *dA = nullptr, *dQ = nullptr, *ddt = nullptr, *dH = nullptr, *dR = nullptr, *dr = nullptr, *dP = nullptr,
         *ddelta = nullptr;
  int32_t *dinfo = nullptr;
  std::vector<void *> owned;
  hipError_t e = hipSuccess;
  auto up = [&](const double *h, size_t cnt, double **d) {
    if (e != hipSuccess || !h) return;
    e = hipMalloc(reinterpret_cast<void **>(d), cnt * 8);
    if (e != hipSuccess) return;
    owned.push_back(*d);
    e = hipMemcpy(*d, h, cnt * 8, hipMemcpyHostToDevice);
  };
  up(P, B * nn, &dP);
  if (predict) { up(A, B * nn, &dA); up(Q, q_shared ? nn : B * nn, &dQ); up(dt, dt_shared ? 1 : B, &ddt); }
  if (update) {
    up(H, B * mn, &dH); up(R, r_shared ? mm : B * mm, &dR); up(r, B * ny, &dr);
    if (e == hipSuccess) { e = hipMalloc(reinterpret_cast<void **>(&ddelta), B * dof * 8); if (e == hipSuccess) owned.push_back(ddelta); }
    if (e == hipSuccess && info) { e = hipMalloc(reinterpret_cast<void **>(&dinfo), B * 4); if (e == hipSuccess) owned.push_back(dinfo); }
  }
  st = SFB_OK;
  if (e == hipSuccess) {
    st = ekf_common(batch, dof, ny, dA, dQ, q_shared, ddt, dt_shared, dH, dR, r_shared, dr, dP, ddelta, dinfo, predict,
                    update, nullptr);
    if (st == SFB_OK) {
      e = hipDeviceSynchronize();
      if (e == hipSuccess) e = hipMemcpy(P, dP, B * nn * 8, hipMemcpyDeviceToHost);
      if (e == hipSuccess && update) e = hipMemcpy(delta, ddelta, B * dof * 8, hipMemcpyDeviceToHost);
      if (e == hipSuccess && update && info) e = hipMemcpy(info, dinfo, B * 4, hipMemcpyDeviceToHost);
    }
  }
  for (void *p : owned) (void)hipFree(p);
  if (e != hipSuccess) return sfb::hip_fail(e, "sfb_ekf_step_batch_host");
  return st;
}

}  // extern "C"
