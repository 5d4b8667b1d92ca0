// The vehicle of examples/mpc_asif_vehicle.cpp (:42-55 dynamics, :95-129 safety filter) as functors usable on the host and
// in HIP device code: state X6 = SE2 x R^3 (pose, body velocities), input U2 = R^2.  With analytic right-Jacobians (the
// reference differentiates its lambdas with autodiff; without the members the fronts fall back to forward differences).
#pragma once
#include <cmath>
#include <cstddef>

#include <smooth_feedback_amd/asif.hpp>
#include <smooth_feedback_amd/ekf.hpp>
#include <smooth_feedback_amd/lie.hpp>
#include <smooth_feedback_amd/mpc.hpp>

namespace sfbx {
using namespace smooth_feedback_amd;

using X6 = Bundle<SE2, Rn<3>>;
using U2 = Rn<2>;

struct VehicleDyn6 {
  SFB_LIE_HD Vec<6> operator()(const X6 & x, const U2 & u) const
  {
    const auto & v = x.part<1>().v;
    return {v[0], v[1], v[2], -0.2 * v[0] + u.v[0], 0.0, -0.4 * v[2] + u.v[1]};
  }
  SFB_LIE_HD void jacobian(const X6 &, const U2 &, Mat<6, 6> & dx, Mat<6, 2> & du) const
  {
    dx = Mat<6, 6>::Zero(); du = Mat<6, 2>::Zero();
    dx(0, 3) = 1; dx(1, 4) = 1; dx(2, 5) = 1; dx(3, 3) = -0.2; dx(5, 5) = -0.4;
    du(3, 0) = 1; du(5, 1) = 1;
  }
};

// two such vehicles driven by one input pair (synthetic: BASELINE.json's "nx=12, nu=2" problem size)
using X12 = Bundle<SE2, Rn<3>, SE2, Rn<3>>;
struct VehicleDyn12 {
  SFB_LIE_HD Vec<12> operator()(const X12 & x, const U2 & u) const
  {
    const auto & v = x.part<1>().v;
    const auto & w = x.part<3>().v;
    return {v[0], v[1], v[2], -0.2 * v[0] + u.v[0], 0.0, -0.4 * v[2] + u.v[1],
            w[0], w[1], w[2], -0.3 * w[0] + u.v[0], 0.0, -0.5 * w[2] + u.v[1]};
  }
  SFB_LIE_HD void jacobian(const X12 &, const U2 &, Mat<12, 12> & dx, Mat<12, 2> & du) const
  {
    dx = Mat<12, 12>::Zero(); du = Mat<12, 2>::Zero();
    dx(0, 3) = 1; dx(1, 4) = 1; dx(2, 5) = 1; dx(3, 3) = -0.2; dx(5, 5) = -0.4;
    dx(6, 9) = 1; dx(7, 10) = 1; dx(8, 11) = 1; dx(9, 9) = -0.3; dx(11, 11) = -0.5;
    du(3, 0) = 1; du(5, 1) = 1; du(9, 0) = 1; du(11, 1) = 1;
  }
};
// running constraint: the input itself (bounded to [-0.5, 0.5]^2 by the MPC's crl / cru)
template<class X>
struct InputBox {
  SFB_LIE_HD Vec<2> operator()(const X &, const U2 & u) const { return {u.v[0], u.v[1]}; }
  SFB_LIE_HD void jacobian(const X &, const U2 &, Mat<2, X::Dof> & dx, Mat<2, 2> & du) const
  {
    dx = Mat<2, X::Dof>::Zero();
    du = Mat<2, 2>::Identity();
  }
};

// the MPC models: desired trajectories of examples/mpc_asif_vehicle.cpp:73-79 + dynamics + running constraint
// (device-callable: mpc_device.hpp linearises them on the GPU; the host MPC objects are built from the same members)
struct VehicleModel6 {
  VehicleDyn6 f;
  InputBox<X6> cr;
  SFB_LIE_HD X6 xdes(double t) const
  {
    X6 x;
    x.part<0>() = rplus(SE2::FromAngle(1.5707963267948966, 2.5, 0.0), SE2::Tangent{t * 1.0, 0.0, t * 0.4});
    x.part<1>().v = {1.0, 0.0, 0.4};
    return x;
  }
  SFB_LIE_HD Vec<6> dxdes(double) const { return {1.0, 0.0, 0.4, 0.0, 0.0, 0.0}; }
  SFB_LIE_HD U2 udes(double) const { return U2{}; }
};
struct VehicleModel12 {
  VehicleDyn12 f;
  InputBox<X12> cr;
  SFB_LIE_HD X12 xdes(double t) const
  {
    X12 x;
    x.part<0>() = rplus(SE2::FromAngle(1.5707963267948966, 2.5, 0.0), SE2::Tangent{t * 1.0, 0.0, t * 0.4});
    x.part<1>().v = {1.0, 0.0, 0.4};
    x.part<2>() = rplus(SE2::FromAngle(1.5707963267948966, 2.5, -1.0), SE2::Tangent{t * 0.8, 0.0, t * 0.3});
    x.part<3>().v = {0.8, 0.0, 0.3};
    return x;
  }
  SFB_LIE_HD Vec<12> dxdes(double) const { return {1.0, 0.0, 0.4, 0, 0, 0, 0.8, 0.0, 0.3, 0, 0, 0}; }
  SFB_LIE_HD U2 udes(double) const { return U2{}; }
};

// barrier: stay 0.7 away from the obstacle at (0, -2.3)
struct VehicleH {
  SFB_LIE_HD Vec<1> operator()(double, const X6 & x) const
  {
    const double dx = x.part<0>().x - 0.0, dy = x.part<0>().y - (-2.3);
    const double nrm = std::sqrt(dx * dx + dy * dy);
    return {(dx * dx + dy * dy) / nrm - 0.7};
  }
  // h = |p - c| - 0.7 and p (+) a = p + R (a_0, a_1) + O(a^2): dh/da = (p - c)' R / |p - c| on the SE2 translation part
  SFB_LIE_HD void jacobian(double, const X6 & x, Mat<1, 6> & J) const
  {
    const auto & g  = x.part<0>();
    const double dx = g.x - 0.0, dy = g.y - (-2.3), nrm = std::sqrt(dx * dx + dy * dy);
    J       = Mat<1, 6>::Zero();
    J(0, 0) = (dx * g.c + dy * g.s) / nrm;
    J(0, 1) = (-dx * g.s + dy * g.c) / nrm;
  }
  SFB_LIE_HD Vec<1> operator()(std::size_t, double t, const X6 & x) const { return (*this)(t, x); }  // swarm callbacks
  SFB_LIE_HD void jacobian(std::size_t, double t, const X6 & x, Mat<1, 6> & J) const { jacobian(t, x, J); }
};

// backup controller: brake and turn
struct VehicleBU {
  SFB_LIE_HD U2 operator()(double, const X6 & x) const
  {
    U2 u;
    u.v = {0.2 * x.part<1>().v[0], -0.5};
    return u;
  }
  SFB_LIE_HD void jacobian(double, const X6 &, Mat<2, 6> & J) const
  {
    J       = Mat<2, 6>::Zero();
    J(0, 3) = 0.2;
  }
  SFB_LIE_HD U2 operator()(std::size_t, double t, const X6 & x) const { return (*this)(t, x); }
  SFB_LIE_HD void jacobian(std::size_t, double t, const X6 & x, Mat<2, 6> & J) const { jacobian(t, x, J); }
};

// an EKF on the vehicle: known, time-varying input; position and forward speed are measured
struct VehicleEkfDyn {
  SFB_LIE_HD Vec<6> operator()(double t, const X6 & x) const
  {
    U2 u;
    u.v = {0.3 * std::cos(2.0 * t), 0.2 * std::sin(3.0 * t)};
    return VehicleDyn6{}(x, u);
  }
};
struct VehicleEkfMeas {
  SFB_LIE_HD Vec<3> operator()(const X6 & x) const { return {x.part<0>().x, x.part<0>().y, x.part<1>().v[0]}; }
};
inline Mat<6, 6> vehicle_ekf_Q()
{
  Mat<6, 6> Q{};
  for (int i = 0; i < 6; ++i) Q(i, i) = 0.02 + 0.01 * i;
  Q(0, 1) = Q(1, 0) = 0.004; Q(3, 5) = Q(5, 3) = -0.003;
  return Q;
}
inline Mat<3, 3> vehicle_ekf_R()
{
  Mat<3, 3> R{};
  R(0, 0) = 0.1; R(1, 1) = 0.12; R(2, 2) = 0.05; R(0, 1) = R(1, 0) = 0.01;
  return R;
}
inline X6 vehicle_state(const double * s)  // (x, y, cos, sin, v0, v1, v2)
{
  X6 g;
  g.part<0>()   = SE2{s[0], s[1], s[2], s[3]};
  g.part<1>().v = {s[4], s[5], s[6]};
  return g;
}
inline void vehicle_state_out(const X6 & g, double * s)
{
  const auto & p = g.part<0>();
  const auto & v = g.part<1>().v;
  s[0] = p.x; s[1] = p.y; s[2] = p.c; s[3] = p.s; s[4] = v[0]; s[5] = v[1]; s[6] = v[2];
}

using MPC6  = MPC<double, X6, U2, VehicleDyn6, InputBox<X6>>;
using MPC12 = MPC<double, X12, U2, VehicleDyn12, InputBox<X12>>;
template<class MPCT, class Model>
MPCT make_vehicle_mpc(int K, double tf)
{
  MPCParams p;
  p.K = (size_t)K; p.tf = tf;
  const Model mdl{};
  MPCT m(mdl.f, mdl.cr, {-0.5, -0.5}, {0.5, 0.5}, p);
  m.set_xdes([mdl](double t) { return mdl.xdes(t); }, [mdl](double t) { return mdl.dxdes(t); });
  m.set_udes([mdl](double t) { return mdl.udes(t); });
  return m;
}

inline ASIFilterParams<U2> vehicle_asif_params(int K)
{
  ASIFilterParams<U2> p;
  p.T        = 2.5;
  p.nh       = 1;
  p.u_weight = {20.0, 1.0};
  p.ulim.rows = 2;
  p.ulim.A    = {1, 0, 0, 1};
  p.ulim.l    = {-0.2, -0.5};
  p.ulim.u    = {0.5, 0.5};
  p.asif.K          = (size_t)K;
  p.asif.alpha      = 5;
  p.asif.dt         = 0.01;
  p.asif.relax_cost = 100;
  p.qp.polish       = false;
  return p;
}

}  // namespace sfbx
