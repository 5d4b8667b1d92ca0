// Minimal Lie-group layer for the host side of the MPC / EKF paths.
//
// The reference gets this from pettni/smooth (absent here): right-invariant conventions
//   rplus(g, a) = g * exp(a),  rminus(a, b) = log(b^-1 * a),  body velocities d^r x_t = f
// (reference README.md:17-18, mpc.hpp:498,505,518, ekf.hpp:137).  Only what the hot path's callers
// need is restated: R^n, SE(2), SO(3) and Bundle<...> with exp/log, ad and dr_expinv (the inverse right
// Jacobian used by MPCCE::jacobian, mpc.hpp:293-301).  Semantics as summarised in SURVEY.md section
// 8 ("smooth semantics the host side must restate"); parity with the real library is pinned only
// by group identities (tests), not by golden values.
#pragma once
#include <array>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <tuple>
#include <utility>
#include <vector>

// Everything here is usable in HIP device code as well (a model written against these types can be linearised or
// integrated on the GPU, asif_device.hpp): the functions are __host__ __device__ when compiled by hipcc.
#if defined(__HIPCC__)
#define SFB_LIE_HD __host__ __device__
#else
#define SFB_LIE_HD
#endif

namespace smooth_feedback_amd {

// ---- tiny fixed-size column-major matrix ----
template<int R, int C>
struct Mat {
  std::array<double, (R * C > 0 ? R * C : 1)> a{};
  static constexpr int rows = R, cols = C;
  SFB_LIE_HD double &operator()(int r, int c) { return a[(size_t)r + (size_t)c * R]; }
  SFB_LIE_HD double operator()(int r, int c) const { return a[(size_t)r + (size_t)c * R]; }
  SFB_LIE_HD static Mat Zero() { return Mat{}; }
  SFB_LIE_HD static Mat Identity()
  {
    Mat m{};
    for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = 1.0;
    return m;
  }
};
template<int N>
using Vec = std::array<double, (N > 0 ? N : 1)>;

template<int R, int K, int C>
SFB_LIE_HD Mat<R, C> operator*(const Mat<R, K> &A, const Mat<K, C> &B)
{
  Mat<R, C> out{};
  for (int c = 0; c < C; ++c)
    for (int k = 0; k < K; ++k)
      for (int r = 0; r < R; ++r) out(r, c) += A(r, k) * B(k, c);
  return out;
}
template<int R, int C>
SFB_LIE_HD Mat<R, C> operator+(Mat<R, C> A, const Mat<R, C> &B)
{
  for (size_t i = 0; i < A.a.size(); ++i) A.a[i] += B.a[i];
  return A;
}
template<int R, int C>
SFB_LIE_HD Mat<R, C> operator*(double s, Mat<R, C> A)
{
  for (auto &v : A.a) v *= s;
  return A;
}
template<int R, int C>
SFB_LIE_HD Vec<R> operator*(const Mat<R, C> &A, const Vec<C> &x)
{
  Vec<R> y{};
  for (int c = 0; c < C; ++c)
    for (int r = 0; r < R; ++r) y[r] += A(r, c) * x[c];
  return y;
}

// ---- R^n as a (commutative) Lie group ----
template<int N>
struct Rn {
  static constexpr int Dof           = N;
  static constexpr bool IsCommutative = true;
  using Tangent                      = Vec<N>;
  Vec<N> v{};
  SFB_LIE_HD static Rn Identity() { return Rn{}; }
  SFB_LIE_HD friend Rn rplus(const Rn &g, const Tangent &a)
  {
    Rn r = g;
    for (int i = 0; i < N; ++i) r.v[i] += a[i];
    return r;
  }
  SFB_LIE_HD friend Tangent rminus(const Rn &a, const Rn &b)
  {
    Tangent t{};
    for (int i = 0; i < N; ++i) t[i] = a.v[i] - b.v[i];
    return t;
  }
  SFB_LIE_HD static Mat<N, N> ad(const Tangent &) { return Mat<N, N>::Zero(); }
  SFB_LIE_HD static Mat<N, N> dr_expinv(const Tangent &) { return Mat<N, N>::Identity(); }
};

// ---- SE(2): tangent order (v_x, v_y, omega) ----
struct SE2 {
  static constexpr int Dof           = 3;
  static constexpr bool IsCommutative = false;
  using Tangent                      = Vec<3>;
  double x = 0, y = 0, c = 1, s = 0;  // translation, cos/sin of the heading

  SFB_LIE_HD static SE2 Identity() { return SE2{}; }
  SFB_LIE_HD static SE2 FromAngle(double th, double px, double py) { return SE2{px, py, std::cos(th), std::sin(th)}; }
  SFB_LIE_HD double angle() const { return std::atan2(s, c); }

  SFB_LIE_HD static SE2 exp(const Tangent &a)
  {
    const double th = a[2], th2 = th * th;
    double A, B;  // A = sin(th)/th, B = (1-cos(th))/th
    if (th2 < 1e-10) {
      A = 1.0 - th2 / 6.0;
      B = th / 2.0 - th * th2 / 24.0;
    } else {
      A = std::sin(th) / th;
      B = (1.0 - std::cos(th)) / th;
    }
    return SE2{A * a[0] - B * a[1], B * a[0] + A * a[1], std::cos(th), std::sin(th)};
  }
  SFB_LIE_HD Tangent log() const
  {
    const double th = angle(), th2 = th * th;
    double A, B;
    if (th2 < 1e-10) {
      A = 1.0 - th2 / 6.0;
      B = th / 2.0 - th * th2 / 24.0;
    } else {
      A = s / th;
      B = (1.0 - c) / th;
    }
    const double den = A * A + B * B;
    return {(A * x + B * y) / den, (-B * x + A * y) / den, th};
  }
  SFB_LIE_HD SE2 inverse() const { return SE2{-(c * x + s * y), -(-s * x + c * y), c, -s}; }
  SFB_LIE_HD friend SE2 operator*(const SE2 &g, const SE2 &h)
  {
    return SE2{g.x + g.c * h.x - g.s * h.y, g.y + g.s * h.x + g.c * h.y, g.c * h.c - g.s * h.s, g.s * h.c + g.c * h.s};
  }
  SFB_LIE_HD friend SE2 rplus(const SE2 &g, const Tangent &a) { return g * exp(a); }
  SFB_LIE_HD friend Tangent rminus(const SE2 &a, const SE2 &b) { return (b.inverse() * a).log(); }

  SFB_LIE_HD static Mat<3, 3> ad(const Tangent &a)
  {
    Mat<3, 3> m{};
    m(0, 1) = -a[2]; m(0, 2) = a[1];
    m(1, 0) = a[2];  m(1, 2) = -a[0];
    return m;
  }
  // inverse of the right Jacobian of exp:  I + ad/2 + (1/th^2 - (1+cos th)/(2 th sin th)) ad^2
  SFB_LIE_HD static Mat<3, 3> dr_expinv(const Tangent &a)
  {
    const double th = a[2], th2 = th * th;
    const double k  = (th2 < 1e-8) ? (1.0 / 12.0 + th2 / 720.0) : (1.0 / th2 - (1.0 + std::cos(th)) / (2.0 * th * std::sin(th)));
    const Mat<3, 3> A = ad(a);
    return Mat<3, 3>::Identity() + 0.5 * A + k * (A * A);
  }
};

// ---- SO(3): unit quaternion (w, x, y, z), tangent = body angular velocity ----
struct SO3 {
  static constexpr int Dof           = 3;
  static constexpr bool IsCommutative = false;
  using Tangent                      = Vec<3>;
  double w = 1, x = 0, y = 0, z = 0;

  SFB_LIE_HD static SO3 Identity() { return SO3{}; }
  SFB_LIE_HD static SO3 exp(const Tangent &a)
  {
    const double th2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
    double A, B;  // A = sin(th/2)/th, B = cos(th/2)
    if (th2 < 1e-10) {
      A = 0.5 - th2 / 48.0;
      B = 1.0 - th2 / 8.0;
    } else {
      const double th = std::sqrt(th2);
      A = std::sin(0.5 * th) / th;
      B = std::cos(0.5 * th);
    }
    return SO3{B, A * a[0], A * a[1], A * a[2]};
  }
  SFB_LIE_HD Tangent log() const
  {
    const double s2 = x * x + y * y + z * z;
    double k;  // angle / sin(angle/2), with the shortest rotation (w >= 0 branch)
    const double ww = (w < 0) ? -w : w, sgn = (w < 0) ? -1.0 : 1.0;
    if (s2 < 1e-10) {
      k = 2.0 / ww - 2.0 / 3.0 * s2 / (ww * ww * ww);
    } else {
      const double sn = std::sqrt(s2);
      k               = 2.0 * std::atan2(sn, ww) / sn;
    }
    return {sgn * k * x, sgn * k * y, sgn * k * z};
  }
  SFB_LIE_HD SO3 inverse() const { return SO3{w, -x, -y, -z}; }
  SFB_LIE_HD friend SO3 operator*(const SO3 &a, const SO3 &b)
  {
    SO3 r{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
    const double nrm = std::sqrt(r.w * r.w + r.x * r.x + r.y * r.y + r.z * r.z);
    r.w /= nrm; r.x /= nrm; r.y /= nrm; r.z /= nrm;
    return r;
  }
  SFB_LIE_HD friend SO3 rplus(const SO3 &g, const Tangent &a) { return g * exp(a); }
  SFB_LIE_HD friend Tangent rminus(const SO3 &a, const SO3 &b) { return (b.inverse() * a).log(); }

  SFB_LIE_HD static Mat<3, 3> ad(const Tangent &a)  // = hat(a)
  {
    Mat<3, 3> m{};
    m(0, 1) = -a[2]; m(0, 2) = a[1];
    m(1, 0) = a[2];  m(1, 2) = -a[0];
    m(2, 0) = -a[1]; m(2, 1) = a[0];
    return m;
  }
  SFB_LIE_HD static Mat<3, 3> dr_expinv(const Tangent &a)
  {
    const double th2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2], th = std::sqrt(th2);
    const double k   = (th2 < 1e-8) ? (1.0 / 12.0 + th2 / 720.0) : (1.0 / th2 - (1.0 + std::cos(th)) / (2.0 * th * std::sin(th)));
    const Mat<3, 3> A = ad(a);
    return Mat<3, 3>::Identity() + 0.5 * A + k * (A * A);
  }
};

// ---- Bundle<G...>: direct product, tangent = concatenation (first part first) ----
template<class... Gs>
struct Bundle {
  static constexpr int Dof           = (Gs::Dof + ...);
  static constexpr bool IsCommutative = (Gs::IsCommutative && ...);
  using Tangent                      = Vec<Dof>;
  std::tuple<Gs...> parts{};

  SFB_LIE_HD static Bundle Identity() { return Bundle{}; }
  template<size_t I>
  SFB_LIE_HD auto &part() { return std::get<I>(parts); }
  template<size_t I>
  SFB_LIE_HD const auto &part() const { return std::get<I>(parts); }

  template<class F>
  SFB_LIE_HD static void for_parts(F &&f)
  {
    for_parts_impl(std::forward<F>(f), std::index_sequence_for<Gs...>{});
  }
  template<class F, size_t... I>
  SFB_LIE_HD static void for_parts_impl(F &&f, std::index_sequence<I...>)
  {
    int off = 0;
    ((f(std::integral_constant<size_t, I>{}, off), off += std::tuple_element_t<I, std::tuple<Gs...>>::Dof), ...);
  }
  template<int N, int O>
  SFB_LIE_HD static Vec<N> seg(const Tangent &a, int off)
  {
    (void)O;
    Vec<N> r{};
    for (int i = 0; i < N; ++i) r[i] = a[off + i];
    return r;
  }

  SFB_LIE_HD friend Bundle rplus(const Bundle &g, const Tangent &a)
  {
    Bundle r = g;
    for_parts([&](auto I, int off) {
      using G = std::tuple_element_t<decltype(I)::value, std::tuple<Gs...>>;
      std::get<decltype(I)::value>(r.parts) = rplus(std::get<decltype(I)::value>(g.parts), seg<G::Dof, 0>(a, off));
    });
    return r;
  }
  SFB_LIE_HD friend Tangent rminus(const Bundle &a, const Bundle &b)
  {
    Tangent t{};
    for_parts([&](auto I, int off) {
      using G       = std::tuple_element_t<decltype(I)::value, std::tuple<Gs...>>;
      const auto ti = rminus(std::get<decltype(I)::value>(a.parts), std::get<decltype(I)::value>(b.parts));
      for (int i = 0; i < G::Dof; ++i) t[off + i] = ti[i];
    });
    return t;
  }
  SFB_LIE_HD static Mat<Dof, Dof> ad(const Tangent &a) { return blockdiag(a, [](auto g, const auto &ai) { return decltype(g)::ad(ai); }); }
  SFB_LIE_HD static Mat<Dof, Dof> dr_expinv(const Tangent &a)
  {
    return blockdiag(a, [](auto g, const auto &ai) { return decltype(g)::dr_expinv(ai); });
  }

private:
  template<class F>
  SFB_LIE_HD static Mat<Dof, Dof> blockdiag(const Tangent &a, F &&f)
  {
    Mat<Dof, Dof> m{};
    for_parts([&](auto I, int off) {
      using G      = std::tuple_element_t<decltype(I)::value, std::tuple<Gs...>>;
      const auto b = f(G{}, seg<G::Dof, 0>(a, off));
      for (int c = 0; c < G::Dof; ++c)
        for (int r = 0; r < G::Dof; ++r) m(off + r, off + c) = b(r, c);
    });
    return m;
  }
};

// ---- components of a group as (kind, dof) pairs, for ad() on the device (sfb_lie_kind in sfb.h) ----
template<class G>
struct LieParts;
template<int N>
struct LieParts<Rn<N>> {
  static void append(std::vector<int32_t> &kind, std::vector<int32_t> &dof) { kind.push_back(0); dof.push_back(N); }
};
template<>
struct LieParts<SE2> {
  static void append(std::vector<int32_t> &kind, std::vector<int32_t> &dof) { kind.push_back(1); dof.push_back(3); }
};
template<>
struct LieParts<SO3> {
  static void append(std::vector<int32_t> &kind, std::vector<int32_t> &dof) { kind.push_back(2); dof.push_back(3); }
};
template<class... Gs>
struct LieParts<Bundle<Gs...>> {
  static void append(std::vector<int32_t> &kind, std::vector<int32_t> &dof) { (LieParts<Gs>::append(kind, dof), ...); }
};

}  // namespace smooth_feedback_amd
