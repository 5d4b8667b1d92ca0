// Time types of the controller fronts.  Mirrors the reference's `Time` concept and `time_trait` (time.hpp:25-89): a
// time type T is anything for which time_trait<T>::plus(t, seconds) -> T and time_trait<T>::minus(t2, t1) -> seconds
// (double) exist.  Provided here: floating point (seconds), std::chrono::time_point, std::chrono::duration.
// The numeric path only ever sees doubles: MPC<..., T> converts at its API boundary.
#pragma once
#include <chrono>
#include <concepts>
#include <type_traits>

namespace smooth_feedback_amd {

template<class T>
struct time_trait;  // specialise for your own clock type

template<class T>
concept Time = requires(T a, T b, double s) {
  { time_trait<T>::plus(a, s) } -> std::convertible_to<T>;
  { time_trait<T>::minus(b, a) } -> std::convertible_to<double>;
};

/// seconds as a floating-point number
template<std::floating_point F>
struct time_trait<F> {
  static constexpr F plus(F t, double s) { return t + static_cast<F>(s); }
  static constexpr double minus(F later, F earlier) { return static_cast<double>(later - earlier); }
};

namespace detail {
using fseconds = std::chrono::duration<double>;
}

/// a point on some std::chrono clock
template<class Clock, class Dur>
struct time_trait<std::chrono::time_point<Clock, Dur>> {
  using TP = std::chrono::time_point<Clock, Dur>;
  static constexpr TP plus(TP t, double s) { return t + std::chrono::duration_cast<Dur>(detail::fseconds(s)); }
  static constexpr double minus(TP later, TP earlier) { return std::chrono::duration_cast<detail::fseconds>(later - earlier).count(); }
};

/// a std::chrono duration since an epoch of the caller's choosing
template<class Rep, class Period>
struct time_trait<std::chrono::duration<Rep, Period>> {
  using D = std::chrono::duration<Rep, Period>;
  static constexpr D plus(D t, double s) { return t + std::chrono::duration_cast<D>(detail::fseconds(s)); }
  static constexpr double minus(D later, D earlier) { return std::chrono::duration_cast<detail::fseconds>(later - earlier).count(); }
};

static_assert(Time<double> && Time<float>);
static_assert(Time<std::chrono::steady_clock::time_point> && Time<std::chrono::nanoseconds>);

}  // namespace smooth_feedback_amd
