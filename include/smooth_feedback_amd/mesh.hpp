// Legendre-Gauss-Radau collocation mesh on [0,1] (host side of the MPC path).
// Mirrors what MPC needs from smooth::feedback::Mesh<Kmesh,Kmesh> (reference
// collocation/mesh.hpp:93-118 equal intervals, :208-297 nodes/weights, :343-365 unscaled
// differentiation matrix) and smooth::lgr_nodes (mesh.hpp:35-48, third party).
#pragma once
#include <cmath>
#include <cstddef>
#include <utility>
#include <vector>

namespace smooth_feedback_amd {

// LGR nodes on [-1,1) and weights for K points: roots of P_{K-1}(x) + P_K(x), node 0 = -1.
inline void lgr_nodes(int K, std::vector<double> &x, std::vector<double> &w)
{
  x.assign(K, 0.0);
  w.assign(K, 0.0);
  auto legendre = [](int n, double t, double &pn, double &pnm1) {  // P_n, P_{n-1}
    double p0 = 1.0, p1 = t;
    if (n == 0) { pn = 1.0; pnm1 = 0.0; return; }
    for (int k = 2; k <= n; ++k) {
      const double p2 = ((2.0 * k - 1.0) * t * p1 - (k - 1.0) * p0) / k;
      p0 = p1; p1 = p2;
    }
    pn = p1; pnm1 = p0;
  };
  x[0] = -1.0;
  for (int i = 1; i < K; ++i) {
    // Chebyshev-like initial guess, then Newton on g(t) = (P_{K-1}(t) + P_K(t)) / (1 + t)
    double t = -std::cos(2.0 * M_PI * i / (2.0 * K - 1.0));
    for (int it = 0; it < 100; ++it) {
      double pK, pKm1;
      legendre(K, t, pK, pKm1);
      const double g = pKm1 + pK;
      // derivatives: (1-t^2) P_n' = n (P_{n-1} - t P_n)
      double pKm2, dummy;
      legendre(K - 1, t, dummy, pKm2);
      const double dK   = K * (pKm1 - t * pK) / (1.0 - t * t);
      const double dKm1 = (K - 1) * (pKm2 - t * pKm1) / (1.0 - t * t);
      // deflate the known root at -1
      const double f = g / (1.0 + t), df = ((dK + dKm1) * (1.0 + t) - g) / ((1.0 + t) * (1.0 + t));
      const double step = f / df;
      t -= step;
      if (std::fabs(step) < 1e-16) break;
    }
    x[i] = t;
  }
  w[0] = 2.0 / (double(K) * K);
  for (int i = 1; i < K; ++i) {
    double pK, pKm1;
    legendre(K, x[i], pK, pKm1);
    w[i] = (1.0 - x[i]) / (double(K) * K * pKm1 * pKm1);
  }
}

// Mesh with `n` equal intervals of K LGR points each (Mesh<K,K>(n), mesh.hpp:93-104).
struct Mesh {
  int n_ivals = 1, K = 4;
  std::vector<double> tau, wts;  // LGR nodes/weights on [-1,1]; extra node +1 with weight 0 appended
  std::vector<double> Dus;       // (K+1) x K, column-major: Dus(j,i) = l_j'(tau_i)

  Mesh() : Mesh(1, 4) {}
  Mesh(int n, int k) : n_ivals(n < 1 ? 1 : n), K(k)
  {
    lgr_nodes(K, tau, wts);
    tau.push_back(1.0);  // lgr_plus_one, mesh.hpp:35-48
    wts.push_back(0.0);
    Dus.assign((size_t)(K + 1) * K, 0.0);
    for (int i = 0; i < K; ++i)       // evaluation node
      for (int j = 0; j <= K; ++j) {  // basis function
        double v = 0.0;
        if (j == i) {
          for (int k2 = 0; k2 <= K; ++k2)
            if (k2 != i) v += 1.0 / (tau[i] - tau[k2]);
        } else {
          double num = 1.0, den = 1.0;
          for (int k2 = 0; k2 <= K; ++k2) {
            if (k2 != j) den *= (tau[j] - tau[k2]);
            if (k2 != j && k2 != i) num *= (tau[i] - tau[k2]);
          }
          v = num / den;
        }
        Dus[(size_t)j + (size_t)i * (K + 1)] = v;
      }
  }
  int N_ivals() const { return n_ivals; }
  int N_colloc() const { return n_ivals * K; }
  double interval_start(int s) const { return double(s) / double(n_ivals); }
  // node tau in [0,1] of global index i in 0..N (N = the extra end point)   mesh.hpp:208-262
  double node(int i) const
  {
    if (i >= N_colloc()) return 1.0;
    const int s = i / K, nu = i % K;
    const double tau0 = interval_start(s), tauf = (s + 1 < n_ivals) ? interval_start(s + 1) : 1.0;
    return tau0 + (tauf - tau0) / 2 * (tau[nu] + 1.0);
  }
  double weight(int i) const  // mesh.hpp:264-297
  {
    if (i >= N_colloc()) return 0.0;
    const int s = i / K, nu = i % K;
    const double tau0 = interval_start(s), tauf = (s + 1 < n_ivals) ? interval_start(s + 1) : 1.0;
    return (tauf - tau0) / 2 * wts[nu];
  }
  // D = alpha * Dus  (mesh.hpp:343-365)
  double alpha(int s) const
  {
    const double tau0 = interval_start(s), tauf = (s + 1 < n_ivals) ? interval_start(s + 1) : 1.0;
    return 2.0 / (tauf - tau0);
  }
  double D(int j, int i) const { return Dus[(size_t)j + (size_t)i * (K + 1)]; }
};

}  // namespace smooth_feedback_amd
