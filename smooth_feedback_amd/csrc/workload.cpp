// Synthetic workload generators (host, CPU only) restating the reference's benchmark inputs.
#include <cmath>
#include <limits>
#include <random>
#include <vector>

#include "../../include/sfb.h"

extern "C" sfb_status sfb_random_qp_batch(uint32_t seed, int64_t batch, int m, int n, double density, double *P,
                                          double *q, double *A, double *l, double *u)
{
  if (batch < 0 || m < 1 || n < 1 || !P || !q || !A || !l || !u) return SFB_ERR_INVALID_ARG;
  // benchmarks/bench.cpp:146  std::default_random_engine rng(5);  one engine for the whole batch (:170)
  std::default_random_engine rng(seed);
  // benchmarks/bench_types.hpp:22-23
  std::bernoulli_distribution bdist(density);
  std::uniform_real_distribution<double> udist(-1, 1);
  std::vector<double> L((size_t)n * n), v((size_t)n), delta((size_t)m);
  for (int64_t b = 0; b < batch; ++b) {
    double *Ab = A + (size_t)b * m * n, *Pb = P + (size_t)b * n * n;
    double *qb = q + (size_t)b * n, *lb = l + (size_t)b * m, *ub = u + (size_t)b * m;
    // :25  A = NullaryExpr(m, n, ...): Eigen fills a col-major matrix column by column
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < m; ++i) Ab[i + (size_t)j * m] = bdist(rng) ? udist(rng) : 0.;
    // :26-29  Lrand (n x n, all entries drawn), lower triangle kept, diagonal >= 0.05 in magnitude
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < n; ++i) {
        const double r        = bdist(rng) ? udist(rng) : 0.;
        L[i + (size_t)j * n]  = (i >= j) ? r : 0.;
      }
    for (int i = 0; i < n; ++i) L[i + (size_t)i * n] = std::max({L[i + (size_t)i * n], -L[i + (size_t)i * n], 0.05});
    // :31-32
    for (int i = 0; i < n; ++i) v[i] = udist(rng);
    for (int i = 0; i < m; ++i) delta[i] = udist(rng);
    // :35  P = L * L'   (plain ascending sums; the inputs are what they are)
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < n; ++i) {
        double s = 0;
        for (int t = 0; t < n; ++t) s += L[i + (size_t)t * n] * L[j + (size_t)t * n];
        Pb[i + (size_t)j * n] = s;
      }
    // :36  q drawn after v and delta (designated initialisers evaluate in declaration order)
    for (int i = 0; i < n; ++i) qb[i] = udist(rng);
    // :38-39
    for (int i = 0; i < m; ++i) {
      lb[i]    = -std::numeric_limits<double>::infinity();
      double s = 0;
      for (int j = 0; j < n; ++j) s += Ab[i + (size_t)j * m] * v[j];
      ub[i] = s + delta[i];
    }
  }
  return SFB_OK;
}
