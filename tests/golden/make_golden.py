"""Generates tests/golden/qp_dense_random.npz: seeded random_qp inputs (benchmarks/bench_types.hpp
generator, std::default_random_engine(5)) and the CPU oracle's outputs for two parameter sets.
Run from the repo root:  python tests/golden/make_golden.py
The reference itself cannot be built in this environment (no Eigen), so these vectors pin the
oracle <-> HIP kernel agreement, not the oracle <-> reference one (see oracle/qp_oracle.h)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import loader as O  # noqa: E402
import smooth_feedback_amd as sfb  # noqa: E402

PARAMS = {
    # library defaults (qp_solver.hpp:29-68)
    "default": {},
    # benchmarks/bench.cpp:148-153
    "bench": {"eps_abs": 1e-6, "eps_rel": 1e-6, "polish": 1, "max_iter": 10000, "scaling": 0},
}


def main():
    B, m, n = 96, 20, 10
    parts = [sfb.random_qp_batch(5, B // 3, m, n, d) for d in (0.05, 0.3, 1.0)]
    P, q, A, l, u = (np.concatenate([p[i] for p in parts]) for i in range(5))
    out = dict(P=P, q=q, A=A, l=l, u=u, params_json=json.dumps(PARAMS))
    for tag, kw in PARAMS.items():
        r = O.qp_dense_solve_batch(P, q, A, l, u, params=O.default_params(**kw))
        for k in ("code", "iter", "x", "y", "obj"):
            out["%s_%s" % (tag, k)] = r[k]
        print(tag, "codes", np.bincount(r["code"], minlength=7), "iters", np.unique(r["iter"]))
    np.savez_compressed(os.path.join(os.path.dirname(__file__), "qp_dense_random.npz"), **out)


if __name__ == "__main__":
    main()
