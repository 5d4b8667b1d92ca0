"""Known-answer QPs of the reference's tests/test_qp.cpp (data only: inputs and expected outputs)."""
import numpy as np

inf = np.inf

_P3 = [[4, 2, 2], [2, 4, 2], [2, 2, 4]]
_PORT_P = [[0.018641, 0.00359853, 0.00130976], [0.00359853, 0.00643694, 0.00488727],
           [0.00130976, 0.00488727, 0.0686828]]

# name -> (P, q, A, l, u, expected_code, expected_primal or None, primal rel tol, expected objective or None, obj tol)
KNOWN_ANSWERS = {
    # tests/test_qp.cpp:54-73
    "Basic": (np.eye(2), [-4, 0.25], np.eye(2), [-1, -1], [1, 1], 0, [1, -0.25], 1e-4, 1. / 2 - 4 - 1. / 32, 1e-4),
    # :149-166
    "Unconstrained": (_P3, [-8, -6, -10], np.zeros((1, 3)), [-inf], [inf], 0, [1, 0, 2], 1e-4, None, None),
    # :168-185
    "HalfConstrained": (_P3, [-8, -6, -10], np.eye(3), [-inf, -inf, -10], [inf, 10, inf], 0, [1, 0, 2], 1e-4, None, None),
    # :187-199
    "PrimalInfeasibleEasy": (np.eye(2), [0.1, 0.1], np.eye(2), [-1, 1], [1, -1], 2, None, None, None, None),
    # :201-213
    "PrimalInfeasibleHard": (np.eye(2), [0.1, 0.1], [[1, 1], [-1, -1]], [0.5, 0.5], [1, 1], 2, None, None, None, None),
    # :215-227
    "PrimalInfeasibleInfinity": (np.eye(2), [0.1, 0.1], [[1, 1], [-1, -1], [1, 0], [0, 1]], [0.5, 0.5, -inf, -inf],
                                 [1, 1, inf, inf], 2, None, None, None, None),
    # :229-242
    "DualInfeasible": ([[1, 0], [0, 0]], [1, -1], np.eye(2), [-1, -inf], [1, inf], 3, None, None, None, None),
    # :244-275
    "PortfolioOptimization": (_PORT_P, [0, 0, 0],
                              [[1, 1, 1], [0.0260022, 0.00810132, 0.0737159], [1, 0, 0], [0, 1, 0], [0, 0, 1]],
                              [-inf, 50, 0, 0, 0], [1000, inf, inf, inf, inf], 0,
                              [497.04552984986384, 0.0, 502.9544801594811], 1e-4, 22634.417849884154 / 2, 5e-2),
    # :314-336
    "TwoDimensional": ([[0.0100131, 0], [0, 0.01]], [-0.329554, 0.536459], [[-0.0639209, -0.168], [-0.467, 0]],
                       [-inf, -inf], [-0.034974, 0.46571], 0, [46.6338, -17.5351], 1e-4, None, None),
}


def is_approx(a, b, tol):
    """Eigen's isApprox: ||a-b|| <= tol * min(||a||, ||b||)."""
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.linalg.norm(a - b) <= tol * min(np.linalg.norm(a), np.linalg.norm(b))


def as_batch(case):
    """-> flat col-major (1, .) buffers for the batch APIs"""
    P, q, A, l, u = (np.asarray(t, dtype=np.float64) for t in case[:5])
    return (P.flatten(order="F")[None], q[None], A.flatten(order="F")[None], l[None], u[None])
