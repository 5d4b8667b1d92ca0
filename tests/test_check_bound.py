"""The sparse kernel's stopping check decides the primal-infeasibility test from a bound on the ordered certificate sum
(qp_solver.hpp:607-621; csrc/qp_sparse.hip sp_check_stopping, FAST PATH): the terms summed in any order differ from the ordered
sum by at most 2 gamma_N S, S = sum |terms|.  This restates the decision on the host (same partial sums: lane-strided, then the
tree of wave_sum) and checks, on adversarial term vectors, that whenever the bound decides it agrees with the ordered sum the
oracle forms -- the property the kernel's bit-identical verdicts rest on."""
import numpy as np


def ordered_sum(terms):
    acc = np.float64(0.0)
    for v in terms:
        acc = acc + v
    return acc


def wave_sum(v):  # csrc/wave_util.h wave_sum: row rotations 8, 4, 2, 1, then (r0 + r1) + (r2 + r3)
    v = v.reshape(4, 16).copy()
    for sh in (8, 4, 2, 1):
        v = v + np.roll(v, sh, axis=1)
    return (v[0, 0] + v[1, 0]) + (v[2, 0] + v[3, 0])


def fast_side(ta, tb, thr):
    """0: ordered sum >= thr, 1: < thr, 2: undecided (the kernel then forms the ordered sum)"""
    m = len(ta)
    ps, pa = np.zeros(64), np.zeros(64)
    for i in range(m):
        ln = i % 64
        ps[ln] = ps[ln] + ta[i]; ps[ln] = ps[ln] + tb[i]
        pa[ln] = pa[ln] + abs(ta[i]); pa[ln] = pa[ln] + abs(tb[i])
    s, S = wave_sum(ps), wave_sum(pa)
    err = 4.0 * (2.0 * m) * np.finfo(np.float64).eps * S
    if not (S < np.inf):
        return 2
    if s - err >= thr:
        return 0
    if s + err < thr:
        return 1
    return 2


def test_bound_never_contradicts_the_ordered_sum():
    rng = np.random.default_rng(20261201)
    decided = undecided = 0
    with np.errstate(all="ignore"):
        for case in range(400):
            m = int(rng.integers(1, 900))
            kind = case % 8
            scale = 10.0 ** rng.uniform(-12, 12, 2 * m) if kind in (0, 1) else np.ones(2 * m)
            t = rng.standard_normal(2 * m) * scale
            if kind == 2:  # massive cancellation: pairs that nearly cancel
                t[1::2] = -t[0::2] * (1.0 + rng.uniform(-1e-15, 1e-15, m))
            if kind == 3:  # one huge term and its negative far apart, small terms in between
                t[0], t[-1] = 1e18, -1e18
            if kind == 4:
                t[rng.integers(0, 2 * m)] = np.inf if rng.random() < 0.5 else np.nan
            if kind == 5:
                t *= 1e-310  # subnormal terms
            ta, tb = t[0::2].copy(), t[1::2].copy()
            exact = ordered_sum(t)  # the kernel's (and the oracle's) order: ta[0], tb[0], ta[1], ...
            # thresholds: far, near, and exactly at the ordered sum and its neighbours
            thrs = [0.0, 1e-4 * abs(t).max(), exact, np.nextafter(exact, np.inf), np.nextafter(exact, -np.inf),
                    exact * (1 + 1e-13), exact * (1 - 1e-13), np.nan, np.inf]
            for thr in thrs:
                side = fast_side(ta, tb, np.float64(thr))
                if side == 0:
                    assert exact >= thr, (case, thr, exact)
                    decided += 1
                elif side == 1:
                    assert exact < thr, (case, thr, exact)
                    decided += 1
                else:
                    undecided += 1
    assert decided > 500 and undecided > 100  # both outcomes exercised


def test_known_side_makes_the_exact_sum_unnecessary():
    """acc < thr known  =>  [max(|A'dy|, acc) < thr]  ==  [|A'dy| < thr]  for every |A'dy| incl. NaN (kernel: side == 1)"""
    rng = np.random.default_rng(7)
    for _ in range(2000):
        thr = rng.standard_normal() * 10
        acc = thr - abs(rng.standard_normal()) - 1e-300
        for aty in (rng.standard_normal() * 10, thr, acc, np.nan, np.inf, 0.0):
            mxv = acc if aty < acc else aty  # the oracle's expression
            assert (mxv < thr) == (aty < thr)
