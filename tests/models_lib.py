"""ctypes loader of the example/test harness library (examples/models.cpp -> libsfb_models.so)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "smooth_feedback_amd", "libsfb_models.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(PATH):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s"])
        import smooth_feedback_amd  # noqa: F401  (loads libsfb.so / torch's HIP runtime first)
        L = C.CDLL(PATH)
        L.sfbx_lie_selftest.restype = C.c_double
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def mpc_dims(variant, K):
    v = [C.c_int() for _ in range(7)]
    assert lib().sfbx_mpc_dims(variant, K, *[C.byref(x) for x in v]) == 0
    return dict(zip(("n", "m", "nnzP", "nnzA", "Nx", "Nu", "N"), [x.value for x in v]))


def mpc_pattern(variant, K, tf=5.0):
    d = mpc_dims(variant, K)
    Pp = np.zeros(d["n"] + 1, np.int32); Pi = np.zeros(d["nnzP"], np.int32); Pv = np.zeros(d["nnzP"])
    Ap = np.zeros(d["m"] + 1, np.int32); Aj = np.zeros(d["nnzA"], np.int32)
    assert lib().sfbx_mpc_pattern(variant, K, C.c_double(tf), _p(Pp), _p(Pi), _p(Pv), _p(Ap), _p(Aj)) == 0
    return d, Pp, Pi, Pv, Ap, Aj


def mpc_assemble_batch(variant, K, batch, seed=0, tf=5.0, threads=8):
    d = mpc_dims(variant, K)
    Av = np.zeros((batch, d["nnzA"])); l = np.zeros((batch, d["m"])); u = np.zeros((batch, d["m"]))
    assert lib().sfbx_mpc_assemble_batch(variant, K, C.c_double(tf), C.c_int64(batch), C.c_uint64(seed), _p(Av), _p(l),
                                         _p(u), threads) == 0
    return Av, l, u


def mesh(n_ivals, K):
    N = n_ivals * K
    nodes = np.zeros(N + 1); w = np.zeros(N + 1); D = np.zeros((K + 1) * K)
    assert lib().sfbx_mesh(n_ivals, K, _p(nodes), _p(w), _p(D)) == 0
    return nodes, w, D.reshape(K, K + 1).T  # D[j, i]


def mpc_stage(variant, K):
    d = mpc_dims(variant, K)
    st = np.zeros(d["n"] + d["m"], np.int32)
    assert lib().sfbx_mpc_stage(variant, K, _p(st)) == 0
    return st
