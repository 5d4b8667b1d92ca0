"""The C-ABI library loads on a CPU-only box and exports every symbol include/sfb.h declares.
No compute call is made here (there is no CPU fallback to call)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    inc = os.path.join(ROOT, "include")
    for fn in os.listdir(inc):
        if fn.endswith(".h"):
            txt = open(os.path.join(inc, fn)).read()
            txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
            names |= set(re.findall(r"\b(sfb_[a-z0-9_]+)\s*\(", txt))
    return names


def test_exports_every_declared_symbol(sfb):
    lib = ctypes.CDLL(sfb._capi.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 6
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing


def test_params_struct_matches_oracle_layout(sfb, oracle):
    a, b = sfb._capi.SfbQPParams, oracle.OracleQPParams
    assert ctypes.sizeof(a) == ctypes.sizeof(b)
    assert [(n, t) for n, t in a._fields_] == [(n, t) for n, t in b._fields_]
    pa, pb = a(), b()
    sfb._capi.lib.sfb_qp_params_default(ctypes.byref(pa))
    oracle.lib().oracle_qp_params_default(ctypes.byref(pb))
    assert bytes(pa) == bytes(pb)


def test_argument_errors_do_not_need_a_device(sfb):
    import numpy as np
    P, q, A, l, u = sfb.random_qp_batch(5, 2, 20, 10, 1.0)
    # sizes the solver cannot take are rejected before any device work
    try:
        sfb.solve_qp_batch_host(np.zeros((1, 0)), np.zeros((1, 0)), np.zeros((1, 0)), np.zeros((1, 3)), np.zeros((1, 3)))
        assert False
    except sfb._capi.SfbError as e:
        assert e.status == sfb._capi.SFB_ERR_INVALID_ARG
    # an iteration limit beyond uint32 is rejected
    try:
        sfb.solve_qp_batch_host(P, q, A, l, u, sfb.QPSolverParams(max_iter=2 ** 33))
        assert False
    except sfb._capi.SfbError as e:
        assert e.status in (sfb._capi.SFB_ERR_INVALID_ARG, sfb._capi.SFB_ERR_NO_DEVICE)


def test_no_silent_cpu_fallback(sfb):
    """Without a GPU the compute entry point must FAIL, never compute on the CPU."""
    if sfb._capi.device_count() > 0:
        return
    P, q, A, l, u = sfb.random_qp_batch(5, 2, 20, 10, 1.0)
    try:
        sfb.solve_qp_batch_host(P, q, A, l, u)
        assert False, "compute call succeeded without a device"
    except sfb._capi.SfbError as e:
        assert e.status in (sfb._capi.SFB_ERR_NO_DEVICE, sfb._capi.SFB_ERR_HIP)


def test_random_qp_generator_properties(sfb):
    """benchmarks/bench_types.hpp:19-41: l = -inf, P = L L' symmetric PSD, u = A v + delta."""
    import numpy as np
    P, q, A, l, u = sfb.random_qp_batch(5, 8, 20, 10, 0.3)
    assert np.all(np.isneginf(l)) and np.all(np.isfinite(u))
    for b in range(8):
        Pb = P[b].reshape(10, 10, order="F")
        assert np.allclose(Pb, Pb.T) and np.linalg.eigvalsh(Pb).min() > -1e-12
    assert np.all(np.abs(q) <= 1) and np.all(np.abs(A) <= 1)
    assert 0.15 < np.mean(A != 0) < 0.45
    P2, *_ = sfb.random_qp_batch(5, 8, 20, 10, 0.3)
    assert np.array_equal(P, P2)


def test_header_is_plain_c_and_fronts_compile_standalone(tmp_path):
    """include/sfb.h is the C-ABI: it must compile as C99 on its own; the C++ front -- under the reference's include
    paths and namespace -- must compile without anything else of this repo (header-only, like the reference)."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("gcc") is None or shutil.which("g++") is None:
        import pytest
        pytest.skip("no host compiler")
    c = tmp_path / "abi.c"
    c.write_text('#include <sfb.h>\nint main(void) { sfb_qp_params p; sfb_mpc_layout l; sfb_workspace *w = 0; (void)p; (void)l; (void)w; return 0; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"), "-c", str(c),
                    "-o", str(tmp_path / "abi.o")], check=True)
    cpp = tmp_path / "front.cpp"
    cpp.write_text("#include <smooth/feedback/qp.hpp>\n#include <smooth/feedback/qp_solver.hpp>\n#include <smooth/feedback/ocp.hpp>\n"
                   "#include <smooth/feedback/ocp_to_qp.hpp>\n#include <smooth/feedback/mpc.hpp>\n#include <smooth/feedback/ekf.hpp>\n"
                   "#include <smooth/feedback/asif.hpp>\n#include <smooth/feedback/mesh.hpp>\n"
                   "int main() { smooth::feedback::QPSolverParams p; smooth::feedback::QuadraticProgram<2, 3> q; (void)p; (void)q; return 0; }\n")
    subprocess.run(["g++", "-std=c++20", "-Wall", "-fsyntax-only", "-I", os.path.join(root, "include"), str(cpp)], check=True)


def test_debug_knobs_go_through_one_entry_point_and_never_the_environment(sfb):
    """sfb_debug_set is the only way to steer a launch shape / engine choice: it knows its names (csrc/knobs.h), and no
    translation unit of libsfb.so reads the environment."""
    import glob
    import os
    import re
    from smooth_feedback_amd import _capi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert _capi.lib.sfb_debug_set(b"SFB_SP_GRID", b"4") == _capi.SFB_OK
    assert _capi.lib.sfb_debug_set(b"SFB_SP_GRID", None) == _capi.SFB_OK          # cleared
    assert _capi.lib.sfb_debug_set(b"SFB_NO_SUCH_KNOB", b"1") == _capi.SFB_ERR_INVALID_ARG
    assert "unknown knob" in _capi.lib.sfb_last_error().decode()
    assert _capi.lib.sfb_debug_set(None, b"1") == _capi.SFB_ERR_INVALID_ARG
    names = set(re.findall(r'"(SFB_[A-Z0-9_]+)"', open(os.path.join(root, "smooth_feedback_amd", "csrc", "knobs.cpp")).read()))
    assert 10 <= len(names) <= 24
    used = set()
    for f in glob.glob(os.path.join(root, "smooth_feedback_amd", "csrc", "*")):
        if os.path.isfile(f) and f.endswith((".hip", ".cpp", ".h")):
            text = open(f).read()
            assert "getenv" not in text, f                                          # the library reads no environment variable
            used |= set(re.findall(r'knob\("(SFB_[A-Z0-9_]+)"\)', text))
    assert used <= names, used - names                                              # every knob that is read can be set
