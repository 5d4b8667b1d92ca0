"""The build guard of the sparse kernel (smooth_feedback_amd/csrc/check_sweep_spills.py, run by the Makefile on the device
assembly of qp_sparse.hip): the triangular sweeps keep stream loads and LDS reads IN FLIGHT across compiler-generated code and
count them by hand (s_waitcnt vmcnt(N) / lgkmcnt(N)), which the compiler does not know.  The guard refuses a build in which
(a) a register spill lies inside a sweep or (b) any instruction names a register whose load has not been waited for.  Here it
is run on hand-written assembly snippets: the guard itself must accept what is right and refuse what is wrong."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GUARD = os.path.join(ROOT, "smooth_feedback_amd", "csrc", "check_sweep_spills.py")

HEAD = """
\tglobal_load_dwordx4 v[2:5], v[40:41], off offset:0
\tglobal_load_dwordx2 v[80:81], v[42:43], off offset:0
\tglobal_load_dwordx4 v[6:9], v[40:41], off offset:1024
\tglobal_load_dwordx2 v[82:83], v[42:43], off offset:512
"""
TAIL = """
\tglobal_load_dwordx4 v[2:5], v[40:41], off offset:2048
\tglobal_load_dwordx2 v[80:81], v[42:43], off offset:1024
\ts_waitcnt vmcnt(0)
\tv_mov_b32_e32 v1, v2
"""


def run(body, tmp_path):
    path = tmp_path / "k.s"
    path.write_text(HEAD + body + TAIL)
    return subprocess.run([sys.executable, GUARD, str(path)], capture_output=True, text=True)


def test_accepts_a_correctly_counted_unit(tmp_path):
    body = """
\ts_waitcnt vmcnt(2)
\tv_add_u32_sdwa v50, v60, v80 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1
\tds_read_b64 v[90:91], v50
\tds_read_b64 v[92:93], v50
\ts_waitcnt lgkmcnt(1)
\tv_fma_f64 v[90:91], -v[2:3], v[90:91], v[90:91]
\ts_waitcnt lgkmcnt(0)
\tv_fma_f64 v[92:93], -v[4:5], v[92:93], v[92:93]
\tds_write_b64 v50, v[90:91]
"""
    r = run(body, tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "no instruction touches a register of a load in flight" in r.stdout


def test_refuses_a_spill_inside_a_sweep(tmp_path):
    r = run("\tscratch_store_dword off, v7, off offset:4\n", tmp_path)
    assert r.returncode != 0 and "spill inside a sweep" in (r.stdout + r.stderr)


def test_refuses_a_copy_of_a_stream_register_in_flight(tmp_path):
    # v[6:9] belongs to the second unit: after vmcnt(2) only the first unit's loads have landed
    r = run("\ts_waitcnt vmcnt(2)\n\tv_mov_b32_e32 v100, v6\n", tmp_path)
    assert r.returncode != 0 and "still in flight" in (r.stdout + r.stderr)


def test_refuses_a_use_of_an_lds_read_before_its_wait(tmp_path):
    body = """
\ts_waitcnt vmcnt(0)
\tds_read_b64 v[90:91], v50
\tds_read_b64 v[92:93], v51
\ts_waitcnt lgkmcnt(1)
\tv_fma_f64 v[94:95], v[92:93], v[2:3], v[90:91]
"""
    r = run(body, tmp_path)
    assert r.returncode != 0 and "still in flight" in (r.stdout + r.stderr)


def test_refuses_an_address_that_is_still_in_flight(tmp_path):
    r = run("\tds_read_b64 v[90:91], v80\n", tmp_path)  # v80: index word of unit 0, no wait yet
    assert r.returncode != 0 and "in flight" in (r.stdout + r.stderr)


def test_the_shipped_assembly_passes_when_it_has_been_built():
    s = os.path.join(ROOT, "smooth_feedback_amd", "csrc", "build", "qp_sparse.s")
    if not os.path.exists(s):
        import pytest
        pytest.skip("device assembly not built here")
    r = subprocess.run([sys.executable, GUARD, s], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
