"""The build guard of the sparse kernel (csrc/check_sweep_spills.py) on synthetic assembly: the rules it enforces on the real
device assembly at every build -- no scratch inside a sweep, no instruction naming a register of a stream load in flight, and
the AccVGPR file shared by number between the compiler and the resident factor stream (qp_sparse.hip kAgprFree)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GUARD = os.path.join(ROOT, "smooth_feedback_amd", "csrc", "check_sweep_spills.py")


def _agpr_free():
    src = open(os.path.join(ROOT, "smooth_feedback_amd", "csrc", "qp_sparse.hip")).read()
    return int(re.search(r"constexpr int kAgprFree\s*=\s*(\d+)\s*;", src).group(1))


def _run(lines, tmp_path):
    f = tmp_path / "k.s"
    f.write_text("\n".join(lines) + "\n")
    return subprocess.run([sys.executable, GUARD, str(f)], capture_output=True, text=True)


SWEEP = ["\tglobal_load_dwordx4 v[10:13], v[2:3], off offset:1024", "\ts_waitcnt vmcnt(0)", "\tv_fma_f64 v[20:21], v[10:11], v[12:13], v[20:21]"]


def test_clean_assembly_passes(tmp_path):
    r = _run(SWEEP, tmp_path)
    assert r.returncode == 0, r.stderr
    assert "none inside a sweep" in r.stdout


def test_scratch_inside_a_sweep_fails(tmp_path):
    r = _run([SWEEP[0], "\tscratch_store_dword off, v5, s32 offset:4", SWEEP[0].replace("1024", "2048"), SWEEP[1]], tmp_path)
    assert r.returncode != 0 and "register spill inside a sweep" in r.stderr


def test_register_of_a_load_in_flight_fails(tmp_path):
    r = _run([SWEEP[0], "\tv_mov_b32_e32 v30, v11", SWEEP[0].replace("v[10:13]", "v[14:17]").replace("1024", "2048"), SWEEP[1]], tmp_path)
    assert r.returncode != 0 and "still in flight" in r.stderr


def test_accvgpr_file_is_shared_by_number(tmp_path):
    free = _agpr_free()
    ours_ok = ["\tv_accvgpr_read_b32 v7, a[%d]" % free, "\tglobal_load_dwordx4 a[%d:0x%x], v[4:5], off" % (free, free + 3)]
    assert _run(SWEEP + ours_ok, tmp_path).returncode == 0
    # the compiler reaching into the resident stream's registers
    r = _run(SWEEP + ["\tv_accvgpr_write_b32 a%d, v6" % free], tmp_path)
    assert r.returncode != 0 and "not the resident stream's" in r.stderr
    r = _run(SWEEP + ["\tv_accvgpr_read_b32 v6, a255"], tmp_path)  # (the compiler prints register numbers without brackets)
    assert r.returncode != 0 and "not the resident stream's" in r.stderr
    if free > 0:
        assert _run(SWEEP + ["\tv_accvgpr_write_b32 a%d, v6" % (free - 1)], tmp_path).returncode == 0  # the compiler's share
        r = _run(SWEEP + ["\tv_accvgpr_read_b32 v7, a[%d]" % (free - 1)], tmp_path)  # ours below the line
        assert r.returncode != 0 and "belongs to the compiler" in r.stderr
