"""bench.py contract on a GPU box: one JSON line with the required keys (N = 1), and the N > 1 code path
(barrier, gather of the small outputs, max-over-ranks timing) with two ranks sharing cuda:0 over gloo
(SFB_BENCH_SHARE_DEVICE=1; the driver runs the real thing over RCCL on 8 GPUs)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_has_roofline_and_cpu_baseline():
    out = subprocess.run([sys.executable, "bench.py", "--workload", "qp_dense", "--batch", "8192", "--steps", "2",
                          "--warmup", "1"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    assert KEYS <= set(d) and {"cpu_baseline", "parity_vs_oracle"} <= set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["dtype"] == "f64" and d["value"] > 0
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"])
    assert d["parity_vs_oracle"]["code_mismatches"] == 0 and d["parity_vs_oracle"]["iter_mismatches"] == 0
    assert d["parity_vs_oracle"]["max_abs_dx"] <= 1e-8


@pytest.mark.parametrize("workload,batch,port", [("qp_dense", 4096, 29533), ("mpc", 96, 29541), ("ekf", 70000, 29549)])
def test_two_ranks_code_path(workload, batch, port):
    """Every workload through the N > 1 path: per-rank shard, barrier, ONE gather of the small per-item outputs per
    step (for mpc: u_0, code, iter sliced out of the solution), max-over-ranks timing; rank 0 checks that its own
    rows come back from the gather unchanged and that the other rank's rows reproduce that rank's checksum."""
    env = dict(os.environ, SFB_BENCH_SHARE_DEVICE="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2",
                          "--workload", workload, "--batch", str(batch), "--steps", "2", "--warmup", "1"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    assert KEYS <= set(d)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 2 * batch and d["config"]["per_gpu_batch"] == batch
    assert d["value"] > 0 and "cpu_baseline" not in d and "secondary" not in d
    assert d["gather_check"] == {"own_rows_intact": True, "peer_rows_received": True, "peer_rows_differ_from_own": True,
                                 "gathered_rows": 2 * batch, "ranks_checked": 2}


def test_eight_ranks_mpc_code_path():
    """The shape of BASELINE configs[3] (8 ranks, one shard of MPC agents each, ONE gather of u_0 / code / iter per
    step) with all ranks on cuda:0 over gloo: sharding.shard_range over the job's batch, rank-derived seeds, host assembly threads divided among the
    ranks, the gather checked against per-rank checksums that travel by a separate all_reduce."""
    env = dict(os.environ, SFB_BENCH_SHARE_DEVICE="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                          "--master-addr", "127.0.0.1", "--master-port", "29557", "bench.py", "--gpus", "8",
                          "--workload", "mpc", "--batch", "64", "--steps", "2", "--warmup", "1"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["config"]["global_batch"] == 8 * 64
    assert d["gather_check"] == {"own_rows_intact": True, "peer_rows_received": True, "peer_rows_differ_from_own": True,
                                 "gathered_rows": 8 * 64, "ranks_checked": 8}


def test_default_line_carries_the_secondary_workloads():
    """The driver's command (no flags): headline = BASELINE configs[2], plus the other two configurations under
    `secondary`, each with its own roofline, CPU baseline and parity figures; the dense one with the FP64-VALU view."""
    out = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1"], cwd=ROOT, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    assert d["config"]["workload"].startswith("mpc_qp_nx12_nu2_K50_b8192")
    assert set(d["secondary"]) == {"qp_dense", "ekf", "qp_dense_sizes"}
    sizes = d["secondary"].pop("qp_dense_sizes")
    assert {(r["n"], r["m"]) for r in sizes} == {(10, 20), (16, 32), (20, 40), (32, 32), (32, 64), (40, 60), (64, 64)}
    for r in sizes:   # every size of the north star's class: measured, and bit-identical to the oracle on its sample
        assert r["qp_per_s"] > 0 and r["parity_vs_oracle"] == {"sample": r["parity_vs_oracle"]["sample"], "code_mismatches": 0,
                                                                "iter_mismatches": 0, "max_abs_dx": 0.0}
    for k, v in d["secondary"].items():
        assert v["value"] > 0 and {"bound", "achieved", "peak", "frac", "traffic", "kernel_ms"} <= set(v["roofline"])
        assert "cpu_baseline" in v and "parity_vs_oracle" in v
    assert d["secondary"]["qp_dense"]["roofline_alt"]["bound"] == "fp64_valu"
    assert d["secondary"]["ekf"]["parity_vs_oracle"]["P_bit_identical"]
    assert d["parity_vs_oracle"]["iter_mismatches"] == 0 and d["parity_vs_oracle"]["max_abs_dx"] == 0.0
    assert d["cpu_baseline"]["single_core"]["cores"] == 1
    # what a user of MPC::operator() sees (end-to-end incl. H2D), driver-timed in the same run
    cl = d["closed_loop"]
    assert cl["end_to_end"]["results_identical_to_device_resident"] and cl["end_to_end"]["value"] > 0
    assert cl["swarm_tick"]["agents"] == 8192 and cl["swarm_tick"]["ms_per_tick"] > 0 and cl["swarm_tick"]["optimal_fraction_last_tick"] == 1.0
    assert cl["single_agent"]["cold_ms"] > 0 and cl["single_agent"]["warm_iterations"] <= cl["single_agent"]["cold_iterations"]
    ekf_cpu = d["secondary"]["ekf"]["cpu_baseline"]
    assert ekf_cpu["single_core"]["cores"] == 1 and d["secondary"]["ekf"]["parity_vs_oracle"].get("all_cores_P_bit_identical", True)


@pytest.mark.parametrize("workload,batch,dtype,port", [("mpc", 96, "float64", 29565), ("qp_dense", 4096, "int32", 29573)])
def test_rccl_carries_the_collectives_of_the_n_gpu_path(workload, batch, dtype, port):
    """The `nccl` (= RCCL) branch of bench.py on the one GPU there is: one rank under torch.distributed.run, process group on
    RCCL with the device bound, and every collective of the N > 1 path issued for real (SFB_BENCH_FORCE_COLLECTIVES=1: the
    world-of-one shortcut of sharding.gather_small_outputs is bypassed) -- all_gather of the small per-item outputs (float64
    (batch, 4) rows of u0 / code / iter for mpc, int32 (batch, 2) for the dense QPs), the int64 checksum all_reduce, the
    barrier, the float64 MAX all_reduce of the elapsed time.  What an 8-GPU run adds is peers, not code."""
    env = dict(os.environ, SFB_BENCH_FORCE_COLLECTIVES="1")
    env.pop("SFB_BENCH_SHARE_DEVICE", None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "1",
                          "--workload", workload, "--batch", str(batch), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                          "--no-pipelined", "--no-secondary", "--no-closed-loop"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert d["gather_check"] == {"own_rows_intact": True, "peer_rows_received": True, "peer_rows_differ_from_own": True,
                                 "gathered_rows": batch, "ranks_checked": 1, "backend": "nccl", "gathered_dtype": dtype,
                                 "checksum_via_all_reduce_matches": True}
