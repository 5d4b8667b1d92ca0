#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native batched QP engine.

    python bench.py --gpus N --steps K --warmup W [--workload mpc|qp_dense|ekf]

One "step" = one pass of the hot path over one batch of synthetic QPs that is already resident in
HBM: by default the MPC configuration the BASELINE metric is quoted on (8 192 agents, nx=12, nu=2,
K=50 -> sparse QPs with n = m = 740, sfb_sparse_qp_solve_batch); `--workload qp_dense` runs
BASELINE configs[1] (65 536 dense QPs n=10, m=20, sfb_qp_dense_solve_batch).  For N > 1 the driver launches one rank per GPU with
torch.distributed.run; every rank owns an independent shard of the batch (weak scaling: per-GPU
batch fixed), the data path has no collective, and only the per-QP (code, iter) words are
gathered over RCCL at the end of each step.  Rank 0 prints ONE JSON line.

The single-GPU default run also reports the other two BASELINE configurations (`secondary`: dense QPs,
EKF ticks) with their own kernel time, roofline and parity figures, so that ONE driver-run line covers all three.

The CPU oracle (oracle/) is used here ONLY for the `cpu_baseline` leg and a parity spot-check of a
bounded sample; it is never the thing measured as `value`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_BYTES_PER_S = 8.0e12  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def host_cpus():
    """What this process may really use of the host: the scheduler affinity mask and the cgroup CPU quota, not the
    number of CPUs the machine has (os.cpu_count()).  Returns (threads to use, description for the bench line)."""
    logical = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = logical
    quota = None
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),                       # cgroup v2: "max 100000" / "800000 100000"
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: [t.strip(), None])):  # cgroup v1
        try:
            with open(path) as f:
                a, b = parse(f.read())
            if b is None:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    b = f.read().strip()
            if a not in ("max", "-1"):
                quota = float(a) / float(b)
            break
        except (OSError, ValueError):
            continue
    usable = affinity if quota is None else max(1, min(affinity, int(quota + 0.5)))
    return usable, {"os_cpu_count": logical, "sched_affinity": affinity, "cgroup_cpu_quota": quota, "threads_used": usable}


def qp_dense_algorithmic_bytes(n, m):
    """P,q,A,l,u in; x,y,obj,iter,code out."""
    return 8 * (n * n + n + m * n + 2 * m) + 8 * (n + m + 1) + 8


class DenseQPWorkload:
    """BASELINE.json configs[1]: batch = 65 536 random dense fp64 QPs, n=10, m=20, generator and
    solver parameters of benchmarks/bench.cpp:146-153 / bench_types.hpp:19-41 (density 1.0)."""
    name = "qp_dense_n10_m20_b65536_density1.0_benchparams"

    def __init__(self, sfb, rank, device, batch=65536, n=10, m=20, density=1.0):
        self.sfb, self.B, self.n, self.m = sfb, batch, n, m
        self.prm = sfb.QPSolverParams(eps_abs=1e-6, eps_rel=1e-6, polish=True, max_iter=10000, scaling=False)
        # std::default_random_engine(5) for rank 0 (the reference's seed), 5 + rank for the other shards
        self.host = sfb.random_qp_batch(5 + rank, batch, m, n, density)
        self.dev = [torch.from_numpy(a).to(device) for a in self.host]
        f64 = dict(dtype=torch.float64, device=device)
        self.x = torch.empty((batch, n), **f64)
        self.y = torch.empty((batch, m), **f64)
        self.obj = torch.empty(batch, **f64)
        self.out = torch.empty((2, batch), dtype=torch.int32, device=device)  # [iter; code]
        self.units_per_step = batch
        self.bytes_per_unit = qp_dense_algorithmic_bytes(n, m)
        self.ws = sfb.Workspace.for_dense(batch, n, m, self.prm)   # created once: no allocation inside a step

    def step(self, stream):
        P, q, A, l, u = self.dev
        self.sfb.solve_qp_batch_device_ws(self.B, self.n, self.m, P.data_ptr(), q.data_ptr(), A.data_ptr(),
                                          l.data_ptr(), u.data_ptr(), self.x.data_ptr(), self.y.data_ptr(),
                                          self.obj.data_ptr(), self.out[0].data_ptr(), self.out[1].data_ptr(),
                                          self.ws, self.prm, stream=stream.cuda_stream)

    def small_outputs(self):
        if not hasattr(self, "small"):
            self.small = torch.empty((self.B, 2), dtype=torch.int32, device=self.out.device)
        self.small.copy_(self.out.T)  # one row per QP: (iter, code)
        return self.small

    def extra(self):
        it = self.out[0].cpu().numpy().astype(np.int64)
        code = self.out[1].cpu().numpy()
        return {"iterations": {"mean": float(it.mean()), "p50": int(np.percentile(it, 50)),
                               "p99": int(np.percentile(it, 99)), "max": int(it.max())},
                "codes": np.bincount(code, minlength=7).tolist()}

    def roofline_alt(self, kern_ms):
        """The dense kernels are bound by dependent FP64 work, not by HBM (SURVEY.md section 8d caveat for cfg2):
        algorithmic flops of the batch -- k^3/3 for the factorisation, 2k^2 + 10m + 3n per ADMM iteration, six
        mat-vecs per stopping check, the polish factorisation and its five refinement solves for Optimal QPs --
        against the vector FP64 peak."""
        n, m = self.n, self.m
        k = n + m
        it = self.out[0].cpu().numpy().astype(np.float64)
        opt = (self.out[1].cpu().numpy() == 0)
        na = n + m / 2.0
        flops = (k ** 3 / 3.0 + it * (2 * k * k + 10 * m + 3 * n) + np.ceil(it / 25.0) * (4 * n * n + 8 * m * n)
                 + opt * (na ** 3 / 3.0 + 15 * na * na)).sum()
        ach = flops / (kern_ms * 1e-3)
        return {"bound": "fp64_valu", "achieved": ach / 1e12, "peak": FP64_VALU_PEAK / 1e12, "unit": "TFLOP/s",
                "frac": ach / FP64_VALU_PEAK, "flops_per_launch": float(flops),
                "note": "algorithmic flops (SURVEY 8d model) / kernel time; the triangular solves are chains of dependent "
                        "fp64 fmas, so the issue-slot utilisation (DESIGN.md: SIMD-cycles per QP-iteration) is the tighter view"}

    def cpu_baseline(self, cores, budget_s=15.0):
        """Oracle (CPU restatement of the reference ADMM) on a bounded sample of the same batch."""
        from oracle import loader as O
        op = O.default_params(eps_abs=1e-6, eps_rel=1e-6, polish=1, max_iter=10000, scaling=0)
        P, q, A, l, u = self.host
        s1 = min(self.B, 512)   # single core, sequential over the batch exactly like benchmarks/bench_types.hpp:93
        t0 = time.perf_counter()
        O.qp_dense_solve_batch(P[:s1], q[:s1], A[:s1], l[:s1], u[:s1], params=op, nthreads=1)
        single = {"value": s1 / (time.perf_counter() - t0), "cores": 1, "sample": "first %d QPs" % s1}
        probe = min(self.B, 64 * cores)
        t0 = time.perf_counter()
        O.qp_dense_solve_batch(P[:probe], q[:probe], A[:probe], l[:probe], u[:probe], params=op, nthreads=cores)
        rate = probe / (time.perf_counter() - t0)
        S = int(min(self.B, max(probe, rate * budget_s)))
        t0 = time.perf_counter()
        ref = O.qp_dense_solve_batch(P[:S], q[:S], A[:S], l[:S], u[:S], params=op, nthreads=cores)
        dt = time.perf_counter() - t0
        x = self.x[:S].cpu().numpy()
        it = self.out[0, :S].cpu().numpy().astype(np.uint32)
        code = self.out[1, :S].cpu().numpy()
        fin = np.isfinite(ref["x"]).all(axis=1)
        parity = {
            "sample": S,
            "code_mismatches": int((code != ref["code"]).sum()),
            "iter_mismatches": int((it != ref["iter"]).sum()),
            "max_abs_dx": float(np.abs(x - ref["x"])[fin].max(initial=0.0)),
        }
        return {"value": S / dt, "unit": "QP solves/s", "cores": cores, "kind": "port",
                "sample": "first %d QPs of rank 0's batch, oracle/qp_oracle.c (CPU restatement of the "
                          "reference ADMM, %d pthreads, static partition), %.1f s" % (S, cores, dt),
                "single_core": single}, parity


class MPCWorkload:
    """BASELINE.json configs[2] (the configuration the metric string is quoted on): a swarm of
    SE2xR3-type vehicles, nx=12, nu=2, K=50 (13 LGR intervals x 4 nodes, N=52 => n = m = 740 per QP),
    transcribed on the host by the C++ front (examples/models.cpp over include/smooth_feedback_amd/
    mpc.hpp, mirroring mpc.hpp:458-519 / ocp_to_qp.hpp), default MPCParams.qp, cold start.
    Agent b: t_b = 0.025 (b mod 400), x_b = xdes(t_b) (+) U(-0.5,0.5)^12."""

    def __init__(self, sfb, rank, device, batch=8192, variant=12, K=50):
        from examples import models_lib as M
        self.sfb, self.B = sfb, batch
        d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
        self.d, self.pat = d, (Pp, Pi, Pv, Ap, Aj)
        self.name = "mpc_qp_nx%d_nu%d_K%d_b%d_default_qp_params" % (d["Nx"], d["Nu"], K, batch)
        t0 = time.perf_counter()
        # host assembly threads: this rank's share of the CPUs the job may use (N ranks on one node share them)
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
        Av, l, u = M.mpc_assemble_batch(variant, K, batch, seed=1000003 * rank + 3, threads=max(1, host_cpus()[0] // max(1, local_world)))
        self.host_assembly_s = time.perf_counter() - t0
        # Structure probe, as the C++ front does (MPCSwarm::step): stored entries of A that are zero in a sample of the
        # batch are declared explicit zeros (ocp_to_qp writes dense Jacobian blocks) and left out of the analysis;
        # the kernel verifies the declaration for every agent (whole-pattern fallback otherwise).  SFB_BENCH_NO_PRUNE=1
        # analyses everything that is stored (A/B knob).
        keep = None if os.environ.get("SFB_BENCH_NO_PRUNE") == "1" else np.any(Av[:: max(1, batch // 64)] != 0.0, axis=0)
        self.plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K), keep=keep)
        self.prm = sfb.QPSolverParams()  # MPCParams.qp{} defaults (mpc.hpp:332)
        self.host = (np.tile(Pv, (batch, 1)), np.zeros((batch, d["n"])), Av, l, u)
        self.dev = [torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in self.host]
        f64 = dict(dtype=torch.float64, device=device)
        self.x = torch.empty((batch, d["n"]), **f64)
        self.y = torch.empty((batch, d["m"]), **f64)
        self.obj = torch.empty(batch, **f64)
        self.out = torch.empty((2, batch), dtype=torch.int32, device=device)
        self.ws = torch.empty((self.plan.workspace_bytes(batch) + 7) // 8, **f64)
        self.units_per_step = batch
        n, m = d["n"], d["m"]
        # values of P, q, A, l, u in; x, y, obj, iter, code out
        self.bytes_per_unit = 8 * (d["nnzP"] + n + d["nnzA"] + 2 * m) + 8 * (n + m) + 16
        self.ub = d["Nx"] * (d["N"] + 1)
        self.small = torch.empty((batch, 4), **f64)  # u0 (2), code, iter: what an MPC tick returns

    def step(self, stream):
        Px, q, Ax, l, u = self.dev
        self.plan.solve_batch_device(self.B, Px.data_ptr(), q.data_ptr(), Ax.data_ptr(), l.data_ptr(), u.data_ptr(),
                                     self.x.data_ptr(), self.y.data_ptr(), self.obj.data_ptr(),
                                     self.out[0].data_ptr(), self.out[1].data_ptr(), self.ws.data_ptr(), self.prm,
                                     stream=stream.cuda_stream)

    def small_outputs(self):
        self.small[:, 0:2] = self.x[:, self.ub:self.ub + 2]
        self.small[:, 2] = self.out[1]
        self.small[:, 3] = self.out[0]
        return self.small

    def pipelined(self, steps, streams=2):
        """Independent batches in flight on `streams` HIP streams (own outputs and workspace each): the end of a
        launch, when only the long-running agents are left and the chip is mostly idle, overlaps with the bulk of
        the next batch.  Reported next to `value` (which times one batch at a time); same inputs, same results."""
        Px, q, Ax, l, u = self.dev
        f64 = dict(dtype=torch.float64, device=Px.device)
        sets = [(self.x, self.y, self.obj, self.out, self.ws)]
        for _ in range(streams - 1):
            sets.append((torch.empty_like(self.x), torch.empty_like(self.y), torch.empty_like(self.obj),
                         torch.empty_like(self.out), torch.empty_like(self.ws)))
        ss = [torch.cuda.Stream() for _ in range(streams)]

        def launch(i):
            x, y, obj, out, ws = sets[i % streams]
            self.plan.solve_batch_device(self.B, Px.data_ptr(), q.data_ptr(), Ax.data_ptr(), l.data_ptr(), u.data_ptr(),
                                         x.data_ptr(), y.data_ptr(), obj.data_ptr(), out[0].data_ptr(), out[1].data_ptr(),
                                         ws.data_ptr(), self.prm, stream=ss[i % streams].cuda_stream)
        torch.cuda.synchronize()
        for i in range(streams):
            launch(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            launch(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        same = all(bool(torch.equal(sets[0][0], sx[0]) and torch.equal(sets[0][3], sx[3])) for sx in sets[1:])
        return {"streams": streams, "steps": steps, "value": self.B * steps / dt, "unit": "QP solves/s",
                "ms_per_step": dt / steps * 1e3, "results_identical_across_streams": same,
                "note": "independent batches overlapped on HIP streams; `value` above times one batch at a time"}

    def ordered_like_a_swarm_tick(self, steps=3):
        """NOT the headline: the same batch launched in descending order of the PREVIOUS solve's iteration counts -- what the
        device-resident swarm does from its second tick on (iteration counts change little from tick to tick; the results
        do not depend on the launch order).  Shows how much of a launch is the tail of long-running agents that a cold
        start cannot know in advance."""
        Px, q, Ax, l, u = self.dev
        order = torch.argsort(self.out[0], descending=True, stable=True).to(torch.int32)
        stream = torch.cuda.current_stream()
        x, y, out = torch.empty_like(self.x), torch.empty_like(self.y), torch.empty_like(self.out)
        def go():
            self.plan.solve_batch_device(self.B, Px.data_ptr(), q.data_ptr(), Ax.data_ptr(), l.data_ptr(), u.data_ptr(), x.data_ptr(),
                                         y.data_ptr(), self.obj.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), self.ws.data_ptr(),
                                         self.prm, stream=stream.cuda_stream, dorder=order.data_ptr())
        go(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(steps):
            go()
        e1.record(stream); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        return {"value": self.B / ms * 1e3, "unit": "QP solves/s", "ms_per_step": ms,
                "results_identical": bool(torch.equal(x, self.x) and torch.equal(out, self.out)),
                "note": "launch order = descending iteration count of the previous solve of the same batch (what MPCSwarmDevice does "
                        "between ticks); `value` above is the cold launch in natural order"}

    def closed_loop(self, ticks=8, reps=3):
        """What a user of MPC::operator() (mpc.hpp:458-519) sees, next to the device-resident `value`:
        end_to_end   -- the host-pointer entry point on the SAME batch: H2D of the QP data + solve + D2H of x, y, obj, iter, code;
        swarm_tick   -- MPCSwarmDeviceLin (include/smooth_feedback_amd/mpc_device.hpp): states up, linearisation + assembly +
                        warm-started solve on the GPU, u0 / code / iter down; closed loop of the bench's vehicle model at the
                        reference example's 25 ms control period (examples/mpc_asif_vehicle.cpp:58,161-169);
        single_agent -- ONE controller through the host entry point (the reference's own use), cold and warm."""
        from examples import models_lib as M
        Px, q, Av, l, u = self.host
        r = self.plan.solve_batch_host(Px, q, Av, l, u, self.prm)  # (first call: staging buffers, workspace)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r = self.plan.solve_batch_host(Px, q, Av, l, u, self.prm, out=r)  # (the caller's result buffers, as through the C entry)
            ts.append(time.perf_counter() - t0)
        it = self.out[0].cpu().numpy().astype(np.uint32)
        moved = sum(a.nbytes for a in self.host) + r.primal.nbytes + r.dual.nbytes + r.objective.nbytes + r.iter.nbytes + r.code.nbytes
        e2e = {"value": self.B / float(np.median(ts)), "unit": "QP solves/s", "ms_per_step": 1e3 * float(np.median(ts)), "reps": reps,
               "bytes_over_pcie_per_step": int(moved), "results_identical_to_device_resident": bool(np.array_equal(r.iter, it)),
               "entry": "sfb_sparse_qp_solve_batch_host (pageable numpy buffers in, the caller's numpy buffers out: nothing is allocated in the timed call)"}
        variant, K = 12, 50
        sw = M.mpc_swarm_devlin_step(variant, K, self.B, ticks, seed=1, want_records=False)
        warm = sw["seconds"][2:]  # tick 0: cold start + structure probe + plan; tick 1: first warm start
        tick = {"ms_per_tick": 1e3 * float(np.mean(warm)), "ms_per_tick_all": [round(1e3 * float(s), 3) for s in sw["seconds"]],
                "agents": self.B, "ticks": ticks, "warm_ticks_averaged": len(warm), "control_period_ms": 25.0,
                "mean_iterations_last_tick": float(sw["iter"].mean()), "optimal_fraction_last_tick": float(np.mean(sw["code"] == 0)),
                "value": self.B / float(np.mean(warm)), "unit": "MPC ticks/s",
                "path": "MPCSwarmDeviceLin::step: states H2D, device linearisation + assembly + warm-started sparse solve, u0/code/iter D2H"}
        # the same closed loop with half the agents: what fits the reference example's 25 ms control period on one GPU
        sw2 = M.mpc_swarm_devlin_step(variant, K, self.B // 2, 6, seed=1, want_records=False)
        tick["half_swarm"] = {"agents": self.B // 2, "ms_per_tick": 1e3 * float(np.mean(sw2["seconds"][2:])),
                              "mean_iterations_last_tick": float(sw2["iter"].mean())}
        P1, q1 = Px[:1], q[:1]
        cold, warm1 = [], []
        r1 = self.plan.solve_batch_host(P1, q1, Av[:1], l[:1], u[:1], self.prm)
        for _ in range(10):
            t0 = time.perf_counter()
            r1 = self.plan.solve_batch_host(P1, q1, Av[:1], l[:1], u[:1], self.prm)
            cold.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            r2 = self.plan.solve_batch_host(P1, q1, Av[:1], l[:1], u[:1], self.prm, warm_x=r1.primal, warm_y=r1.dual)
            warm1.append(time.perf_counter() - t0)
        single = {"cold_ms": 1e3 * float(np.median(cold)), "cold_iterations": int(r1.iter[0]), "warm_ms": 1e3 * float(np.median(warm1)),
                  "warm_iterations": int(r2.iter[0]), "warm_note": "the same QP warm-started from its own solution: scaling + factorisation + polish + PCIe, "
                  "the floor of a warm tick", "entry": "sfb_sparse_qp_solve_batch_host, batch 1 (n = m = %d)" % self.d["n"]}
        return {"end_to_end": e2e, "swarm_tick": tick, "single_agent": single}

    def phases_ms(self):
        """The reference's verbose summary (qp_solver.hpp:550-565: Matrix filling / Factorization / Iteration / Polish) as DATA for
        the headline batch: sfb_sparse_qp_solve_batch_phases = the same solves through the TRACE instance of the kernel (one wave
        per agent, the whole batch resident in turn), per-agent device wall-clock per phase; batch means in ms.  NOT the timed
        launch (that one is time-sliced and ordered): what an agent's solve consists of when the chip is full."""
        Px, q, Ax, l, u = self.dev
        ph = torch.zeros((self.B, 6), dtype=torch.float64, device=Px.device)
        x2, y2, out2 = torch.empty_like(self.x), torch.empty_like(self.y), torch.empty_like(self.out)
        self.plan.solve_batch_device_phases(self.B, Px.data_ptr(), q.data_ptr(), Ax.data_ptr(), l.data_ptr(), u.data_ptr(),
                                            x2.data_ptr(), y2.data_ptr(), 0, out2[0].data_ptr(), out2[1].data_ptr(),
                                            self.ws.data_ptr(), ph.data_ptr(), self.prm)
        torch.cuda.synchronize()
        m = (ph.mean(dim=0) * 1e-3).cpu().numpy()
        names = ("scaling_and_precheck", "matrix_filling", "factorization", "iteration", "polish", "unscale_and_report")
        return {**{k: float(v) for k, v in zip(names, m)}, "per_agent_total": float(m.sum()),
                "results_identical_to_the_timed_launch": bool(torch.equal(x2, self.x) and torch.equal(out2, self.out)),
                "unit": "ms per agent (batch mean, device wall clock)", "entry": "sfb_sparse_qp_solve_batch_phases (TRACE instance, one wave per agent)"}

    def extra(self):
        it = self.out[0].cpu().numpy().astype(np.int64)
        code = self.out[1].cpu().numpy()
        return {"iterations": {"mean": float(it.mean()), "p50": int(np.percentile(it, 50)),
                               "p99": int(np.percentile(it, 99)), "max": int(it.max())},
                "codes": np.bincount(code, minlength=7).tolist(), "nnzL": int(self.plan.nnzL),
                "nnzA_stored": int(self.plan.nnzA), "nnzA_analysed": int(self.plan.nnzA_kept),
                "nnzL_whole_pattern_same_order": int(self.plan.nnzL_fallback),
                "factor_stream_bytes_per_iteration_per_qp": int(16 * self.plan.nnzL),
                "host_assembly_s": self.host_assembly_s}

    def cpu_baseline(self, cores, budget_s=20.0):
        """The oracle on the WHOLE stored pattern (explicit zeros included, as the reference's SimplicialLDLT
        factorises them) with the plan's elimination and summation orders: baseline and parity check in one."""
        from oracle import loader as O
        Pp, Pi, Pv, Ap, Aj = self.pat
        Px, q, Av, l, u = self.host
        op = O.default_params()  # max_iter unset, like the device run (device cap 2e7 is never reached here)
        s1 = min(self.B, 24)     # single core, one agent after the other like a loop over MPC::operator()
        t0 = time.perf_counter()
        O.qp_sparse_solve_batch(Pp, Pi, Px[:s1], q[:s1], Ap, Aj, Av[:s1], l[:s1], u[:s1], perm=self.plan.perm,
                                forder=self.plan.factor_order(), params=op, nthreads=1)
        single = {"value": s1 / (time.perf_counter() - t0), "cores": 1, "sample": "first %d agents" % s1}
        S = int(min(self.B, max(cores, 2 * cores)))
        t0 = time.perf_counter()
        ref = O.qp_sparse_solve_batch(Pp, Pi, Px[:S], q[:S], Ap, Aj, Av[:S], l[:S], u[:S], perm=self.plan.perm, forder=self.plan.factor_order(),
                                      params=op, nthreads=cores)
        dt = time.perf_counter() - t0
        if dt < budget_s / 4 and S < self.B:  # grow the sample towards the budget
            S2 = int(min(self.B, S * max(2, int(budget_s / 2 / max(dt, 1e-3)))))
            t0 = time.perf_counter()
            ref = O.qp_sparse_solve_batch(Pp, Pi, Px[:S2], q[:S2], Ap, Aj, Av[:S2], l[:S2], u[:S2],
                                          perm=self.plan.perm, forder=self.plan.factor_order(), params=op, nthreads=cores)
            dt, S = time.perf_counter() - t0, S2
        x = self.x[:S].cpu().numpy()
        it = self.out[0, :S].cpu().numpy().astype(np.uint32)
        code = self.out[1, :S].cpu().numpy()
        du = np.abs(x[:, self.ub:self.ub + 2] - ref["x"][:, self.ub:self.ub + 2])
        parity = {"sample": S, "code_mismatches": int((code != ref["code"]).sum()),
                  "iter_mismatches": int((it != ref["iter"]).sum()), "max_abs_du0": float(du.max()),
                  "max_abs_dx": float(np.abs(x - ref["x"]).max()),
                  "scope": "kernel vs the CPU restatement on the SAME assembled QPs (whole stored pattern, the plan's elimination order); "
                           "the MPC transcription that produced them (mpc.hpp / lie.hpp restated from memory of pettni/smooth) is pinned only "
                           "by the reference's structural tests: u0 parity with the real reference is unpinned (DESIGN.md section 2)"}
        return {"value": S / dt, "unit": "QP solves/s", "cores": cores, "kind": "port",
                "sample": "first %d agents of rank 0's batch, oracle/qp_sparse_oracle.c (CPU restatement of the "
                          "reference's sparse ADMM path on the whole stored pattern, same elimination order, %d pthreads), %.1f s"
                          % (S, cores, dt), "single_core": single}, parity


class EKFWorkload:
    """BASELINE.json configs[4]: 1 048 576 independent SE2xR3 filters (Dof 6, Ny 3), one fused
    predict (Euler substep, ekf.hpp:84-96) + update (ekf.hpp:119-138) per item per launch, per-item
    A, Q, H, R, r (P = I + GG'/6, Q = 0.1 I, R = 0.1 I3, dt = 0.025,
    A = -ad(f) + df/dx of the vehicle model of examples/mpc_asif_vehicle.cpp:42-50 at a random body velocity).
    Consecutive steps filter the evolving covariance, as consecutive ticks of a filter do."""

    def __init__(self, sfb, rank, device, batch=1 << 20, dof=6, ny=3):
        self.sfb, self.B, self.n, self.m = sfb, batch, dof, ny
        self.name = "ekf_predict_update_dof%d_ny%d_b%d" % (dof, ny, batch)
        rng = np.random.default_rng(7 + rank)
        n, m, B = dof, ny, batch
        G = rng.uniform(-1, 1, (B, n, n))
        flat = lambda M: np.ascontiguousarray(M.transpose(0, 2, 1).reshape(M.shape[0], -1))
        if n == 6:
            v = rng.uniform(-1, 1, (B, 3))  # body velocity (vx, vy, omega)
            A = np.zeros((B, 6, 6))
            A[:, 0, 1], A[:, 0, 2] = v[:, 2], -v[:, 1]    # -ad_se2(v)
            A[:, 1, 0], A[:, 1, 2] = -v[:, 2], v[:, 0]
            A[:, 0, 3] = A[:, 1, 4] = A[:, 2, 5] = 1.0    # d(pose rate)/d(velocity)
            A[:, 3, 3], A[:, 5, 5] = -0.2, -0.4           # velocity damping
        else:
            A = rng.uniform(-1, 1, (B, n, n))
        self.host = dict(P=flat(np.eye(n)[None] + G @ G.transpose(0, 2, 1) / n), A=flat(A),
                         Q=np.tile((0.1 * np.eye(n)).flatten(), (B, 1)), H=flat(rng.uniform(-1, 1, (B, m, n))),
                         R=np.tile((0.1 * np.eye(m)).flatten(), (B, 1)), r=rng.uniform(-1, 1, (B, m)),
                         dt=np.full(B, 0.025))
        self.dev = {k: torch.from_numpy(v).to(device) for k, v in self.host.items()}
        self.delta = torch.empty((B, n), dtype=torch.float64, device=device)
        self.units_per_step = B
        # in: P, A, Q (36 each), H 18, R 9, r 3, dt 1; out: P 36, delta 6
        self.bytes_per_unit = 8 * (3 * n * n + m * n + m * m + m + 1) + 8 * (n * n + n)
        self.small = self.delta
        self.launches = 0

    def step(self, stream):
        d = self.dev
        self.sfb.ekf_predict_update_batch_device(self.B, self.n, self.m, d["A"].data_ptr(), d["Q"].data_ptr(), 0,
                                                 d["dt"].data_ptr(), 0, d["H"].data_ptr(), d["R"].data_ptr(), 0,
                                                 d["r"].data_ptr(), d["P"].data_ptr(), self.delta.data_ptr(),
                                                 stream=stream.cuda_stream)
        self.launches += 1

    def small_outputs(self):
        return self.delta

    def cpu_baseline(self, cores, budget_s=15.0):
        from oracle import loader as O
        h = self.host
        S = int(min(self.B, max(1000, 2.0e6 * budget_s / max(1, self.launches))))  # ~2.5 M steps/s on one core
        P = h["P"][:S].copy()
        t0 = time.perf_counter()
        for _ in range(self.launches):  # the same ticks the GPU ran, on the first S filters
            Pp = O.ekf_predict_batch(h["A"][:S], h["Q"][:S], h["dt"][:S], P)
            P, dref, _ = O.ekf_update_batch(h["H"][:S], h["R"][:S], h["r"][:S], Pp, self.n)
        dt = time.perf_counter() - t0
        Pg = self.dev["P"][:S].cpu().numpy()
        delta = self.delta[:S].cpu().numpy()
        parity = {"sample": S, "ticks": self.launches, "P_bit_identical": bool(np.array_equal(Pg, P)),
                  "delta_bit_identical": bool(np.array_equal(delta, dref)),
                  "max_abs_dP": float(np.abs(Pg - P).max()), "max_abs_ddelta": float(np.abs(delta - dref).max()),
                  "max_abs_P": float(np.abs(Pg).max())}
        single = {"value": S * self.launches / dt, "cores": 1, "sample": "first %d filters x %d ticks, 1 thread, %.1f s" % (S, self.launches, dt)}
        if cores <= 1:
            return {"value": single["value"], "unit": "EKF steps/s", "cores": 1, "kind": "port",
                    "sample": "first %d filters of rank 0's batch x %d ticks, oracle/ekf_oracle.c (scalar C restatement of "
                              "ekf.hpp:84-96,119-138), 1 thread, %.1f s" % (S, self.launches, dt), "single_core": single}, parity
        # all cores: static partition of a `cores` times larger sample, one thread per part
        # (the oracle's C loops run without the GIL)
        from concurrent.futures import ThreadPoolExecutor
        SA = int(min(self.B, S * cores))
        parts = [(i * SA // cores, (i + 1) * SA // cores) for i in range(cores)]

        def ticks_of(part):
            a, b = part
            Pc = h["P"][a:b].copy()
            for _ in range(self.launches):
                Pq = O.ekf_predict_batch(h["A"][a:b], h["Q"][a:b], h["dt"][a:b], Pc)
                Pc, _, _ = O.ekf_update_batch(h["H"][a:b], h["R"][a:b], h["r"][a:b], Pq, self.n)
            return Pc
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            res = list(ex.map(ticks_of, parts))
        dta = time.perf_counter() - t0
        parity["all_cores_sample"] = SA
        parity["all_cores_P_bit_identical"] = bool(np.array_equal(self.dev["P"][:SA].cpu().numpy(), np.concatenate(res)))
        return {"value": SA * self.launches / dta, "unit": "EKF steps/s", "cores": cores, "kind": "port",
                "sample": "first %d filters of rank 0's batch x %d ticks, oracle/ekf_oracle.c (scalar C restatement of "
                          "ekf.hpp:84-96,119-138), static partition over %d threads, %.1f s" % (SA, self.launches, cores, dta),
                "single_core": single}, parity


WORKLOADS = {"mpc": MPCWorkload, "qp_dense": DenseQPWorkload, "ekf": EKFWorkload}


# FETCH_SIZE on gfx950 reports 1/2 of the bytes of 16-byte-per-lane coalesced reads (MI355X_MICROARCH.md,
# HBM section); for the 8-byte-per-lane reads of the dense kernel the factor was calibrated on its exactly
# known read volume (profiles/README.md).  WRITE_SIZE matched the known write volume of the EKF kernel.
# The sparse kernel mixes 16-byte streams (factor, vectors) with 8-byte gathers (factorisation, checks): x2 is exact
# for the former (~85 % of its reads) and over-counts the latter, so its traffic figure is an upper bound (~ +10 %).
FETCH_CORRECTION = {"mpc": 2.0, "ekf": 2.0, "qp_dense": 1.0 / 0.58}
PROFILE_TAG = "r6"
FP64_VALU_PEAK = 78.6e12  # MI355X vector FP64 (half the 157.3 TFLOP/s FP32 vector rate of MI355X_MICROARCH.md)


def source_hash():
    """sha256 over the kernel sources: a committed PMC summary is only quoted for the build it was measured on."""
    import hashlib
    root = os.path.dirname(os.path.abspath(__file__))
    h = hashlib.sha256()
    d = os.path.join(root, "smooth_feedback_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode()); h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(workload, workload_name):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/<tag>_<workload>/summary.json,
    written by scripts/profile_one.sh + scripts/summarize_profiles.py for this very bench command).  None when
    there is no summary for this workload configuration OR the kernel sources have changed since it was taken."""
    root = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(root, "profiles", "%s_%s" % (PROFILE_TAG, workload), "summary.json")
    try:
        with open(path) as f:
            s = json.load(f)
        if s["bench_line_under_profiler"]["config"]["workload"] != workload_name:
            return None, "no PMC summary for this workload configuration"
        if s.get("source_hash") != source_hash():
            return None, "PMC summary %s is of another build (source hash differs): traffic not quoted" % os.path.relpath(path, root)
        rd = s["FETCH_SIZE"]["mean_per_dispatch_KB"] * 1024.0 * FETCH_CORRECTION[workload]
        wr = s["WRITE_SIZE"]["mean_per_dispatch_KB"] * 1024.0
        return rd + wr, "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH x%.2f, %s" % (
            FETCH_CORRECTION[workload], os.path.relpath(path, root))
    except (OSError, KeyError, ValueError):
        return None, "no PMC summary"


KERNEL_NAME = {"mpc": "qp_sparse_kernel (launch in predicted order: first launch to the first check, rank kernel, LAT loop launch with the polishers next to it, finish launch)", "qp_dense": "qp_dense4_iterate_kernel (+ setup and finish kernels of the same launch)",
               "ekf": "ekf_fused_persistent_kernel (persistent waves, the next tile's covariances requested straight into LDS)"}


def roofline_of(workload, wl, kern_ms, with_traffic):
    achieved = wl.units_per_step * wl.bytes_per_unit / (kern_ms * 1e-3)
    r = {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK_BYTES_PER_S / 1e9, "unit": "GB/s",
         "frac": achieved / HBM_PEAK_BYTES_PER_S, "traffic": None, "kernel": KERNEL_NAME[workload], "kernel_ms": kern_ms,
         "algorithmic_bytes_per_unit": wl.bytes_per_unit}
    if workload != "ekf":
        r["note"] = ("achieved = SURVEY 8(d) algorithmic I/O bytes / kernel time; the ADMM iterations re-stream the LDL' "
                     "factor from HBM (sparse kernel) or are FP64-issue bound (dense kernel), so the I/O-only fraction is "
                     "small by construction -- see DESIGN.md")
    if with_traffic:
        r["traffic"], r["traffic_source"] = pmc_traffic(workload, wl.name)
        if r["traffic"]:  # what the kernel actually moves through HBM, as a rate and a fraction of peak
            r["traffic_GBps"] = r["traffic"] / (kern_ms * 1e-3) / 1e9
            r["traffic_frac"] = r["traffic"] / (kern_ms * 1e-3) / HBM_PEAK_BYTES_PER_S
            if workload == "mpc":
                r["traffic_note"] = ("FETCH_SIZE / WRITE_SIZE count the L2's traffic with the fabric: more than half of it (the loop "
                                     "launch's factor streams, profiles/%s_mpc/summary.json: qp_sparse_kernel<true>) is served by the "
                                     "256 MB Infinity Cache, not by HBM -- an upper bound of the HBM bytes") % PROFILE_TAG
    return r


def secondary_line(sfb, workload, device, steps=3):
    """One of the other BASELINE configurations, measured the same way as the headline (HIP events on the launch
    stream, inputs resident in HBM) with a bounded parity / CPU sample: reported under `secondary`."""
    wl = WORKLOADS[workload](sfb, 0, device)
    stream = torch.cuda.current_stream()
    wl.step(stream)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for k in range(steps):
        ev[k][0].record(stream)
        wl.step(stream)
        ev[k][1].record(stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    rec = {"value": wl.units_per_step * steps / elapsed, "unit": "EKF steps/s" if workload == "ekf" else "QP solves/s",
           "steps": steps, "ms_per_step": elapsed / steps * 1e3, "dtype": "f64",
           "config": {"workload": wl.name, "per_gpu_batch": wl.B},
           "roofline": roofline_of(workload, wl, kern_ms, True)}
    if hasattr(wl, "extra"):
        rec["workload_stats"] = wl.extra()
    if hasattr(wl, "roofline_alt"):
        rec["roofline_alt"] = wl.roofline_alt(kern_ms)
    cores, cpuinfo = host_cpus()
    rec["cpu_baseline"], rec["parity_vs_oracle"] = wl.cpu_baseline(cores, budget_s=4.0)
    rec["cpu_baseline"]["host"] = cpuinfo
    return rec


DENSE_SIZES = [(10, 20), (16, 32), (20, 40), (32, 32), (32, 64), (40, 60), (64, 64)]


def dense_sizes_table(sfb, device, cores):
    """The north star's size class (dense QPs with n <= 64) beyond the one size of BASELINE configs[1]: random QPs of
    benchmarks/bench_types.hpp:19-41 at several (n, m), under the reference benchmark's parameters (bench.cpp:148-153)
    and under the library defaults (max_iter 10 000 in both).  Device-resident, HIP-event timed.  QP-iterations/s is
    quoted next to QP/s because iteration counts differ by orders of magnitude between sizes and parameter sets (and
    a single QP that runs into max_iter on its own wave sets the time of a batch of fast ones).  Kernels: n + m <= 32
    four QPs per wave (qp_dense4); 32 < n + m <= 128 one per wave, packed factor in LDS, iterate in registers, block
    sweeps, as setup / time-sliced loop / finish launches over per-QP records in the caller's workspace (qp_dense_mid).
    Parity: the first 2 048 QPs of every batch against the dense oracle."""
    from oracle import loader as O
    rows = []
    for n, m in DENSE_SIZES:
        k = n + m
        B = 65536  # the batch of BASELINE configs[1]: one QP that runs into max_iter sets the time of a small batch
        P, q, A, l, u = sfb.random_qp_batch(5, B, m, n, 1.0)
        d = [torch.from_numpy(a).to(device) for a in (P, q, A, l, u)]
        f64 = dict(dtype=torch.float64, device=device)
        x, y, obj = torch.empty((B, n), **f64), torch.empty((B, m), **f64), torch.empty(B, **f64)
        out = torch.empty((2, B), dtype=torch.int32, device=device)
        stream = torch.cuda.current_stream()
        ws = None
        for name, prm, okw in (("reference_benchmark", sfb.QPSolverParams(eps_abs=1e-6, eps_rel=1e-6, polish=True, max_iter=10000, scaling=False),
                                dict(eps_abs=1e-6, eps_rel=1e-6, polish=1, max_iter=10000, scaling=0)),
                               ("library_defaults", sfb.QPSolverParams(max_iter=10000), dict(max_iter=10000))):
            need = sfb.Workspace.dense_bytes(B, n, m, prm)   # the caller's workspace: no allocation inside the timed call
            if ws is None or ws.nbytes < need:
                ws = sfb.Workspace(need)  # (a re-created one frees the old through __del__)
            def go():
                sfb.solve_qp_batch_device_ws(B, n, m, *[a.data_ptr() for a in d], x.data_ptr(), y.data_ptr(), obj.data_ptr(),
                                             out[0].data_ptr(), out[1].data_ptr(), ws, prm, stream=stream.cuda_stream)
            go(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream); go(); e1.record(stream); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            it = out[0].cpu().numpy().astype(np.int64); code = out[1].cpu().numpy()
            S = min(B, 2048)
            ref = O.qp_dense_solve_batch(P[:S], q[:S], A[:S], l[:S], u[:S], params=O.default_params(**okw), nthreads=cores)
            fin = np.isfinite(ref["x"]).all(axis=1)
            opt = code == 0
            na = n + m / 2.0
            flops = (k ** 3 / 3.0 + it * (2 * k * k + 10 * m + 3 * n) + np.ceil(it / 25.0) * (4 * n * n + 8 * m * n)
                     + opt * (na ** 3 / 3.0 + 15 * na * na)).sum()
            rows.append({"n": n, "m": m, "params": name, "batch": B, "ms": ms, "qp_per_s": B / ms * 1e3,
                         "qp_iterations_per_s": float(it.sum()) / ms * 1e3, "iterations": {"mean": float(it.mean()), "max": int(it.max())},
                         "codes": np.bincount(code, minlength=7).tolist(),
                         "roofline_alt": {"bound": "fp64_valu", "achieved": flops / (ms * 1e-3) / 1e12, "peak": FP64_VALU_PEAK / 1e12,
                                          "unit": "TFLOP/s", "frac": flops / (ms * 1e-3) / FP64_VALU_PEAK},
                         "parity_vs_oracle": {"sample": S, "code_mismatches": int((code[:S] != ref["code"]).sum()),
                                              "iter_mismatches": int((it[:S] != ref["iter"]).sum()),
                                              "max_abs_dx": float(np.abs(x[:S].cpu().numpy() - ref["x"])[fin].max(initial=0.0))}})
        del d, x, y, obj, out, ws
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="mpc", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the multi-stream throughput measurement (mpc)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads reported next to the headline")
    ap.add_argument("--no-closed-loop", action="store_true", help="skip the end-to-end / swarm-tick / single-agent figures (mpc)")
    ap.add_argument("--debug-knob", action="append", default=[], metavar="NAME=VALUE",
                    help="a debug knob of the library (sfb_debug_set, csrc/knobs.h) for this run: measurements and profiling only; "
                         "recorded in the line as `debug_knobs`")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (no CPU fallback)")
    # SFB_BENCH_SHARE_DEVICE=1 (test hook for 1-GPU boxes): all ranks use cuda:0 and rendezvous over gloo
    share = os.environ.get("SFB_BENCH_SHARE_DEVICE") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    # SFB_BENCH_FORCE_COLLECTIVES=1 (test hook): a world of ONE rank still initialises the process group (RCCL unless the device
    # is shared) and issues every collective of the N > 1 path -- the gather of the small outputs, the checksum all_reduce, the
    # barrier, the max-over-ranks reduction -- so that the first 8-GPU run is not the first time RCCL sees these dtypes / shapes
    collectives = world > 1 or os.environ.get("SFB_BENCH_FORCE_COLLECTIVES") == "1"
    if collectives:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)  # RCCL over xGMI

    import smooth_feedback_amd as sfb
    debug_knobs = sfb.debug_set_from(",".join(args.debug_knob)) if args.debug_knob else {}

    # Sharding (smooth_feedback_amd/sharding.py, the module the gloo tests cover): the job's batch is `world` times the
    # per-GPU batch, rank r owns the contiguous range shard_range(total, r, world) of it -- weak scaling, so every range
    # has the per-GPU size -- and the only exchange is gather_small_outputs of the per-item (u0, code, iter) rows.
    from smooth_feedback_amd.sharding import gather_small_outputs, shard_range
    per_gpu = args.batch if args.batch is not None else WORKLOADS[args.workload].__init__.__defaults__[0]
    total = per_gpu * world
    lo, hi = shard_range(total, rank, world)
    wl = WORKLOADS[args.workload](sfb, rank, device, batch=hi - lo)
    stream = torch.cuda.current_stream()
    gathered = None

    def one_step():
        nonlocal gathered
        if hasattr(wl, "pre_step"):
            wl.pre_step()
        wl.step(stream)
        if collectives:  # the only exchange on this path: final gather of the small outputs (RCCL/xGMI)
            gathered = gather_small_outputs(wl.small_outputs(), total, force=True)

    def barrier():
        if collectives:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        if hasattr(wl, "pre_step"):
            wl.pre_step()
        ev[k][0].record(stream)   # HIP events on the stream the kernel is launched on
        wl.step(stream)
        ev[k][1].record(stream)
        if collectives:
            gathered = gather_small_outputs(wl.small_outputs(), total, force=True)
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    gather_check = None
    if collectives:
        # What came out of the gather, checked against an independent path: every rank puts an exact (integer,
        # wrap-around) checksum of ITS rows into its slot of a vector that is summed over the ranks with an all_reduce;
        # the rows received from rank r must reproduce rank r's checksum.  Shards come from different seeds, so a
        # peer's rows must also differ from this rank's own.
        def checksum(tns):
            return tns.contiguous().view(torch.int32).to(torch.int64).sum()
        own = wl.small_outputs()
        cs = torch.zeros(world, dtype=torch.int64, device=device)
        cs[rank] = checksum(own)
        dist.all_reduce(cs, op=dist.ReduceOp.SUM)
        peers = [r for r in range(world) if r != rank]
        rows = {r: gathered[slice(*shard_range(total, r, world))] for r in range(world)}  # rank r's rows of the gathered tensor
        gather_check = {"own_rows_intact": bool(torch.equal(rows[rank], own)),
                        "peer_rows_received": all(bool(checksum(rows[r]) == cs[r]) for r in peers),
                        "peer_rows_differ_from_own": all(not bool(torch.equal(rows[r], own)) for r in peers),
                        "gathered_rows": int(gathered.shape[0]), "ranks_checked": world}
        if world == 1:  # (forced collectives) say which backend carried them and what went through it
            gather_check.update({"backend": dist.get_backend(), "gathered_dtype": str(gathered.dtype).replace("torch.", ""),
                                 "checksum_via_all_reduce_matches": bool(cs[0] == checksum(own))})

    el = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if collectives:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    if rank == 0:
        units = wl.units_per_step * world * args.steps
        value = units / elapsed
        rec = {
            "metric": "EKF predict+update steps/sec" if args.workload == "ekf" else "QP solves/sec",
            "value": value,
            "unit": "EKF steps/s" if args.workload == "ekf" else "QP solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": wl.name, "per_gpu_batch": wl.B, "global_batch": wl.B * world,
                       "parallelism": "batch-sharded x%d, one gather of the small per-item outputs (u0 / code / iter)" % world},
            "roofline": roofline_of(args.workload, wl, kern_ms, world == 1),
        }
        if debug_knobs:  # NOT a production line: launch shapes / engines were steered through sfb_debug_set
            rec["debug_knobs"] = debug_knobs
        if gather_check is not None:
            rec["gather_check"] = gather_check
        if hasattr(wl, "extra"):
            rec["workload_stats"] = wl.extra()
            ws = rec["workload_stats"]
            if "factor_stream_bytes_per_iteration_per_qp" in ws:
                eff = ws["factor_stream_bytes_per_iteration_per_qp"] * ws["iterations"]["mean"] * wl.units_per_step
                rec["roofline"]["factor_stream_GBps"] = eff / (kern_ms * 1e-3) / 1e9
        if hasattr(wl, "roofline_alt"):
            rec["roofline_alt"] = wl.roofline_alt(kern_ms)
        if world == 1 and hasattr(wl, "pipelined") and not args.no_pipelined:
            rec["ordered_like_a_swarm_tick"] = wl.ordered_like_a_swarm_tick()
            rec["pipelined"] = wl.pipelined(max(4, 2 * args.steps))
            rec["phases_ms"] = wl.phases_ms()
            if args.batch is None and not args.no_closed_loop:
                rec["closed_loop"] = wl.closed_loop()
        if not args.no_cpu_baseline and world == 1:
            cores, cpuinfo = host_cpus()
            rec["cpu_baseline"], rec["parity_vs_oracle"] = wl.cpu_baseline(cores)
            # `cores` = threads the oracle was run with = CPUs this process may use (affinity mask, cgroup quota), which
            # can be far fewer than the machine's logical CPUs; single_core next to it shows the scaling that was real
            rec["cpu_baseline"]["host"] = cpuinfo
            sc = rec["cpu_baseline"].get("single_core")
            if sc and sc.get("value"):
                rec["cpu_baseline"]["speedup_over_single_core"] = rec["cpu_baseline"]["value"] / sc["value"]
        if world == 1 and not args.no_secondary and not args.no_cpu_baseline and args.batch is None:
            del wl  # free the headline workload's device memory first
            torch.cuda.empty_cache()
            rec["secondary"] = {w: secondary_line(sfb, w, device) for w in sorted(WORKLOADS) if w != args.workload}
            rec["secondary"]["qp_dense_sizes"] = dense_sizes_table(sfb, device, host_cpus()[0])
        print(json.dumps(rec), flush=True)
    if collectives:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
