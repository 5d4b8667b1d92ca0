"""Device-side MPC assembly (sfb_mpc_assemble_batch) and the device-resident swarm (sfb_mpc_swarm) on the GPU:
A, l, u bit-identical to the host transcription; swarm ticks bit-identical to host assembly + batched solve,
cold and warm-started.  Needs an MI355X."""
import numpy as np
import pytest

from examples import models_lib as M

pytestmark = pytest.mark.gpu


def _assemble_on_device(L, rec, shared=None):
    import torch
    B = rec.shape[0]
    d_rec = torch.from_numpy(rec).cuda()
    d_sh = torch.from_numpy(shared).cuda() if shared is not None else None
    dA = torch.full((B, L.nnzA), np.nan, dtype=torch.float64, device="cuda")
    dl = torch.full((B, L.m), np.nan, dtype=torch.float64, device="cuda")
    du = torch.full((B, L.m), np.nan, dtype=torch.float64, device="cuda")
    L.assemble_batch_device(B, d_rec.data_ptr(), dA.data_ptr(), dl.data_ptr(), du.data_ptr(),
                            d_sh.data_ptr() if d_sh is not None else 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return dA.cpu().numpy(), dl.cpu().numpy(), du.cpu().numpy()


@pytest.mark.parametrize("variant,K,batch", [(6, 10, 7), (6, 50, 33), (12, 50, 300)])
def test_device_assembly_is_bit_identical_to_the_host_transcription(sfb, variant, K, batch):
    Av, l, u = M.mpc_assemble_batch(variant, K, batch, seed=9)
    L, rec = M.mpc_records(variant, K, batch, seed=9)
    A2, l2, u2 = _assemble_on_device(L, rec)
    assert np.array_equal(A2, Av) and np.array_equal(l2, l) and np.array_equal(u2, u)
    # time-invariant linearisation: Jacobians read from ONE shared record
    own, shared = L.split_shared(rec)
    A3, l3, u3 = _assemble_on_device(L, own, shared)
    assert np.array_equal(A3, Av) and np.array_equal(l3, l) and np.array_equal(u3, u)


def test_swarm_tick_equals_host_assembly_plus_solve(sfb):
    """One tick of sfb_mpc_swarm from records == SparseQPPlan.solve_batch_host on the host transcription,
    bit for bit, cold; then a second tick warm-started from what the swarm kept on the device."""
    variant, K, B = 12, 50, 96
    d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
    plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K))
    prm = sfb.QPSolverParams(max_iter=4000)
    Px = np.tile(Pv, (B, 1)); q = np.zeros((B, d["n"]))
    L, rec = M.mpc_records(variant, K, B, seed=21)
    Av, l, u = M.mpc_assemble_batch(variant, K, B, seed=21)
    swarm = sfb.MPCSwarm(plan, L, Pv, np.zeros(d["n"]), B)
    du0, code, it, x, y = swarm.step_host(rec, prm, full=True)
    r = plan.solve_batch_host(Px, q, Av, l, u, prm)
    ub = d["Nx"] * (d["N"] + 1)
    assert np.array_equal(code, r.code) and np.array_equal(it, r.iter)
    assert np.array_equal(x, r.primal) and np.array_equal(y, r.dual) and np.array_equal(du0, r.primal[:, ub:ub + 2])
    assert (code == 0).all()
    # tick 2: other states, warm start = solutions of tick 1 (all storable)
    L, rec2 = M.mpc_records(variant, K, B, seed=22)
    Av2, l2, u2 = M.mpc_assemble_batch(variant, K, B, seed=22)
    du0b, codeb, itb = swarm.step_host(rec2, prm)
    r2 = plan.solve_batch_host(Px, q, Av2, l2, u2, prm, warm_x=r.primal, warm_y=r.dual)
    assert np.array_equal(codeb, r2.code) and np.array_equal(itb, r2.iter) and np.array_equal(du0b, r2.primal[:, ub:ub + 2])
    # an agent whose solve is not storable keeps its OLDER warm start (mpc.hpp:510-516): make tick 3 infeasible for
    # agent 0 (c = +inf puts an upper bound at -inf: PrimalInfeasible, qp_solver.hpp:361-374), then tick 4 must start
    # it from tick 2's solution
    rec3 = rec2.copy()
    N, nx, nu, ncr = L.N, L.nx, L.nu, L.ncr
    o_c = N * (2 * nx + nx * nx + nx * nu)
    rec3[0, o_c] = np.inf
    _, code3, _, x3, y3 = swarm.step_host(rec3, prm, full=True)
    assert code3[0] == 2 and (code3[1:] == 0).all()
    du0d, coded, itd = swarm.step_host(rec2, prm)
    wx = x3.copy(); wy = y3.copy()
    wx[0], wy[0] = r2.primal[0], r2.dual[0]                                                   # what agent 0 kept
    r4 = plan.solve_batch_host(Px, q, Av2, l2, u2, prm, warm_x=wx, warm_y=wy)
    assert np.array_equal(coded, r4.code) and np.array_equal(itd, r4.iter) and np.array_equal(du0d, r4.primal[:, ub:ub + 2])
    # the shared-Jacobian form of the records gives the same tick (the vehicle's Jacobians do not depend on the agent)
    own, shared = L.split_shared(rec2)
    du0s, codes_, its = swarm.step_host(own, prm, shared_jac=shared)
    r5 = plan.solve_batch_host(Px, q, Av2, l2, u2, prm, warm_x=r4.primal, warm_y=r4.dual)
    assert np.array_equal(codes_, r5.code) and np.array_equal(its, r5.iter) and np.array_equal(du0s, r5.primal[:, ub:ub + 2])
    # no warm start requested: cold solve, and the stored warm starts are forgotten by reset_warmstart()
    du0e, codee, ite = swarm.step_host(rec, prm, warmstart=False)
    assert np.array_equal(ite, it) and np.array_equal(du0e, du0)
    swarm.reset_warmstart()
    du0f, codef, itf = swarm.step_host(rec, prm)
    assert np.array_equal(itf, it) and np.array_equal(du0f, du0)
    swarm.close()


@pytest.mark.parametrize("variant,K,B", [(6, 30, 64), (12, 50, 40), (6, 10, 7), (12, 20, 13), (6, 10, 1)])
def test_device_swarm_front_matches_host_swarm_front(sfb, variant, K, B):
    """MPCSwarmDevice (records -> device assembly, device-resident warm start) against MPCSwarm (host assembly,
    warm start through the host) over three closed-loop ticks: same inputs, codes and iteration counts.  (The device
    assembly's table form takes four agents per thread: 7, 13 and 1 leave its last group ragged.)"""
    u_h, c_h, i_h = M.mpc_swarm_step(variant, K, B, 3)
    u_d, c_d, i_d = M.mpc_swarm_step(variant, K, B, 3, device=True)
    assert np.array_equal(c_h, c_d) and np.array_equal(i_h, i_d) and np.array_equal(u_h, u_d)
    assert (c_d == 0).all() and np.all(np.abs(u_d) <= 0.5 + 1e-6)


def _packed_layout(sfb, L, rec):
    keep = L.jac_keep_of(rec)
    parts = [(int(k), int(d)) for k, d in zip(L.kind, L.dof)]
    return sfb.MPCLayout(L.nx, L.nu, L.ncr, L.kmesh, L.nivals, L.tf, L.alpha, L.D, parts=parts, crl=L.crl, cru=L.cru, jac_keep=keep)


@pytest.mark.parametrize("variant,K,batch", [(6, 10, 7), (12, 50, 120)])
def test_packed_records_assemble_to_the_same_bits(sfb, variant, K, batch):
    """sfb_mpc_layout::jac_keep: records that carry only the Jacobian entries flagged non-zero (the dense blocks of a
    bundle state are mostly structural zeros) assemble to the same A, l, u as the full records, which equal the host
    transcription bit for bit."""
    Av, l, u = M.mpc_assemble_batch(variant, K, batch, seed=9)
    L, rec = M.mpc_records(variant, K, batch, seed=9)
    Lp = _packed_layout(sfb, L, rec)
    packed = Lp.pack_records(rec)
    assert packed.shape[1] == Lp.record_doubles() < L.record_doubles() * 0.6
    A2, l2, u2 = _assemble_on_device(Lp, packed)
    assert np.array_equal(A2, Av) and np.array_equal(l2, l) and np.array_equal(u2, u)
    print("record doubles", L.record_doubles(), "->", Lp.record_doubles())


def test_swarm_switches_record_packing_in_place(sfb):
    """sfb_mpc_swarm_set_jac_keep: a swarm created for packed records takes a packed tick, is switched to unpacked
    records (what a front does when a linearisation does not fit the flags) and back; warm starts survive the switches
    -- every tick equals the same tick of a swarm that always used full records."""
    variant, K, B = 12, 50, 48
    d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
    plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K))
    prm = sfb.QPSolverParams(max_iter=4000)
    L, rec1 = M.mpc_records(variant, K, B, seed=21)
    _, rec2 = M.mpc_records(variant, K, B, seed=22)
    _, rec3 = M.mpc_records(variant, K, B, seed=23)
    Lp = _packed_layout(sfb, L, np.concatenate([rec1, rec2, rec3]))
    ref = sfb.MPCSwarm(plan, L, Pv, np.zeros(d["n"]), B)
    sw = sfb.MPCSwarm(plan, Lp, Pv, np.zeros(d["n"]), B)
    sw._rec_doubles = Lp.record_doubles()
    for t, rec in enumerate((rec1, rec2, rec3)):
        want = ref.step_host(rec, prm)
        if t == 1:
            assert sw.set_jac_keep(None) == L.record_doubles()
            got = sw.step_host(rec, prm)
            assert sw.set_jac_keep(Lp.jac_keep) == Lp.record_doubles()
        else:
            got = sw.step_host(Lp.pack_records(rec), prm)
        for a, b in zip(got, want):
            assert np.array_equal(a, b), "tick %d" % t
    ref.close(); sw.close()


def test_device_swarm_front_falls_back_to_full_records(sfb, monkeypatch):
    """MPCSwarmDevice packs its records with flags probed at construction and checks every record while packing; with
    a probe that saw nothing (test knob) the first tick does not fit, the front switches the swarm to full records and
    carries on: same closed loop as the host front."""
    u_h, c_h, i_h = M.mpc_swarm_step(6, 30, 64, 3)
    monkeypatch.setenv("SFB_MPC_PACK_PROBE_EMPTY", "1")
    u_d, c_d, i_d = M.mpc_swarm_step(6, 30, 64, 3, device=True)
    assert np.array_equal(c_h, c_d) and np.array_equal(i_h, i_d) and np.array_equal(u_h, u_d)
