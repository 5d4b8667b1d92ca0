"""N > 1 path on CPU: world_size-2 gloo processes shard a batch, each 'solves' its shard (here: the
CPU oracle stands in for the device results, this test is about the host-side sharding/gather logic)
and the gathered small outputs equal the single-process result."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, total, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from smooth_feedback_amd.sharding import gather_small_outputs, shard_range
    from oracle import loader as O
    import smooth_feedback_amd as sfb
    P, q, A, l, u = sfb.random_qp_batch(5, total, 20, 10, 1.0)   # every rank generates the same global batch
    lo, hi = shard_range(total, rank, world)
    r = O.qp_dense_solve_batch(P[lo:hi], q[lo:hi], A[lo:hi], l[lo:hi], u[lo:hi], params=O.default_params(max_iter=500))
    small = torch.from_numpy(np.stack([r["iter"].astype(np.int64), r["code"].astype(np.int64)], axis=1))
    full = gather_small_outputs(small, total)
    if rank == 0:
        ret.put(full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_partition_the_batch():
    sys.path.insert(0, ROOT)
    from smooth_feedback_amd.sharding import shard_range
    for total in (0, 1, 7, 64, 65537):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_two_rank_gather_equals_single_process():
    sys.path.insert(0, ROOT)
    from oracle import loader as O
    import smooth_feedback_amd as sfb
    O.lib()
    total, world = 37, 2
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, ret)) for r in range(world)]
    for p in procs:
        p.start()
    full = ret.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    P, q, A, l, u = sfb.random_qp_batch(5, total, 20, 10, 1.0)
    ref = O.qp_dense_solve_batch(P, q, A, l, u, params=O.default_params(max_iter=500))
    assert np.array_equal(full[:, 0], ref["iter"].astype(np.int64))
    assert np.array_equal(full[:, 1], ref["code"].astype(np.int64))
