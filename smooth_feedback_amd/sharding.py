"""Batch sharding over the GPUs of one node: items are independent (no cross-problem reduction
anywhere in qp_solver.hpp, mpc.hpp:458-519 or ekf.hpp), so each rank owns a contiguous range and the
only exchange is one gather of the small per-item outputs (u_0, code, iter) at the end of a step.
Backend-agnostic torch.distributed (nccl == RCCL over xGMI on the GPU node, gloo in CPU tests)."""
import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int):
    """Contiguous split [lo, hi) of `total` items; the remainder goes to the lowest ranks."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_small_outputs(local: torch.Tensor, total: int, force: bool = False):
    """All ranks contribute their shard's rows (dim 0); returns the (total, ...) tensor on every rank.
    Shards may differ by one row, so pad to the largest shard for the collective.
    force: issue the collective even in a world of one (tests: the RCCL path on a 1-GPU box)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(total, r, world) for r in range(world)]
    maxrows = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((maxrows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(out, sizes)], dim=0)
