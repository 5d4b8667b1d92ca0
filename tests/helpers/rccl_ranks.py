"""Run under torch.distributed.run by tests/test_bench_gpu.py: `world` ranks, ALL bound to cuda:0, process group on nccl
(= RCCL), then the one exchange of the N-GPU path -- sharding.gather_small_outputs -- on an UNEQUAL split (total = 2 B + 1 rows:
shards differ by one row, the padding of sharding.py:27-31), the int64 checksum all_reduce and the MAX all_reduce of bench.py.
Prints one JSON line per rank; an exception of the backend (RCCL may refuse two ranks on one device) is reported as data."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smooth_feedback_amd.sharding import gather_small_outputs, shard_range  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    B = int(sys.argv[1])
    total = 2 * B + 1
    res = {"rank": rank, "world": world, "total": total, "backend": "nccl"}
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
        lo, hi = shard_range(total, rank, world)
        rows = torch.arange(lo, hi, device="cuda", dtype=torch.float64)
        local = torch.stack([rows, rows * 0.5, rows + 0.25, -rows], dim=1)  # (u0, u1, code, iter)-shaped rows
        out = gather_small_outputs(local, total, force=True)
        allr = torch.arange(0, total, device="cuda", dtype=torch.float64)
        expect = torch.stack([allr, allr * 0.5, allr + 0.25, -allr], dim=1)
        res["shard_rows"] = hi - lo
        res["gathered_rows"] = int(out.shape[0])
        res["gather_exact"] = bool(torch.equal(out, expect))
        chk = torch.tensor([int(rows.sum().item())], dtype=torch.int64, device="cuda")
        dist.all_reduce(chk, op=dist.ReduceOp.SUM)
        res["checksum_ok"] = int(chk.item()) == total * (total - 1) // 2
        tmax = torch.tensor([float(rank + 1)], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        res["max_ok"] = float(tmax.item()) == float(world)
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001 -- the backend's refusal is the measurement
        res["error"] = "%s: %s" % (type(e).__name__, str(e)[:600])
    print("RCCL_RANKS " + json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
