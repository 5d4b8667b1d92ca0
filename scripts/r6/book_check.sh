#!/bin/bash
# round 6, after the per-device launch bookkeeping (csrc/qp_sparse.hip SparseDeviceBook): sparse + MPC GPU tests incl. two plans on
# two streams, the RCCL tests, the headline step, two batches in flight, one controller through the host entry, and the HIP API
# calls of steady-state solves (rocprofv3 --hip-trace, no counters)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_qp_sparse_gpu.py tests/test_mpc_gpu.py tests/test_bench_gpu.py -x -q -m gpu 2>&1 | tail -3
cat gpurun_out/rccl_two_ranks_one_device.json 2>/dev/null | head -40
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --workload mpc 2> gpurun_out/book_bench.err | tail -1 > gpurun_out/book_bench.json
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/book_bench.json"))
print("headline %.0f QP/s %.3f ms" % (d["value"], d["ms_per_step"]))
print("pipelined", json.dumps(d.get("pipelined")))
cl = d.get("closed_loop", {})
print("swarm_tick", cl.get("swarm_tick", {}).get("ms_per_tick"), "end_to_end", cl.get("end_to_end", {}).get("ms_per_step"))
print("single_agent", json.dumps(cl.get("single_agent")))
print("parity", d.get("parity_vs_oracle"))
PY
