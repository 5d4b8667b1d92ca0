#!/bin/bash
# MPC GPU tests + the headline launch time of the current build (run on the GPU box).  Every command is bounded: a wrong
# kernel runs its items into the device's iteration cap and would otherwise hold the box for minutes.
cd ${GRAFT_REPO_ROOT:-.}
timeout 300 python -m pytest tests/test_mpc_gpu.py -x -q -m gpu 2>&1 | tail -2 || exit 1
B="timeout 120 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pipelined --no-secondary --workload mpc"
for rep in 1 2; do echo "launch: $($B 2>/dev/null | tail -1 | python -c 'import sys,json; r=json.loads(sys.stdin.read()); print("%.2f ms" % r["ms_per_step"])')"; done
echo "with parity: $(timeout 200 python bench.py --steps 5 --warmup 2 --no-pipelined --no-secondary --workload mpc 2>/dev/null | tail -1 | python -c 'import sys,json; r=json.loads(sys.stdin.read()); p=r.get("parity_vs_oracle",{}); print("%.2f ms" % r["ms_per_step"], p.get("code_mismatches"), p.get("iter_mismatches"), p.get("max_abs_dx"))')"
timeout 120 python scripts/single_agent_latency.py 2>/dev/null | tail -1
