#!/bin/bash
# A/B on the GPU box: pre-polishing waves next to the loop launch of the predicted order (SFB_SP_PREPOLISH = their number, 0 = off)
cd ${GRAFT_REPO_ROOT:-.}
timeout 600 python -m pytest tests/test_mpc_gpu.py -x -q -m gpu 2>&1 | tail -3
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pipelined --no-secondary --workload mpc"
for W in 0 64 128 192 256 384 512; do
  echo "SFB_SP_PREPOLISH=$W: $(SFB_SP_PREPOLISH=$W timeout 300 $B 2>/dev/null | tail -1 | python -c 'import sys,json; r=json.loads(sys.stdin.read()); p=r.get("parity_vs_oracle",{}); print("%.2f ms" % r["ms_per_step"])')"
done
echo "default with parity: $(timeout 600 python bench.py --steps 5 --warmup 2 --no-pipelined --no-secondary --workload mpc 2>/dev/null | tail -1 | python -c 'import sys,json; r=json.loads(sys.stdin.read()); p=r.get("parity_vs_oracle",{}); print("%.2f ms" % r["ms_per_step"], p.get("code_mismatches"), p.get("iter_mismatches"), p.get("max_abs_dx"))')"
