#!/bin/bash
# Polishers behind the start gate (sp_gate_kernel): sparse + MPC tests, then the headline step and warm ticks for forced polisher
# counts (default: the rank kernel chooses one or two per compute unit)
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_qp_sparse_gpu.py tests/test_mpc_gpu.py -x -q -m gpu 2>&1 | tail -2
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pipelined --no-secondary --no-closed-loop --workload mpc"
for v in ${COUNTS:-256 512 768 1024 256}; do
  h=$(timeout 300 $B --debug-knob SFB_SP_POLISHERS=$v 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms %s' % (d['ms_per_step'], d['parity_vs_oracle']['max_abs_dx'] if d.get('parity_vs_oracle') else ''))")
  t=$(timeout 300 python scripts/r6/tick_knobs.py "SFB_SP_POLISHERS=$v" 2>&1 | grep "warm ticks" | sed -e 's/.*mean of the last six //')
  echo "polishers $v  headline $h   tick $t"
done
