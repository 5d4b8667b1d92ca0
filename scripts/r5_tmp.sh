cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mpc_gpu.py tests/test_qp_sparse_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 500 python scripts/r5/lone_phases.py 2>&1 | grep "polish_iter=5"
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/bench_w.json 2> gpurun_out/bench_w.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_w.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'pipelined', d['pipelined']['value'])
c=d['closed_loop']
print('tick', c['swarm_tick']['ms_per_tick'], 'half', c['swarm_tick']['half_swarm']['ms_per_tick'], 'single', c['single_agent']['cold_ms'], c['single_agent'].get('warm_ms'), 'e2e', c['end_to_end']['ms_per_step'])
print(d['phases_ms'])
PY
