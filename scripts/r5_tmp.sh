cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for P in 16 32 64 96 128 256; do
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --debug-knob SFB_SP_POLISHERS=$P > gpurun_out/bench_w.json 2> gpurun_out/bench_w.err
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_w.json').read().strip().splitlines()[-1])
c=d['closed_loop']
print($P, round(d['value']), round(d['ms_per_step'],2), 'pipelined', round(d['pipelined']['value']), 'tick', round(c['swarm_tick']['ms_per_tick'],2), 'half', round(c['swarm_tick']['half_swarm']['ms_per_tick'],2), 'e2e', round(c['end_to_end']['ms_per_step'],2), 'single', round(c['single_agent']['cold_ms'],3))
PY
done
