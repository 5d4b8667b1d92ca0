cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for i in 1 2; do
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-closed-loop > gpurun_out/bench_w.json 2> gpurun_out/bench_w.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_w.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'pipelined', d['pipelined']['value'], d['pipelined']['ms_per_step'])
PY
done
