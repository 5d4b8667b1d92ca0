cd ${GRAFT_REPO_ROOT:-.}
bash scripts/r4/chk_ab.sh
KNOBS=SFB_SP_FORCE_LAT=1 N=300 SEED=77 timeout 300 python scripts/fuzz_sparse.py 2>&1 | tail -1
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pipelined --no-secondary --no-closed-loop --workload mpc"
for g in 1024 896; do
  timeout 300 $B --debug-knob SFB_SP_LAT_WAVES=$g 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('lat waves %-6s %9.0f QP/s  %.3f ms' % ('$g', d['value'], d['ms_per_step']))"
done
