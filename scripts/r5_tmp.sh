cd ${GRAFT_REPO_ROOT:-.}
bash scripts/r4/chk_ab.sh
timeout 600 python bench.py --steps 6 --warmup 2 --no-pipelined --no-secondary --no-closed-loop --workload mpc 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['parity_vs_oracle'])"
