cd ${GRAFT_REPO_ROOT:-.}
timeout 400 python -m pytest tests/test_qp_sparse_gpu.py tests/test_mpc_gpu.py tests/test_mpc_devlin_gpu.py tests/test_mpc_assembly_gpu.py -x -q -m gpu 2>&1 | tail -2
KNOBS=SFB_SP_GRID=4 N=300 BMAX=40 SEED=79 timeout 300 python scripts/fuzz_sparse.py 2>&1 | tail -1
KNOBS=SFB_SP_GRID=3,SFB_SP_PAUSE=27 N=200 BMAX=30 SEED=80 timeout 300 python scripts/fuzz_sparse.py 2>&1 | tail -1
timeout 300 python bench.py --steps 6 --warmup 2 --no-pipelined --no-secondary --no-closed-loop --workload mpc 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k: d['parity_vs_oracle'][k] for k in ('sample','code_mismatches','iter_mismatches','max_abs_dx')})"
