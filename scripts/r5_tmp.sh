cd ${GRAFT_REPO_ROOT:-.}
bash scripts/r4/chk_ab.sh
KNOBS=SFB_SP_FORCE_LAT=1 N=400 SEED=77 timeout 300 python scripts/fuzz_sparse.py 2>&1 | tail -1
KNOBS=SFB_SP_GRID=4 N=400 BMAX=40 SEED=78 timeout 300 python scripts/fuzz_sparse.py 2>&1 | tail -1
