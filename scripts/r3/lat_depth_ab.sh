cd ${GRAFT_REPO_ROOT:-.}
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pipelined --no-secondary --workload mpc"
for L in "" smooth_feedback_amd/libsfb_d16.so; do
  for LO in 384 448 512; do
    echo "lib=${L:-product} LAT_LO=$LO: $(SFB_LIB_PATH=$L SFB_SP_LAT_LO=$LO $B 2>/dev/null | tail -1 | python -c 'import sys,json; r=json.loads(sys.stdin.read()); print(r["ms_per_step"], r["parity_vs_oracle"]["iter_mismatches"] if "parity_vs_oracle" in r else None)')"
  done
done
