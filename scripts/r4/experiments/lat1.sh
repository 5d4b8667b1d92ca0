cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_qp_sparse_gpu.py tests/test_mpc_gpu.py -x -q -m gpu 2>&1 | tail -3
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pipelined --no-secondary --no-closed-loop --workload mpc"
for v in "A=1" "SFB_PLAN_WIDE=0" "SFB_LIB_PATH=smooth_feedback_amd/libsfb_c4.so" "SFB_LIB_PATH=smooth_feedback_amd/libsfb_c4h.so" "A=1"; do
  echo "== $v"
  env $v timeout 300 $B 2>&1 | tail -1 | cut -c1-180
done
