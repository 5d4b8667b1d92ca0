cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mpc_assembly_gpu.py tests/test_mpc_devlin_gpu.py tests/test_vehicle_swarm_gpu.py tests/test_multi_device_gpu.py -x -q -m gpu 2>&1 | tail -3
bash scripts/r4/tick_trace.sh 2>&1 | grep -A8 "^tick 5"
