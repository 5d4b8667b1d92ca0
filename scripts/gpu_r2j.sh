#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_mpc_gpu.py tests/test_qp_sparse_gpu.py tests/test_mpc_assembly_gpu.py tests/test_asif_gpu.py -m gpu -x -q 2>&1 | tail -12
echo "== lone wave LDL phases =="; B=1 SFB_LIB_PATH=$PWD/smooth_feedback_amd/libsfb_prof.so timeout 300 python scripts/ldl_prof.py 2>&1 | grep -v amdgpu.ids
for b in 1 8192; do echo "=== B $b ==="; B=$b timeout 300 python scripts/mpc_time.py 2>&1 | grep -v amdgpu.ids | sed -n 2,3p; done
