#!/bin/bash
# Kernel timeline of warm closed-loop ticks of MPCSwarmDeviceLin (8 192 agents, nx = 12, K = 50): which dispatches make up a tick
# and how long each takes (rocprofv3 --kernel-trace), plus the stage laps of SFB_MPC_TIMING=1.
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$PWD
mkdir -p gpurun_out
timeout 300 python -c "
import sys; sys.path.insert(0, '$ROOT')
import smooth_feedback_amd as sfb; sfb.debug_set('SFB_MPC_TIMING', '1')
from examples import models_lib as M
r = M.mpc_swarm_devlin_step(12, 50, ${B:-8192}, 6, seed=1, want_records=False)
print('ms per tick', [round(1e3 * s, 2) for s in r['seconds']], 'mean iterations', r['iter'].mean())
" 2>&1 | grep -v amdgpu.ids | tail -40
cd /tmp; export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/tick_trace
rm -rf $OUT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python -c "
import sys; sys.path.insert(0, '$ROOT')
from examples import models_lib as M
r = M.mpc_swarm_devlin_step(12, 50, ${B:-8192}, 6, seed=1, want_records=False)
print('ms per tick', [round(1e3 * s, 2) for s in r['seconds']])
" > $OUT.log 2>&1
python - <<PY
import csv
rows = []
for r in csv.DictReader(open("$OUT/t_kernel_trace.csv")):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("sfb::", "").replace("(anonymous namespace)::", "").split("(")[0][:70], int(r["Grid_Size_X"])))
rows.sort()
# ticks start at mpc_linearise_kernel
starts = [i for i, r in enumerate(rows) if "mpc_linearise_kernel" in r[2]]
for ti, s in enumerate(starts):
    e = starts[ti + 1] if ti + 1 < len(starts) else len(rows)
    t0 = rows[s][0]
    print("tick", ti, "kernels", e - s, "span %.2f ms" % ((rows[e - 1][1] - t0) / 1e6))
    if ti >= len(starts) - 2:
        for r in rows[s:e]:
            print("   +%8.3f ms  %8.3f ms  grid %8d  %s" % ((r[0] - t0) / 1e6, (r[1] - r[0]) / 1e6, r[3], r[2]))
PY
