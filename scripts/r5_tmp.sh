cd ${GRAFT_REPO_ROOT:-.}
for P in 0 1 2 3; do
echo "== policy $P"
KNOBS_POLICY=$P PLAN_DEBUG=1 timeout 500 python - <<'PY' 2>&1 | grep -v "amdgpu.ids\|segment columns\|slots per unit\|debug knob" | grep "unit engine\|B=    1 polish_iter=5\|B= 8192 polish_iter=5" | cut -c1-250
import os, sys, runpy
import smooth_feedback_amd as sfb
sfb.debug_set("SFB_PLAN_POLICY", os.environ["KNOBS_POLICY"])
sys.argv = ["lone_phases.py"]
runpy.run_path("scripts/r5/lone_phases.py", run_name="__main__")
PY
done
