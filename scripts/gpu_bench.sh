#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
SECONDS=0; timeout 900 python bench.py 2> gpurun_out/bench_default.err | tail -1 > gpurun_out/bench_default.json; echo "bench.py wall $SECONDS s"
tail -3 gpurun_out/bench_default.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/bench_default.json"))
def show(name, d):
    print(name, "value %.4g %s" % (d["value"], d["unit"]), "kernel_ms %.3f" % d["roofline"]["kernel_ms"], "frac %.3g" % d["roofline"]["frac"],
          "traffic", d["roofline"].get("traffic"), d["roofline"].get("traffic_source"), "alt", d.get("roofline_alt", {}).get("frac"))
    print("   cpu", {k: (v if not isinstance(v, dict) else v) for k, v in d.get("cpu_baseline", {}).items() if k != "sample"})
    print("   parity", d.get("parity_vs_oracle"))
show("headline", r)
for k, v in r.get("secondary", {}).items():
    if isinstance(v, dict) and "roofline" in v: show(k, v)
    else: print(k, json.dumps(v)[:1500])
print("pipelined", r.get("pipelined", {}).get("value"))
PY
