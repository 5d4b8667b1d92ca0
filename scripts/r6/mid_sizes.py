"""The dense size table of bench.py alone (secondary.qp_dense_sizes): QP-iterations/s per size and parameter set, parity with the
oracle on a sample per size.  Usage on the GPU box: python scripts/r6/mid_sizes.py [before.json] -- with a bench line of an
earlier build as argument the ratio per row is printed next to it."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import smooth_feedback_amd as sfb

rows = bench.dense_sizes_table(sfb, torch.device("cuda:0"), bench.host_cpus()[0])
before = {}
if len(sys.argv) > 1:
    b = json.load(open(sys.argv[1]))
    for r in b.get("secondary", {}).get("qp_dense_sizes", []):
        before[(r["n"], r["m"], r["params"])] = r
for r in rows:
    key = (r["n"], r["m"], r["params"])
    ratio = r["qp_iterations_per_s"] / before[key]["qp_iterations_per_s"] if key in before else float("nan")
    print("(%3d,%3d) %-20s %8.3f ms  %10.4g QP-it/s  x%.3f  parity %s" % (r["n"], r["m"], r["params"], r["ms"], r["qp_iterations_per_s"], ratio,
          json.dumps(r.get("parity_vs_oracle", r.get("parity")))[:120]))
