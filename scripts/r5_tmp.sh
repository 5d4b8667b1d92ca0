# scratch script of round 5 (whatever was measured last on the GPU box); the recipes that matter are the named scripts next to it
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/bench_r5_final.json 2> gpurun_out/bench_r5_final.err
cut -c1-400 gpurun_out/bench_r5_final.json
