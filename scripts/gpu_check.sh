#!/bin/bash
# quick GPU loop: parity tests of the dense QP kernel + bench line (run on the GPU box)
cd ${GRAFT_REPO_ROOT:-.}
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 300 python bench.py --steps 5 --warmup 1 "$@" 2>&1 | tail -1 | cut -c1-2000
