#!/bin/bash
# A/B of variant builds of libsfb.so on the GPU box (LIBS="smooth_feedback_amd/libsfb_x1.so ..."; "-" = the product build): every
# variant is copied over libsfb.so of the box's scratch copy of the repository in turn (the C++ harness links it by name), then
# the headline step and warm swarm ticks, twice each
cd ${GRAFT_REPO_ROOT:-.}
cp smooth_feedback_amd/libsfb.so /tmp/libsfb_product.so
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pipelined --no-secondary --no-closed-loop --workload mpc"
for rep in 1 2; do
  for v in ${LIBS:--}; do
    if [ "$v" = "-" ]; then cp /tmp/libsfb_product.so smooth_feedback_amd/libsfb.so; else cp $v smooth_feedback_amd/libsfb.so; fi
    h=$(timeout 300 $B 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms %s' % (d['ms_per_step'], d['parity_vs_oracle']['max_abs_dx'] if d.get('parity_vs_oracle') else ''))")
    t=$(timeout 300 python scripts/r6/tick_knobs.py "" 2>&1 | grep "warm ticks" | sed -e 's/.*mean of the last six //')
    echo "$v  headline $h   tick $t"
  done
done
cp /tmp/libsfb_product.so smooth_feedback_amd/libsfb.so
