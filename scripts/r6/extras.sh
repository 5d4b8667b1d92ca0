#!/bin/bash
# Round-6 evidence besides the per-workload passes of scripts/profile_r6.sh (run on the GPU box; output under gpurun_out/r6_extras/,
# the text files are copied into profiles/r6_*/ by hand):
#   launch_trace.txt   dispatch durations of the headline step (kernel trace)
#   tick_trace.txt     kernel timeline of warm MPCSwarmDeviceLin ticks
#   lat_counters.txt   SQ / TCP / TCC counters of the LAT loop launch (polishers off)
#   phase_counters.txt counters per phase (SFB_SP_PHASED=1)
#   hip_api_steady.txt HIP API calls of steady-state device-pointer solves (--hip-trace + --kernel-trace, NO counters)
#   ekf_ab.txt         the persistent EKF kernel against the one-tile-per-wave kernel: durations and LDS / VMEM counters
ROOT=${GRAFT_REPO_ROOT:-$PWD}
X=$ROOT/gpurun_out/r6_extras; mkdir -p $X
cd $ROOT
bash scripts/r4/launch_trace.sh A=1 > $X/launch_trace.txt 2>&1
bash scripts/r4/tick_trace.sh > $X/tick_trace.txt 2>&1
bash scripts/r5/lat_counters.sh > $X/lat_counters.txt 2>&1
bash scripts/r4/phase_counters.sh > $X/phase_counters.txt 2>&1
cd /tmp; export TMPDIR=/tmp
# ---- HIP API calls of a steady-state solve ----
OUT=$ROOT/gpurun_out/r6_hip_api; rm -rf $OUT
timeout 600 rocprofv3 --hip-trace --kernel-trace --output-format csv -d $OUT -o t -- python $ROOT/bench.py --workload mpc --steps 4 --warmup 2 \
  --no-cpu-baseline --no-pipelined --no-secondary --no-closed-loop > $OUT.log 2>&1
python3 - > $X/hip_api_steady.txt <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/t_hip_api_trace.csv", recursive=True)
k = glob.glob("$OUT/**/t_kernel_trace.csv", recursive=True)
rows = sorted(((int(r["Start_Timestamp"]), r["Function"]) for r in csv.DictReader(open(f[0]))), key=lambda x: x[0])
# the timed steps: the calls between the last four launches of the rank kernel ... simply the calls of the LAST solve: from the
# hipMemsetAsync that precedes the last first-launch to the end of the stream's enqueue
names = [n for _, n in rows]
idx = [i for i, n in enumerate(names) if n == "hipLaunchKernel" or n == "hipModuleLaunchKernel" or n == "hipExtModuleLaunchKernel"]
print("HIP API calls of the whole run: %d; by name:" % len(names))
for n, c in collections.Counter(names).most_common(): print("   %6d  %s" % (c, n))
# one steady-state solve = the API calls between two consecutive hipEventRecord pairs of bench.py's timed loop: take the last window
# that contains exactly the launches of one solve (7 kernel launches: first, rank, loop, polishers, finish, final + ...)
ev = [i for i, n in enumerate(names) if n == "hipEventRecord"]
print("\\ncalls from the start of the last timed step's enqueue to its end (between bench.py's event records):")
# bench.py records an event before and after each step on the launch stream; the library records its own: find the longest tail window
tail = names[idx[-8] - 6: idx[-1] + 6] if len(idx) >= 8 else names[-60:]
for n in tail: print("   ", n)
PY
# ---- EKF A/B ----
EK="python $ROOT/bench.py --workload ekf --steps 10 --warmup 2 --no-cpu-baseline --no-secondary"
for v in persistent one_tile; do
  K=""; [ $v = one_tile ] && K="--debug-knob SFB_EKF_PERSISTENT=0"
  rm -rf $ROOT/gpurun_out/r6_ekf_$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r6_ekf_$v/trace -o t -- $EK $K > /dev/null 2>&1
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
             "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --output-format csv -d $ROOT/gpurun_out/r6_ekf_$v/p$i -o p -- $EK $K > /dev/null 2>&1
  done
done
python3 - > $X/ekf_ab.txt <<PY
import csv, glob, collections
for v in ("persistent", "one_tile"):
    print("==", v)
    f = glob.glob("$ROOT/gpurun_out/r6_ekf_%s/trace/**/t_kernel_stats.csv" % v, recursive=True)
    for r in csv.DictReader(open(f[0])):
        if "ekf" in r["Name"]: print("   %-60s calls %s  average %.4f ms" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e6))
    for i in (1, 2, 3):
        g = glob.glob("$ROOT/gpurun_out/r6_ekf_%s/p%d/**/p_counter_collection.csv" % (v, i), recursive=True)
        if not g: continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(g[0])):
            if "ekf" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, vals in sorted(acc.items()): print("   %-26s %16.0f per dispatch (%d)" % (c, sum(vals) / len(vals), len(vals)))
PY
echo extras done
