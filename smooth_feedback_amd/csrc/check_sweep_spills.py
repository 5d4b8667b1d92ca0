#!/usr/bin/env python3
"""Build guard for qp_sparse.hip.  The triangular sweeps count their outstanding loads by hand (inline-asm stream
loads + `s_waitcnt vmcnt(N)`), so scratch (spill) traffic INSIDE a sweep would corrupt the counts.  Spills elsewhere
are merely slow.  Usage: check_sweep_spills.py <device assembly>.  A sweep region = a maximal run of the hand-issued
`global_load_dwordx4 ... off offset:N` stream loads (gaps of fewer than 400 lines); fails if a scratch instruction
lies inside one."""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
stream = [i for i, l in enumerate(lines) if re.search(r"global_load_dwordx4 .*off offset:\d+", l)]
if not stream:
    sys.exit("check_sweep_spills: no stream loads found (the sweeps changed?)")
regions, start, prev = [], stream[0], stream[0]
for i in stream[1:]:
    if i - prev > 400:
        regions.append((start, prev))
        start = i
    prev = i
regions.append((start, prev))
bad = [i for i, l in enumerate(lines) if re.search(r"\bscratch_(load|store)", l) and any(a <= i <= b for a, b in regions)]
if bad:
    sys.exit("check_sweep_spills: register spill inside a sweep (line %d of %s): hand-counted s_waitcnt would be wrong"
             % (bad[0] + 1, sys.argv[1]))
nspill = sum(1 for l in lines if re.search(r"\bscratch_(load|store)", l))
print("check_sweep_spills: %d sweep regions, %d scratch instructions, none inside a sweep" % (len(regions), nspill))
