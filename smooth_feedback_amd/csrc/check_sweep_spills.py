#!/usr/bin/env python3
"""Build guard for qp_sparse.hip.  The triangular sweeps count their outstanding loads by hand (inline-asm stream
loads + `s_waitcnt vmcnt(N)`), so scratch (spill) traffic INSIDE a sweep would corrupt the counts.  Spills elsewhere
are merely slow.  Usage: check_sweep_spills.py <device assembly>.  A sweep region = a maximal run of the hand-issued
`global_load_dwordx4 ... off offset:N` stream loads (gaps of fewer than 400 lines); fails if a scratch instruction
lies inside one.

Second check (the sweeps keep stream loads IN FLIGHT across compiler-generated code: the prefetch registers lx / ix are
written asynchronously, which the compiler does not know): inside every sweep region the outstanding loads are replayed
in program order -- a load enters a FIFO with its destination registers, `s_waitcnt vmcnt(N)` retires all but the N
youngest -- and any other instruction that names a register of a load still in flight (a copy, a rename, a read before
the counted wait) fails the build."""
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


def regs_of(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(a) for a in re.findall(r"\bv(\d+)\b", text))
    return out


hazards = 0
for a, b in regions:
    fifo = []  # destination register sets of the loads in flight, oldest first
    lds = []   # the same for the LDS reads of the hand-scheduled units (retired by s_waitcnt lgkmcnt(N), in order)
    for i in range(a, b + 1):
        t = lines[i].split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        if op.startswith(("global_load", "flat_load", "buffer_load")):
            dst = regs_of(t.split(",")[0])
            busy = set().union(*fifo) if fifo else set()
            src = regs_of(",".join(t.split(",")[1:]))
            if src & busy:
                sys.exit("check_sweep_spills: line %d of %s: address of a load is a register of a load still in flight: %s" % (i + 1, sys.argv[1], t))
            fifo.append(dst)
            continue
        if op.startswith(("global_store", "flat_store", "buffer_store")):
            fifo.append(set())  # stores count in vmcnt on gfx9
        if op.startswith("ds_read"):
            busy_l = set().union(*lds) if lds else set()
            if regs_of(",".join(t.split(",")[1:])) & (busy_l | (set().union(*fifo) if fifo else set())):
                sys.exit("check_sweep_spills: line %d of %s: address of an LDS read is a register still in flight: %s" % (i + 1, sys.argv[1], t))
            lds.append(regs_of(t.split(",")[0]))
            continue
        if op.startswith("ds_write"):
            busy_l = set().union(*lds) if lds else set()
            if regs_of(t) & (busy_l | (set().union(*fifo) if fifo else set())):
                sys.exit("check_sweep_spills: line %d of %s: an LDS write names a register still in flight: %s" % (i + 1, sys.argv[1], t))
            lds.append(set())  # writes count in lgkmcnt as well
            continue
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", t)
            if m:
                n = int(m.group(1))
                while len(fifo) > n:
                    fifo.pop(0)
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m:
                n = int(m.group(1))
                while len(lds) > n:
                    lds.pop(0)
            if re.fullmatch(r"s_waitcnt\s+0(x0+)?", t):
                fifo, lds = [], []
            continue
        busy = (set().union(*fifo) if fifo else set()) | (set().union(*lds) if lds else set())
        if busy and (regs_of(t) & busy):
            hazards += 1
            sys.exit("check_sweep_spills: line %d of %s touches a register of a stream load that is still in flight "
                     "(before its counted s_waitcnt): %s" % (i + 1, sys.argv[1], t))
# Third check (rounds 5-6): the LAT kernel keeps the head of the forward factor stream RESIDENT in its AccVGPRs across compiler-
# generated code (lat_resident_load / the resident prefix of the forward sweep).  The compiler does not know that, and it does use
# AccVGPRs itself when the architectural registers are short (gfx950 allocates long-lived values there).  The two sides share the
# file by number: a[0 .. kAgprFree) are the compiler's, a[kAgprFree .. 255] the resident stream's (kAgprFree: a constant of
# qp_sparse.hip, read from the source next to this script).  Every access to an AccVGPR >= kAgprFree has to be one of ours -- a
# literal `v_accvgpr_read_b32 v, a[N]` of the prefix or a `global_load_dwordx4 a[N:M]` of the load -- and none of ours may name
# one below it.
import os
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "qp_sparse.hip")).read()
AGPR_FREE = int(re.search(r"constexpr int kAgprFree\s*=\s*(\d+)\s*;", src).group(1))
NUM = r"(0x[0-9a-fA-F]+|\d+)"  # (the assembler prints a literal register number the way the operand was formed: a[64:0x43])
ours = re.compile(r"v_accvgpr_read_b32 v\d+, a\[" + NUM + r"\]|global_load_dwordx4 a\[" + NUM + ":" + NUM + r"\]")
for i, l in enumerate(lines):
    code = l.split(";")[0]
    idx = [int(x, 0) for x in re.findall(r"\ba\[?" + NUM, code)] + [int(x, 0) for pr in re.findall(r"\ba\[" + NUM + ":" + NUM + r"\]", code) for x in pr]
    if not idx:
        continue
    if ours.search(code):
        if min(idx) < AGPR_FREE:
            sys.exit("check_sweep_spills: line %d of %s: the resident stream names an AccVGPR that belongs to the compiler: %s" % (i + 1, sys.argv[1], l.strip()))
    elif max(idx) >= AGPR_FREE:
        sys.exit("check_sweep_spills: line %d of %s: an access to the resident stream's AccVGPRs (a%d and up) that is not the resident stream's: %s"
                 % (i + 1, sys.argv[1], AGPR_FREE, l.strip()))
nspill = sum(1 for l in lines if re.search(r"\bscratch_(load|store)", l))
print("check_sweep_spills: %d sweep regions, %d scratch instructions, none inside a sweep; no instruction touches a register of a "
      "load in flight" % (len(regions), nspill))
