"""Sweep schedules of the headline plan simulated on the CPU, before writing kernels for them ("the floor of
this formulation").  Builds the pattern of L for the pruned MPC pattern in the plan's elimination order (symbolic
elimination in Python; nnz(L) must equal the plan's) and reports
  * the critical paths of the two sweeps in dependent updates, the longest rows / columns, the width bound;
  * G-term slots: a slot = up to G consecutive updates of ONE target accumulated in registers (list scheduling like
    sparse_plan.cpp, 64 or 128 slots per unit): units, fill, streamed values with lane-masked loads;
  * row owners: every target's chain accumulated by one lane (M chains per wave), LDS reads prefetched, a dependent read
    usable D steps after the producing row's last fma: steps, the same with unlimited lanes.
No GPU needed:  python scripts/r3/sched_sim.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import smooth_feedback_amd as sfb
from examples import models_lib as M
sys.setrecursionlimit(100000)


def pattern_of_L(variant=12, K=50):
    d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
    Av, l, u = M.mpc_assemble_batch(variant, K, 64, seed=3, threads=8)
    keep = np.any(Av != 0.0, axis=0)
    plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K), keep=keep)
    perm = np.array(plan.perm)
    n, m = d["n"], d["m"]
    k = n + m
    adj = [set() for _ in range(k)]
    for j in range(n):
        for p in range(Pp[j], Pp[j + 1]):
            i = Pi[p]
            if i != j:
                adj[i].add(j); adj[j].add(i)
    for i in range(m):
        for p in range(Ap[i], Ap[i + 1]):
            if keep[p]:
                adj[n + i].add(Aj[p]); adj[Aj[p]].add(n + i)
    pinv = np.empty(k, int)
    pinv[perm] = np.arange(k)
    struct = [None] * k
    for old in range(k):
        struct[pinv[old]] = set(int(pinv[x]) for x in adj[old] if pinv[x] > pinv[old])
    cols = []
    for j in range(k):
        s = struct[j]
        if s:
            p = min(s)
            struct[p] |= (s - {p})
        cols.append(sorted(s))
    return k, cols


def chains(k, cols, forward):
    rows = [[] for _ in range(k)]
    for j in range(k):
        for i in cols[j]:
            rows[i].append(j)
    return [list(r) for r in rows] if forward else [list(reversed(c)) for c in cols]


def critical_path(terms):
    fin = [None] * len(terms)
    def f(t):
        if fin[t] is None:
            cur = 0
            for p in terms[t]:
                cur = max(cur, f(p)) + 1
            fin[t] = cur
        return fin[t]
    return max(f(t) for t in range(len(terms)))


def consumers(terms):
    cons = [[] for _ in terms]
    for t, ps in enumerate(terms):
        for pos, p in enumerate(ps):
            cons[p].append((t, pos))
    return cons


def g_slots(terms, G, cap):
    k, cons = len(terms), consumers(terms)
    after = [None] * k
    def aft(t):
        if after[t] is None:
            after[t] = max([(len(terms[c]) - pos) + aft(c) for c, pos in cons[t]], default=0)
        return after[t]
    for t in range(k): aft(t)
    final = [(-1 if not terms[t] else None) for t in range(k)]
    nxt, active, steps, s = [0] * k, set(t for t in range(k) if terms[t]), [], 0
    ready = lambda p: final[p] is not None and final[p] < s
    while active:
        cand = sorted((-(len(terms[t]) - nxt[t] + after[t]), t) for t in active if ready(terms[t][nxt[t]]))
        this = []
        for _, t in cand[:cap]:
            g = 0
            while g < G and nxt[t] + g < len(terms[t]) and ready(terms[t][nxt[t] + g]): g += 1
            this.append((t, g))
        for t, g in this:
            nxt[t] += g
            if nxt[t] == len(terms[t]):
                final[t] = s; active.discard(t)
        steps.append(this); s += 1
    return steps


def row_owners(terms, Mch, D):
    k, cons = len(terms), consumers(terms)
    H = [None] * k
    def hh(t):
        if H[t] is None:
            H[t] = max([D + (len(terms[c]) - pos) + hh(c) for c, pos in cons[t]], default=0)
        return H[t]
    for t in range(k): hh(t)
    asap = [None] * k
    def ff(t):
        if asap[t] is None:
            if not terms[t]: asap[t] = -10 ** 9
            else:
                cur = 0
                for p in terms[t]: cur = max(cur + 1, ff(p) + D)
                asap[t] = cur
        return asap[t]
    cp = max(ff(t) for t in range(k))
    final = [(-10 ** 9 if not terms[t] else None) for t in range(k)]
    todo = [t for t in range(k) if terms[t]]
    unstarted, mach, s, done = set(todo), [None] * Mch, 0, 0
    while done < len(todo):
        newly = []
        for mi in range(Mch):
            if mach[mi] is None: continue
            t, pos = mach[mi]
            if pos < 0: mach[mi] = (t, 0); continue   # the step that loads the right-hand side
            p = terms[t][pos]
            if final[p] is not None and final[p] + D <= s:
                pos += 1
                if pos == len(terms[t]): newly.append(t); mach[mi] = None; done += 1
                else: mach[mi] = (t, pos)
        for t in newly: final[t] = s
        free = [mi for mi in range(Mch) if mach[mi] is None]
        if free and unstarted:
            cand = sorted((-(len(terms[t]) + H[t]), t) for t in unstarted if final[terms[t][0]] is not None and final[terms[t][0]] + D <= s + 2)
            for (_, t), mi in zip(cand, free):
                mach[mi] = (t, -1); unstarted.discard(t)
        s += 1
    return s, cp


if __name__ == "__main__":
    k, cols = pattern_of_L()
    nnz = sum(len(c) for c in cols)
    print("k %d nnz(L) %d width bound %d units of 128" % (k, nnz, -(-nnz // 128)))
    for fwd in (True, False):
        T = chains(k, cols, fwd)
        name = "forward" if fwd else "backward"
        print("%s: critical path %d dependent updates; longest chains %s" % (name, critical_path(T), sorted(len(x) for x in T)[-8:]))
        for G in (1, 2, 4, 8):
            for cap in (64, 128):
                st = g_slots(T, G, cap)
                S = cap // 64
                streamed = sum(-(-len(x) // S) * S * G for x in st)
                print("   G %d, %3d slots per unit: %3d units, fill of the used slots %.2f, streamed values %.2f x nnz" % (
                    G, cap, len(st), sum(g for x in st for _, g in x) / max(1, sum(len(x) for x in st) * G), streamed / nnz))
        for Mch in (64, 128):
            for D in (2, 3, 4):
                steps, cp = row_owners(T, Mch, D)
                print("   row owners, %3d chains per wave, D %d: %3d steps (unlimited lanes: %d)" % (Mch, D, steps, cp))
