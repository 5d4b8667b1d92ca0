"""Microbenchmark of the triangular-sweep UNIT of the sparse kernel for a lone wave (scripts/ubench/sweep_units.hip):
what a unit of 128 updates costs as a function of its instruction mix.  Generates sweep_units_asm.h: for every variant
a HEAD macro (first eight units' loads + entry state) and a BLOCK macro (eight units).  Same register conventions as
scripts/r4/gen_sweep_lat.py (v168 = 16 lane, v169 = 8 lane, v170 = LDS base; v176.. literal)."""
import sys
sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/r4/experiments")
import io, contextlib
with contextlib.redirect_stdout(io.StringIO()):
    import gen_sweep_lat as g

v2, v4, LX = g.v2, g.v4, g.LX
A0, B0, A1, B1 = g.A0, g.B0, g.A1, g.B1


def drop(lines, what):
    keep = []
    for l in lines:
        if what == "unpack" and l.startswith("v_add_u32_sdwa"): continue
        if what == "loads" and l.startswith("global_load"): continue
        if what == "lds" and l.startswith("ds_"): continue
        if what == "fma" and l.startswith("v_fma"): continue
        keep.append(l)
    if what == "loads":
        keep = [l for l in keep if "vmcnt" not in l]
    if what == "lds":
        keep = [l for l in keep if "lgkmcnt" not in l]
    return keep


# ---- direct 32-bit LDS addresses in the index stream: (p0, t0, p1, t1) per lane and unit, no unpack ----
IXD = lambda d: 208 + 4 * d
DA0, DB0, DA1, DB1, DC0, DD0, DC1, DD1 = 240, 242, 244, 246, 248, 250, 252, 254


def dloads(d):
    return ["global_load_dwordx4 %s, v168, %%[sv%d] offset:%d" % (v4(LX(d)), d // 4, (d % 4) * 1024),
            "global_load_dwordx4 %s, v168, %%[si%d] offset:%d" % (v4(IXD(d)), d // 4, (d % 4) * 1024)]


def dreads(regs, d):
    a, b, c, e = regs
    return ["ds_read_b64 %s, v%d" % (v2(a), IXD(d)), "ds_read_b64 %s, v%d" % (v2(b), IXD(d) + 1),
            "ds_read_b64 %s, v%d" % (v2(c), IXD(d) + 2), "ds_read_b64 %s, v%d" % (v2(e), IXD(d) + 3)]


def direct_block(one_wait, vm=14):
    t = []
    for d in range(8):
        if one_wait:
            t += ["s_waitcnt lgkmcnt(0)", "v_fma_f64 %s, -%s, %s, %s" % (v2(DA0), v2(LX(d)), v2(DA0), v2(DB0)),
                  "v_fma_f64 %s, -%s, %s, %s" % (v2(DA1), v2(LX(d) + 2), v2(DA1), v2(DB1))]
        else:
            t += ["s_waitcnt lgkmcnt(2)", "v_fma_f64 %s, -%s, %s, %s" % (v2(DA0), v2(LX(d)), v2(DA0), v2(DB0)),
                  "s_waitcnt lgkmcnt(0)", "v_fma_f64 %s, -%s, %s, %s" % (v2(DA1), v2(LX(d) + 2), v2(DA1), v2(DB1))]
        t += ["ds_write_b64 v%d, %s" % (IXD(d) + 1, v2(DA0)), "ds_write_b64 v%d, %s" % (IXD(d) + 3, v2(DA1))]
        # unit d+1's stream has arrived: in flight behind it are the loads of units d+2 .. d+7 (12) [+ nothing of unit d yet]
        t += ["s_waitcnt vmcnt(12)"] + dreads((DA0, DB0, DA1, DB1), (d + 1) % 8) + dloads(d)
    return t


def direct_head():
    t = []
    for d in range(8):
        t += dloads(d)
    return t + ["s_waitcnt vmcnt(14)"] + dreads((DA0, DB0, DA1, DB1), 0)


def direct_wide_block():
    """pairs (0,1), (2,3), ...: the reads of both units in flight together, one LDS round trip per pair"""
    t = []
    for pq in range(4):
        d0, d1 = 2 * pq, 2 * pq + 1
        t += ["s_waitcnt lgkmcnt(6)", "v_fma_f64 %s, -%s, %s, %s" % (v2(DA0), v2(LX(d0)), v2(DA0), v2(DB0)),
              "s_waitcnt lgkmcnt(4)", "v_fma_f64 %s, -%s, %s, %s" % (v2(DA1), v2(LX(d0) + 2), v2(DA1), v2(DB1)),
              "s_waitcnt lgkmcnt(2)", "v_fma_f64 %s, -%s, %s, %s" % (v2(DC0), v2(LX(d1)), v2(DC0), v2(DD0)),
              "s_waitcnt lgkmcnt(0)", "v_fma_f64 %s, -%s, %s, %s" % (v2(DC1), v2(LX(d1) + 2), v2(DC1), v2(DD1)),
              "ds_write_b64 v%d, %s" % (IXD(d0) + 1, v2(DA0)), "ds_write_b64 v%d, %s" % (IXD(d0) + 3, v2(DA1)),
              "ds_write_b64 v%d, %s" % (IXD(d1) + 1, v2(DC0)), "ds_write_b64 v%d, %s" % (IXD(d1) + 3, v2(DC1))]
        t += ["s_waitcnt vmcnt(8)"] + dreads((DA0, DB0, DA1, DB1), (d0 + 2) % 8) + dreads((DC0, DD0, DC1, DD1), (d0 + 3) % 8)
        t += dloads(d0) + dloads(d1)
    return t


def direct_wide_head():
    t = []
    for d in range(8):
        t += dloads(d)
    return t + ["s_waitcnt vmcnt(12)"] + dreads((DA0, DB0, DA1, DB1), 0) + dreads((DC0, DD0, DC1, DD1), 1)


# ---- deeper stream: D units in flight, the VALUES in AGPRs (a wave alone on its SIMD has 256 of them for free), moved to
# VGPRs one unit ahead of their use (4 v_accvgpr_read per unit); packed indices in VGPRs ----
AL = 160                                        # the current unit's values: v[160:163]
AIX = lambda d: 176 + 2 * d                     # packed indices of unit d (D <= 24: v176 .. v223)


def a_unpack(d, s):
    out = []
    for dst, src, half in ((g.P0(s), AIX(d), 1), (g.T0(s), AIX(d), 0), (g.P1(s), AIX(d) + 1, 1), (g.T1(s), AIX(d) + 1, 0)):
        out.append("v_add_u32_sdwa %s, v170, v%d dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_%d" % (dst, src, half))
    return out


def a_loads(d):
    return ["global_load_dwordx4 a[%d:%d], v168, %%[sv%d] offset:%d" % (4 * d, 4 * d + 3, d // 4, (d % 4) * 1024),
            "global_load_dwordx2 %s, v169, %%[sj%d] offset:%d" % (v2(AIX(d)), d // 8, (d % 8) * 512)]


def a_fetch(d):
    return ["v_accvgpr_read_b32 v%d, a%d" % (AL + i, 4 * d + i) for i in range(4)]


def agpr_block(D):
    t = []
    for d in range(D):
        cs, ns = ("X", "Y") if d % 2 == 0 else ("Y", "X")
        t += ["s_waitcnt lgkmcnt(2)", "v_fma_f64 %s, -%s, %s, %s" % (v2(A0), v2(AL), v2(A0), v2(B0)),
              "s_waitcnt lgkmcnt(0)", "v_fma_f64 %s, -%s, %s, %s" % (v2(A1), v2(AL + 2), v2(A1), v2(B1)),
              "ds_write_b64 %s, %s" % (g.T0(cs), v2(A0)), "ds_write_b64 %s, %s" % (g.T1(cs), v2(A1))]
        t += g.reads((A0, B0, A1, B1), ns)
        t += ["s_waitcnt vmcnt(%d)" % (2 * (D - 3))] + a_fetch((d + 1) % D) + a_unpack((d + 2) % D, cs) + a_loads(d)
    return t


def agpr_head(D):
    t = []
    for d in range(D):
        t += a_loads(d)
    return (t + ["s_waitcnt vmcnt(%d)" % (2 * (D - 1))] + a_fetch(0) + a_unpack(0, "X") + g.reads((A0, B0, A1, B1), "X") +
            ["s_waitcnt vmcnt(%d)" % (2 * (D - 2))] + a_unpack(1, "Y"))


VARIANTS = {
    "n17": (g.head(True), g.narrow_block(), "packed"),
    "n17_nounpack": (g.head(True), drop(g.narrow_block(), "unpack"), "packed"),
    "n17_noloads": (g.head(True), drop(g.narrow_block(), "loads"), "packed"),
    "n17_nolds": (g.head(True), drop(g.narrow_block(), "lds"), "packed"),
    "n17_nofma": (g.head(True), drop(g.narrow_block(), "fma"), "packed"),
    "w17": (g.head(True), g.wide_block(), "packed"),
    "d13": (direct_head(), direct_block(False), "direct"),
    "d12": (direct_head(), direct_block(True), "direct"),
    "d13_noloads": (direct_head(), drop(direct_block(False), "loads"), "direct"),
    "d13_nolds": (direct_head(), drop(direct_block(False), "lds"), "direct"),
    "dw13": (direct_wide_head(), direct_wide_block(), "direct"),
    "dw13_noloads": (direct_wide_head(), drop(direct_wide_block(), "loads"), "direct"),
    "w17_noloads": (g.head(True), drop(g.wide_block(), "loads"), "packed"),
    "a8": (agpr_head(8), agpr_block(8), "packed", 8),
    "a16": (agpr_head(16), agpr_block(16), "packed", 16),
    "a24": (agpr_head(24), agpr_block(24), "packed", 24),
    "a16_nolds": (agpr_head(16), drop(agpr_block(16), "lds"), "packed", 16),
    "a24_nolds": (agpr_head(24), drop(agpr_block(24), "lds"), "packed", 24),
}

if __name__ == "__main__":
    print("// GENERATED by scripts/ubench/gen_sweep_units.py -- do not edit")
    print("#pragma once")
    for name, var in VARIANTS.items():
        h, b = var[0], var[1]
        g.emit("SWU_%s_HEAD" % name, [l.replace("%[si]", "%[sj0]") for l in h])
        g.emit("SWU_%s_BLOCK" % name, [l.replace("%[si]", "%[sj0]") for l in b])
        print("// %s: %d instructions per block of %d units" % (name, len(b), var[3] if len(var) > 3 else 8))
    print("#define SWU_CLOBBERS " + ", ".join('"v%d"' % r for r in list(range(160, 164)) + list(range(168, 171)) + list(range(176, 256))) + ", " +
          ", ".join('"a%d"' % r for r in range(96)))
    print("#define SWU_VARIANTS(X) " + " ".join("X(%s, %d, %d, %d)" % (n, 1 if v[2] == "direct" else 0, len(v[1]), v[3] if len(v) > 3 else 8)
                                                 for n, v in VARIANTS.items()))
