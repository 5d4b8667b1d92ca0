"""gpurun_out/prof_<tag>_<name>/ (written by scripts/profile_r3.sh / profile_r4.sh on the GPU box) -> profiles/<tag>_<name>/
(python scripts/summarize_profiles_r3.py [tag], default r3):
kernel_stats.csv (rocprofv3's own --stats summary) and summary.json: per product kernel the per-dispatch means of its
duration, of FETCH_SIZE / WRITE_SIZE and of the SQ / TCC counters (every counter group from its own pass), the
launch resources rocprofv3 reports (VGPRs, LDS, grid) and a few derived ratios.  The keys bench.py reads for
`roofline.traffic` (FETCH_SIZE / WRITE_SIZE summed over the kernels of a launch) are kept as in round 2."""
import csv, json, os, shutil, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DOMINANT = {"mpc": "qp_sparse_kernel", "mpc_phases": "qp_sparse_kernel", "qp_dense": "qp_dense4_iterate_kernel", "ekf": "ekf_",
            "dense_mid": "qp_dense"}


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("sfb::", "")
    return n.split("(")[0]


def main():
    from bench import source_hash
    tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
    for wl in ("mpc", "mpc_phases", "qp_dense", "dense_mid", "ekf"):
        src = os.path.join(ROOT, "gpurun_out", "prof_%s_%s" % (tag, wl))
        if not os.path.isdir(src):
            continue
        dst = os.path.join(ROOT, "profiles", "%s_%s" % (tag, wl))
        os.makedirs(dst, exist_ok=True)
        shutil.copy(os.path.join(src, "trace", "t_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
        out = {"source_hash": source_hash(), "kernel_stats": [], "kernels": {}}
        with open(os.path.join(src, "trace", "t_kernel_stats.csv")) as f:
            for r in csv.DictReader(f):
                out["kernel_stats"].append({k: r[k] for k in ("Name", "Calls", "AverageNs", "Percentage")})
        # durations per dispatch of the product kernels (in launch order) from the kernel trace
        tr = os.path.join(src, "trace", "t_kernel_trace.csv")
        durs = collections.defaultdict(list)
        if os.path.exists(tr):
            with open(tr) as f:
                for r in csv.DictReader(f):
                    if "sfb::" in r["Kernel_Name"]:
                        durs[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
        for k, v in durs.items():
            v.sort()
            out["kernels"].setdefault(k, {})["duration_ms_per_dispatch"] = [round(d, 4) for _, d in v]
        launch_tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
        launch_n = {"FETCH_SIZE": set(), "WRITE_SIZE": set()}
        rank_n = {"FETCH_SIZE": set(), "WRITE_SIZE": set()}
        for i in range(1, 10):
            path = os.path.join(src, "pmc%d" % i, "p_counter_collection.csv")
            if not os.path.exists(path):
                continue
            per = collections.defaultdict(lambda: collections.defaultdict(list))  # kernel -> counter -> [(dispatch, value)]
            with open(path) as f:
                for r in csv.DictReader(f):
                    if "sfb::" not in r["Kernel_Name"]:
                        continue
                    k = short(r["Kernel_Name"])
                    per[k][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
                    res = out["kernels"].setdefault(k, {}).setdefault("launch", {})
                    res.update(vgprs=int(r["VGPR_Count"]), accum_vgprs=int(r["Accum_VGPR_Count"]), sgprs=int(r["SGPR_Count"]),
                               lds_bytes_per_block=int(r["LDS_Block_Size"]), scratch_bytes=int(r["Scratch_Size"]),
                               workgroup=int(r["Workgroup_Size"]))
                    res.setdefault("grids", [])
                    if int(r["Grid_Size"]) not in res["grids"]:
                        res["grids"].append(int(r["Grid_Size"]))
                    if r["Counter_Name"] in launch_tot:
                        launch_tot[r["Counter_Name"]] += float(r["Counter_Value"])
                        if DOMINANT[wl] in r["Kernel_Name"]:
                            launch_n[r["Counter_Name"]].add(r["Dispatch_Id"])
                        if "sp_rank_kernel" in r["Kernel_Name"]:
                            rank_n[r["Counter_Name"]].add(r["Dispatch_Id"])
            for k, cs in per.items():
                for c, vals in cs.items():
                    vals.sort()
                    ent = out["kernels"].setdefault(k, {}).setdefault("counters", {})
                    ent[c] = {"dispatches": len(vals), "mean_per_dispatch": sum(v for _, v in vals) / len(vals)}
                    if wl == "mpc_phases":  # three dispatches per launch: setup / ADMM / polish + report
                        ph = [[], [], []]
                        for j, (_, v) in enumerate(vals):
                            ph[j % 3].append(v)
                        ent[c]["mean_per_phase"] = {n: (sum(p) / len(p) if p else None) for n, p in zip(("setup", "admm", "finish"), ph)}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            if launch_n[c]:
                n = len(launch_n[c]) // (3 if wl == "mpc_phases" else 1)
                if wl == "mpc" and rank_n[c]:  # launch in predicted order: several sparse dispatches and ONE rank kernel per launch
                    n = len(rank_n[c])
                out[c] = {"dispatches": n, "mean_per_dispatch_KB": launch_tot[c] / max(1, n)}
        for k, e in out["kernels"].items():  # derived ratios
            c = {n: v["mean_per_dispatch"] for n, v in e.get("counters", {}).items()}
            d = {}
            if c.get("SQ_WAVE_CYCLES"):
                if "SQ_WAIT_INST_ANY" in c: d["wave_cycles_waiting_frac"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
                if "SQ_WAIT_INST_LDS" in c: d["wave_cycles_waiting_on_lds_frac"] = c["SQ_WAIT_INST_LDS"] / c["SQ_WAVE_CYCLES"]
                if "SQ_ACTIVE_INST_VALU" in c: d["valu_active_over_wave_cycles"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]
            if c.get("SQ_BUSY_CYCLES") and "SQ_ACTIVE_INST_VALU" in c:
                d["valu_active_over_sq_busy_cycles"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_BUSY_CYCLES"]
            if c.get("SQ_INSTS_LDS") and "SQ_LDS_BANK_CONFLICT" in c:
                d["lds_bank_conflict_cycles_per_lds_inst"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_INSTS_LDS"]
            if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
                d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
            if c.get("SQ_WAVES") and "SQ_INSTS_VALU" in c:
                d["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
            if d:
                e["derived"] = d
        for line in open(os.path.join(src, "trace.log")):
            if line.startswith('{"metric"'):
                out["bench_line_under_profiler"] = json.loads(line)
        with open(os.path.join(dst, "summary.json"), "w") as f:
            json.dump(out, f, indent=1)
        print(dst)
        for k, e in out["kernels"].items():
            print("   ", k[:60], e.get("launch", {}), {a: round(b, 4) for a, b in e.get("derived", {}).items()})


if __name__ == "__main__":
    main()
