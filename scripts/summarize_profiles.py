"""gpurun_out/prof_<tag>_<wl>/ (written by scripts/profile_all.sh on the GPU box) -> profiles/<tag>_<wl>/:
kernel_stats.csv (rocprofv3's own --stats summary) and summary.json (per-dispatch means of the product
kernel's duration and of the FETCH_SIZE / WRITE_SIZE counters, each from its own pass)."""
import csv, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# product kernels; a launch of the dense path is three kernels (setup / iterate / finish): counters are summed
# over all of them and divided by the number of launches (= dispatches of the dominant kernel)
KERNELS = ("qp_sparse_kernel", "qp_dense_kernel", "qp_dense4_", "ekf_kernel")
DOMINANT = ("qp_sparse_kernel", "qp_dense_kernel", "qp_dense4_iterate_kernel", "ekf_kernel")


def main(tag):
    for wl in ("mpc", "ekf", "qp_dense"):
        src = os.path.join(ROOT, "gpurun_out", "prof_%s_%s" % (tag, wl))
        if not os.path.isdir(src):
            continue
        dst = os.path.join(ROOT, "profiles", "%s_%s" % (tag, wl))
        os.makedirs(dst, exist_ok=True)
        shutil.copy(os.path.join(src, "trace", "t_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
        from bench import source_hash
        out = {"source_hash": source_hash(), "kernel_stats": []}  # bench.py quotes the traffic only for this build
        with open(os.path.join(src, "trace", "t_kernel_stats.csv")) as f:
            for r in csv.DictReader(f):
                out["kernel_stats"].append({k: r[k] for k in ("Name", "Calls", "AverageNs", "Percentage")})
        for ctr, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
            path = os.path.join(src, sub, "p_counter_collection.csv")
            if not os.path.exists(path):
                continue
            total, launches = 0.0, set()
            with open(path) as f:
                for r in csv.DictReader(f):
                    if r["Counter_Name"] == ctr and any(k in r["Kernel_Name"] for k in KERNELS):
                        total += float(r["Counter_Value"])
                        if any(k in r["Kernel_Name"] for k in DOMINANT):
                            launches.add(r["Dispatch_Id"])
            if launches:
                out[ctr] = {"dispatches": len(launches), "mean_per_dispatch_KB": total / len(launches)}
        for line in open(os.path.join(src, "trace.log")):
            if line.startswith('{"metric"'):
                out["bench_line_under_profiler"] = json.loads(line)
        with open(os.path.join(dst, "summary.json"), "w") as f:
            json.dump(out, f, indent=1)
        print(dst, {k: v for k, v in out.items() if k in ("FETCH_SIZE", "WRITE_SIZE")},
              [(s["Name"][:40], s["AverageNs"]) for s in out["kernel_stats"][:2]])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r1")
