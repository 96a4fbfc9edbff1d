#!/usr/bin/env python3
"""Copy the judged summaries of the last gpurun profile run from gpurun_out/ into profiles/ (round tag arg)."""
import collections, csv, glob, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = "profiles"
newest = lambda pat: sorted(glob.glob(pat), key=os.path.getmtime)[-1]
shutil.copy(newest("gpurun_out/prof_r1/*/*_kernel_stats.csv"), f"{out}/{tag}_bench_kernel_stats.csv")
rows = []
for d in ["pmc_fetch_r1", "pmc_write_r1", "pmcA", "pmcB"]:
    agg, meta = collections.defaultdict(list), {}
    for row in csv.DictReader(open(newest(f"gpurun_out/{d}/*/*_counter_collection.csv"))):
        name = row["Kernel_Name"]
        if "render_kernel" in name or "resolve_kernel" in name:
            k = (name.split("(")[0].replace("void ", ""), row["Counter_Name"])
            agg[k].append(float(row["Counter_Value"]))
            meta[k] = (row["Grid_Size"], row["Workgroup_Size"], row["LDS_Block_Size"], row["Scratch_Size"], row["VGPR_Count"], row["SGPR_Count"])
    for k, v in sorted(agg.items()):
        rows.append([k[0], k[1], len(v), f"{sum(v)/len(v):.6g}", *meta[k]])
with open(f"{out}/{tag}_bench_pmc_summary.csv", "w") as fo:
    w = csv.writer(fo)
    w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch", "grid", "workgroup", "lds_bytes", "scratch", "vgpr", "sgpr"])
    w.writerows(rows)
shutil.copy("gpurun_out/bench_r1.json", f"{out}/{tag}_bench.json")
print("wrote", out)
