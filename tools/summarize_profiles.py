#!/usr/bin/env python3
"""Copy the judged summaries of the last `tools/profile_round.sh TAG` run from gpurun_out/prof_TAG/ into profiles/.
The counter summary is stamped with the hash of the csrc sources the profiled library was built from (written on the GPU
box by profile_round.sh), which bench.py compares with the running library (roofline.counters_stale)."""
import collections, csv, glob, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
name = sys.argv[2] if len(sys.argv) > 2 else "bench"     # profiles/TAG_NAME_*: `bench` = the headline command
src, out = f"gpurun_out/prof_{tag}" if name == "bench" else f"gpurun_out/prof_{tag}_{name}", "profiles"
newest = lambda pat: sorted(glob.glob(pat), key=os.path.getmtime)[-1]
shutil.copy(newest(f"{src}/stats/*/*_kernel_stats.csv"), f"{out}/{tag}_{name}_kernel_stats.csv")
rows = []
for d in sorted(os.listdir(src)):
    if not d.startswith("pmc") or not os.path.isdir(f"{src}/{d}"):
        continue
    agg, meta = collections.defaultdict(list), {}
    for row in csv.DictReader(open(newest(f"{src}/{d}/*/*_counter_collection.csv"))):
        kname = row["Kernel_Name"]
        if "render_kernel" in kname or "resolve_kernel" in kname or "primary_rays" in kname:
            k = (kname.split("(")[0].replace("void ", ""), row["Counter_Name"])
            agg[k].append(float(row["Counter_Value"]))
            meta[k] = (row["Grid_Size"], row["Workgroup_Size"], row["LDS_Block_Size"], row["Scratch_Size"], row["VGPR_Count"], row["SGPR_Count"])
    for k, v in sorted(agg.items()):
        rows.append([k[0], k[1], len(v), f"{sum(v)/len(v):.6g}", *meta[k]])
with open(f"{out}/{tag}_{name}_pmc_summary.csv", "w") as fo:
    fo.write(f"# csrc_sha256: {open(f'{src}/csrc_sha256.txt').read().strip()}\n")
    if os.path.exists(f"{src}/config.txt"):   # workload W H spp depth the counters were taken at (bench.py keys its lookup on it)
        fo.write(f"# config: {open(f'{src}/config.txt').read().strip()}\n")
    w = csv.writer(fo)
    w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch", "grid", "workgroup", "lds_bytes", "scratch", "vgpr", "sgpr"])
    w.writerows(rows)
if os.path.exists(f"{src}/bench.json"):
    shutil.copy(f"{src}/bench.json", f"{out}/{tag}_{name}.json")
print("wrote", out)
