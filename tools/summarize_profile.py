#!/usr/bin/env python3
"""Developer tool: turn gpurun_out/prof_<tag>/ (tools/profile_round.sh) into the committed summaries
profiles/<tag>_bench_kernel_stats.csv, profiles/<tag>_pmc_k_render.json and the HBM-traffic entry that
bench.py reads (profiles/hbm_traffic.json).  Usage: python tools/summarize_profile.py <tag> [workload spp]"""
import csv, glob, json, os, shutil, sys

tag = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "cornell-box"
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 500
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
ks = glob.glob(os.path.join(src, "trace", "**", "*_kernel_stats.csv"), recursive=True)
if ks:
    shutil.copy(ks[0], os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
    print(open(ks[0]).read()[:900])
# timed launches only: the non-counting lean/full instantiation (k_render<false, ...>)
pmc = {}
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*_counter_collection.csv"), recursive=True):
    per = {}
    for row in csv.DictReader(open(f)):
        if "k_render<false" not in row["Kernel_Name"]:
            continue
        per.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
        per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for name, d in per.items():
        vals = list(d.values())
        pmc[name] = {"launches": len(vals), "mean_per_launch": sum(vals) / len(vals)}
json.dump(dict(sorted(pmc.items())), open(os.path.join(dst, f"{tag}_pmc_k_render.json"), "w"), indent=1)
print(json.dumps({k: v["mean_per_launch"] for k, v in sorted(pmc.items())}, indent=0))
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    p = os.path.join(dst, "hbm_traffic.json")
    t = json.load(open(p)) if os.path.exists(p) else {}
    fs, ws = pmc["FETCH_SIZE"]["mean_per_launch"], pmc["WRITE_SIZE"]["mean_per_launch"]
    t[workload] = {"spp": spp, "fetch_size_kib": fs, "write_size_kib": ws,
                   "hbm_bytes_per_launch": (2 * fs + ws) * 1024, "round": tag}
    t["_note"] = t.get("_note", "").split(" Source:")[0] + f" Source: profiles/{tag}_pmc_k_render.json"
    json.dump(t, open(p, "w"), indent=1)
