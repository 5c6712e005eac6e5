#!/usr/bin/env python3
"""Developer tool: turn gpurun_out/prof_<tag>/ (tools/profile_round.sh) into the committed summaries

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the bench command
  profiles/<tag>_pmc.json           per-launch means of every PMC counter for the timed kernel
  profiles/pmc_summary.json         what bench.py reads: per workload, the counters per launch + the rays of
                                    that launch, so bench.py can scale them to the launch it times

Usage: python tools/summarize_profile.py <tag> <workload> [rays_per_launch]
The timed kernel is k_render<false, ...> for the render workloads and k_trace_closest<false> for the ray
microbenchmarks.  HBM bytes are corrected as MI355X_MICROARCH.md prescribes for gfx950:
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (FETCH_SIZE tallies 128-byte requests as 64 bytes)."""
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "cornell-box"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
kernel_key = "k_trace_closest<false" if workload.startswith("s") and "rays" in workload else "k_render<false"
scene_key = "T<float>" if workload.endswith("-f32") else "T<double>"  # DSceneT<..> (K1) / RenderArgsT<..> (K3); the fp32 workloads also render one fp64 frame (their parity check)


def timed(name):
    return kernel_key in name and scene_key in name


ks = glob.glob(os.path.join(src, "trace", "**", "*_kernel_stats.csv"), recursive=True)
stats = {}
if ks:
    shutil.copy(ks[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
    for row in csv.DictReader(open(ks[0])):
        if timed(row["Name"]):
            stats = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"]), "name": row["Name"]}
# K1 workloads on large scenes also launch the timed kernel behind K4 (sorted batches, an extra of the bench): those
# dispatches come after the first k_ray_keys of the run and are left out of the workload's own figures
kt = glob.glob(os.path.join(src, "trace", "**", "*_kernel_trace.csv"), recursive=True)
if kt and stats:
    rows = list(csv.DictReader(open(kt[0])))
    first_sort = min((int(r["Dispatch_Id"]) for r in rows if "k_ray_keys" in r["Kernel_Name"]), default=None)
    if first_sort is not None:
        d = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in rows if timed(r["Kernel_Name"]) and int(r["Dispatch_Id"]) < first_sort]
        if d:
            stats = {"calls": len(d), "avg_ns": sum(d) / len(d), "name": stats["name"], "note": "launches before the first K4 sort of the run"}
pmc = {}
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*_counter_collection.csv"), recursive=True):
    per = {}
    all_rows = list(csv.DictReader(open(f)))
    first_sort = min((int(r["Dispatch_Id"]) for r in all_rows if "k_ray_keys" in r["Kernel_Name"]), default=None)
    for row in all_rows:
        if not timed(row["Kernel_Name"]):
            continue
        if first_sort is not None and int(row["Dispatch_Id"]) > first_sort:
            continue
        per.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
        per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for name, d in per.items():
        vals = list(d.values())
        pmc[name] = {"launches": len(vals), "mean_per_launch": sum(vals) / len(vals)}
json.dump({"kernel_stats": stats, "pmc": dict(sorted(pmc.items()))}, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)

# the bench JSON line of the trace pass names the workload's rays per launch
rays = float(sys.argv[3]) if len(sys.argv) > 3 else None
log = os.path.join(src, "log.txt")
if rays is None and os.path.exists(log):
    for line in open(log):
        if line.startswith("{") and '"metric"' in line:
            d = json.loads(line)
            rays = d["config"].get("rays_per_frame") or d["config"].get("rays_per_launch")
            break
m = {k: v["mean_per_launch"] for k, v in pmc.items()}
entry = {"round": tag, "kernel": stats.get("name", kernel_key), "rays_per_launch": rays,
         "kernel_ms_rocprof": round(stats.get("avg_ns", 0.0) / 1e6, 4), "counters_per_launch": m}
if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
    entry["hbm_bytes_per_launch"] = (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024
    entry["hbm_read_bytes_per_launch"] = 2 * m["FETCH_SIZE"] * 1024
    entry["hbm_write_bytes_per_launch"] = m["WRITE_SIZE"] * 1024
if "GRBM_GUI_ACTIVE" in m and "SQ_ACTIVE_INST_VALU" in m:
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_* count quad-cycles summed over all SIMDs (256 CUs x 4)
    cycles = m["GRBM_GUI_ACTIVE"] / 8.0
    entry["valu_busy_frac"] = round(4.0 * m["SQ_ACTIVE_INST_VALU"] / (1024.0 * cycles), 4)
    entry["clock_ghz"] = round(cycles / (stats["avg_ns"]), 3) if stats else None
if "SQ_WAIT_ANY" in m and "SQ_WAVE_CYCLES" in m:
    entry["wait_any_frac"] = round(m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 4)
if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
    entry["l2_hit_rate"] = round(m["TCC_HIT_sum"] / max(1.0, m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 4)
p = os.path.join(dst, "pmc_summary.json")
t = json.load(open(p)) if os.path.exists(p) else {}
t["_note"] = ("Per-launch PMC counters of the timed kernel from separate rocprofv3 --pmc passes (tools/profile_round.sh), "
              "summarised by tools/summarize_profile.py.  hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 as "
              "MI355X_MICROARCH.md prescribes for gfx950.  bench.py scales these by rays of its own launch / rays_per_launch.")
t[workload] = entry
json.dump(t, open(p, "w"), indent=1)
print(json.dumps({k: v for k, v in entry.items() if k != "counters_per_launch"}, indent=1))
print(json.dumps(m, indent=0))
