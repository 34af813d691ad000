#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (tools/profile.sh) into profiles/<tag>_summary.md + the raw kernel_stats.csv."""
import collections
import csv
import os
import shutil
import sys

tag = sys.argv[1]
src = os.path.join("gpurun_out", f"prof_{tag}")
os.makedirs("profiles", exist_ok=True)
stats = list(csv.DictReader(open(os.path.join(src, "trace", "t_kernel_stats.csv"))))
shutil.copy(os.path.join(src, "trace", "t_kernel_stats.csv"), os.path.join("profiles", f"{tag}_kernel_stats.csv"))


def short(name):
    return name.split("(")[0].replace("void ", "").replace("opusfe::", "")


def pmc(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    if not os.path.exists(path):
        return agg
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            k = short(r["Kernel_Name"])
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    return agg


fetch = pmc(os.path.join(src, "pmc_fetch", "f_counter_collection.csv"), "FETCH_SIZE")
write = pmc(os.path.join(src, "pmc_write", "w_counter_collection.csv"), "WRITE_SIZE")
with open(os.path.join("profiles", f"{tag}_summary.md"), "w") as f:
    f.write(f"# rocprofv3 summary `{tag}`\n\nCommand: `tools/profile.sh {tag} {' '.join(sys.argv[2:])}` "
            "(kernel-trace --stats; FETCH_SIZE and WRITE_SIZE in separate --pmc passes).\n"
            "FETCH_SIZE / WRITE_SIZE are rocprofv3's KB figures per launch; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x "
            "and is uncalibrated for the narrow per-lane accesses of the parser / lowering kernels (MI355X_MICROARCH.md, HBM section): "
            "read them as relative evidence of over-fetch, not as absolutes.\n\n"
            "| kernel | calls | avg ms | % time | FETCH_SIZE MB/launch | WRITE_SIZE MB/launch |\n|---|---|---|---|---|---|\n")
    for r in stats[:28]:
        k = short(r["Name"])
        fs = fetch[k][1] / fetch[k][0] / 1e3 if fetch[k][0] else float("nan")
        ws = write[k][1] / write[k][0] / 1e3 if write[k][0] else float("nan")
        f.write(f"| {k} | {r['Calls']} | {float(r['AverageNs']) / 1e6:.4f} | {float(r['Percentage']):.2f} | {fs:.1f} | {ws:.1f} |\n")
# machine-readable per-launch traffic for bench.py's roofline.traffic (same workload only): bytes = FETCH_SIZE + WRITE_SIZE
# as rocprofv3 reports them (KB -> B); the 2x correction of the guide applies to wide coalesced reads only and is NOT applied
import json
import re
m = re.search(r"--bytes\s+(\S+)", " ".join(sys.argv[2:]))
import hashlib
sys.path.insert(0, os.getcwd())
try:
    import opus_amd
    sha = hashlib.sha1(open(opus_amd.library_path(), "rb").read()).hexdigest()
except Exception:
    sha = None
traffic = {"tag": tag, "source_bytes": int(float(m.group(1))) if m else 1000000000,      # (bench.py's default)
           "unit": "bytes per launch", "library_sha1": sha, "kernels": {}}
for r in stats:
    k = short(r["Name"])
    if fetch[k][0] and write[k][0]:
        name = k[2:] if k.startswith("k_") else k
        traffic["kernels"][name] = {"fetch": fetch[k][1] / fetch[k][0] * 1e3, "write": write[k][1] / write[k][0] * 1e3, "avg_ms": float(r["AverageNs"]) / 1e6}
# bench.py prices the flat lowering as ONE unit per step, "lower_flat" (all kernels of lower_par.h): traffic and time per STEP, summed
# over every launch of every one of its kernels (steps = the launches of k_flat_scan, one per step)
flat = [(k, fetch[k], write[k], r) for r in stats for k in [short(r["Name"])] if (k.startswith("par::k_flat_") or k.startswith("par::k_sched_")) and fetch[k][0] and write[k][0]]
scan = [x for x in flat if x[0].startswith("par::k_flat_scan")]
if flat and scan:
    steps_f, steps_w, steps_t = scan[0][1][0], scan[0][2][0], int(scan[0][3]["Calls"])
    traffic["kernels"]["lower_flat"] = {"fetch": sum(f[1] for _, f, _, _ in flat) / steps_f * 1e3, "write": sum(w[1] for _, _, w, _ in flat) / steps_w * 1e3,
                                        "avg_ms": sum(float(r["TotalDurationNs"]) for _, _, _, r in flat) / steps_t / 1e6, "calls": sum(int(r["Calls"]) for _, _, _, r in flat),
                                        "note": "per step: every launch of every kernel of the flat lowering"}
json.dump(traffic, open(os.path.join("profiles", f"{tag}_traffic.json"), "w"), indent=1)
print(open(os.path.join("profiles", f"{tag}_summary.md")).read())
