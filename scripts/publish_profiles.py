#!/usr/bin/env python3
"""Copy what scripts/collect_profiles.sh / refresh_evidence.sh left under gpurun_out/ into profiles/ (the tracked
evidence): rocprofv3 summaries, kernel stats, the merged counter traffic (profiles/pmc_traffic.json, stamped with the
kernel source hash by summarize_profiles.py) and the bench lines.   python scripts/publish_profiles.py r02"""
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(root)
merged = {}
for d in sorted(glob.glob(f"gpurun_out/prof_{tag}_*")):
    wl = os.path.basename(d)[len(f"prof_{tag}_"):]
    pj = f"{d}/{tag}_{wl}_pmc.json"
    if not os.path.exists(pj):
        continue
    merged.update(json.load(open(pj)))
    shutil.copy(f"{d}/{tag}_{wl}_rocprof_summary.txt", f"profiles/{tag}_{wl}_rocprof_summary.txt")
    ks = glob.glob(f"{d}/stats/*/*_kernel_stats.csv")
    if ks:
        shutil.copy(ks[0], f"profiles/{tag}_{wl}_kernel_stats.csv")
if merged:
    # workloads that were not re-collected in this call keep their entry (and its source-hash stamp: bench.py drops a
    # figure whose stamp is not the hash of the current sources)
    try:
        old = json.load(open("profiles/pmc_traffic.json"))
    except Exception:
        old = {}
    old.update(merged)
    json.dump(old, open("profiles/pmc_traffic.json", "w"), indent=1)
for f in glob.glob(f"gpurun_out/bench/{tag}_bench_*.json"):
    if os.path.getsize(f) > 0:
        shutil.copy(f, "profiles/" + os.path.basename(f))
print({k: (round(v["hbm_bytes_per_launch"] / 1e6, 1), v["source_hash"]) for k, v in merged.items()})
