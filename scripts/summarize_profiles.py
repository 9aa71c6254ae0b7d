#!/usr/bin/env python3
"""Condense a scripts/collect_profiles.sh run into small text/JSON files (written next to the raw data;
copy them into profiles/ to commit)."""
import collections
import csv
import glob
import json
import os
import sys


def main():
    out, tag, wl = sys.argv[1], sys.argv[2], sys.argv[3]
    lines = []
    for f in glob.glob(os.path.join(out, "stats", "*", "*_kernel_stats.csv")):
        rows = list(csv.DictReader(open(f)))
        lines.append(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload {wl} --steps 50 --warmup 5 --no-cpu")
        lines.append(f"{'kernel':100s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>10s} {'pct':>6s}")
        for r in rows[:14]:
            lines.append(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['AverageNs']) / 1e3:10.2f} "
                         f"{float(r['TotalDurationNs']) / 1e6:10.3f} {float(r['Percentage']):6.2f}")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(out, "pmc_*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            key = ("panel2" if "panel2" in kn else "panel" if "panel" in kn else "rows" if "normal_rows" in kn else
                   "csr" if ("csr_rows" in kn or "sell_rows" in kn) else "direct" if "direct_rows" in kn else None)
            if key:
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    pmc = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
    lines.append("")
    lines.append(f"# rocprofv3 --pmc <one group per pass> -- python3 scripts/probe_hv.py --workload {wl} (mean per launch)")
    for k, d in pmc.items():
        for c, v in sorted(d.items()):
            lines.append(f"{k:6s} {c:24s} {v:.6g}")
    # HBM-side traffic per H*v = sum over the H*v kernels of (FETCH_SIZE + WRITE_SIZE) KiB
    # gfx950: FETCH_SIZE reports 1/2 of the bytes of 16-byte-per-lane coalesced streams (MI355X_MICROARCH.md,
    # HBM section): the row kernel stages V as one contiguous double2 stream -> doubled (98 MB = V once, as it must
    # be).  The panel kernels read 512..1024-byte row segments at a row stride: taken as is -- measured on config 2,
    # the two-column kernel (16 B per lane) at the one-column kernel's panel width reports the same FETCH_SIZE,
    # TCC_REQ and TCC_MISS as the one-column kernel (8 B per lane), and FETCH_SIZE does not move between 512- and
    # 1024-byte segments (165.6 / 162.3 / 161.7 thousand KiB at widths 64 / 112 / 128), so no halving shows for this
    # access shape; the value also sits at the physical minimum V + result minus what kernel A left in the L2s.
    # The CSR / SELL kernels load 8 bytes per lane -> taken as is.
    tr = 0.0
    for k, d in pmc.items():
        f = d.get("FETCH_SIZE", 0.0) * (2.0 if k == "rows" else 1.0)
        tr += (f + d.get("WRITE_SIZE", 0.0)) * 1024.0
    lines.append("")
    lines.append(f"HBM/fabric bytes per H*v (FETCH_SIZE+WRITE_SIZE, KiB->B, summed over the H*v kernels): {tr:.4g}")
    open(os.path.join(out, f"{tag}_{wl}_rocprof_summary.txt"), "w").write("\n".join(lines) + "\n")
    json.dump({wl: {"hbm_bytes_per_launch": tr, "per_kernel": pmc}}, open(os.path.join(out, f"{tag}_{wl}_pmc.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
