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
        lines.append(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload {wl} --steps 50 --warmup 5 --no-cpu --no-resident")
        lines.append(f"{'kernel':100s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>10s} {'pct':>6s}")
        for r in rows[:14]:
            lines.append(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['AverageNs']) / 1e3:10.2f} "
                         f"{float(r['TotalDurationNs']) / 1e6:10.3f} {float(r['Percentage']):6.2f}")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(out, "pmc_*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            # (the two launches of a row staged in halves are two kernels of one product: ib_rows_kernel<..., 1> / <..., 2>)
            key = ("sb_rows" if "sb_rows" in kn else "sb_cols" if "sb_cols" in kn else "ib_rowsB" if ("ib_rows" in kn and ", 0, 2>" in kn) else "ib_rows" if "ib_rows" in kn else "ib_cols" if "ib_cols" in kn else "tile" if "dw_tile" in kn else "panel2" if "panel2" in kn else "panel" if "panel" in kn else "rows" if "normal_rows" in kn else
                   "csr" if ("csr_rows" in kn or "sell_rows" in kn) else "direct" if "direct_rows" in kn else None)
            if key:
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    pmc = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
    lines.append("")
    lines.append(f"# rocprofv3 --pmc <one group per pass> -- python3 scripts/probe_hv.py --workload {wl} (mean per launch)")
    for k, d in pmc.items():
        for c, v in sorted(d.items()):
            lines.append(f"{k:8s} {c:24s} {v:.6g}")
    # HBM-side traffic per H*v = sum over the H*v kernels of (2 * FETCH_SIZE + WRITE_SIZE) KiB.
    # gfx950: FETCH_SIZE counts 64 bytes per 128-byte fabric read request (= TCC_EA0_RDREQ * 64), i.e. exactly HALF of
    # the bytes read, for every access shape of these kernels -- contiguous or 512 / 1024-byte row segments, 8 or 16
    # bytes per lane -- calibrated on known byte counts in profiles/r02_fetch_calibration.txt
    # (scripts/calibrate_fetch.sh); WRITE_SIZE is exact there.
    tr = 0.0
    for k, d in pmc.items():
        tr += (2.0 * d.get("FETCH_SIZE", 0.0) + d.get("WRITE_SIZE", 0.0)) * 1024.0
    lines.append("")
    lines.append(f"HBM/fabric bytes per H*v (2*FETCH_SIZE + WRITE_SIZE, KiB->B, summed over the H*v kernels): {tr:.4g}")
    open(os.path.join(out, f"{tag}_{wl}_rocprof_summary.txt"), "w").write("\n".join(lines) + "\n")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from edipack_amd import capi
    json.dump({wl: {"hbm_bytes_per_launch": tr, "source_hash": capi.kernel_source_hash(), "per_kernel": pmc}}, open(os.path.join(out, f"{tag}_{wl}_pmc.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
