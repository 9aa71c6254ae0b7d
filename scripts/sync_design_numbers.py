#!/usr/bin/env python3
"""Write the measurement table of DESIGN.md section 6 (between the BENCH TABLE markers) from the bench lines in
profiles/r04_bench_*.json, so that the text always quotes the committed evidence (round-3 figures in brackets from
profiles/r03_bench_*.json).   python scripts/sync_design_numbers.py   [--check: exit 1 when the table is stale]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
TAG, PREV = "r04", "r03"
ROWS = [("cfg1", "cfg1 normal 1orb Nbath4 (2,3)"), ("cfg2", "cfg2 normal 2orb Nbath6 (7,7)"),
        ("cfg2_handover", "cfg2 through `edigpu_normal_create`"), ("cfg3", "cfg3 normal 3orb hybrid 8 (5,6)"),
        ("cfg3_ns15", "cfg3 ladder Ns=15"), ("cfg3_ns16", "cfg3 ladder Ns=16"), ("cfg3_ns17", "cfg3 ladder Ns=17"),
        ("cfg3_replica_ns15", "Ns=15 with a replica bath (3 orb x 4 replicas)"),
        ("cfg4", "cfg4 superc 2orb hybrid 8 Sz=0"), ("cfg4_ns12", "cfg4 ladder Ns=12"),
        ("cfg5", "cfg5 nonsu2 3orb hybrid 10 N=13, on the fly"), ("cfg5_stored", "cfg5 the same, stored"),
        ("cfg5_stored_ns11", "cfg5 structure, stored, Ns=11")]
BEGIN, END = "<!-- BENCH TABLE BEGIN (scripts/sync_design_numbers.py) -->", "<!-- BENCH TABLE END -->"


def sp(x):
    return f"{x:,.0f}".replace(",", " ")


def load(tag, wl):
    p = f"profiles/{tag}_bench_{wl}.json"
    return json.load(open(p)) if os.path.exists(p) and os.path.getsize(p) else None


def fmt(x, f):
    return "—" if x is None else format(x, f)


def table():
    out = ["| workload | Dim | it/s (r3) | H·v ms (r3) | GB/s alg. | frac | counter traffic per H·v | CPU H·v/s (cores) | CPU 1 thread |",
           "|---|---|---|---|---|---|---|---|---|"]
    for wl, name in ROWS:
        d, p = load(TAG, wl), load(PREV, wl)
        if d is None:
            continue
        r, c = d["roofline"], d.get("cpu_baseline") or {}
        dim = int(d["config"]["workload"].split("Dim=")[1].split(" ")[0])
        pv = f" ({p['value']:.0f})" if p else ""
        ph = f" ({p['roofline']['ms_per_launch']:.3f})" if p else ""
        tr = f"{r['traffic'] / 1e6:.0f} MB = {r['traffic_frac']:.2f}" if r.get("traffic") else "—"
        frac = r["frac"] if r.get("frac") is not None else None
        note = "" if r.get("frac_reference_format") is None else f" (ref. format {r['frac_reference_format']:.2f})"
        cpu = f"{c['value']:.3g} ({c['cores']})" if c.get("value") else "—"
        c1 = f"{c['single_thread_value']:.3g}" if c.get("single_thread_value") else "—"
        out.append(f"| {name} | {sp(dim)} | {d['value']:.0f}{pv} | {r['ms_per_launch']:.3f}{ph} | {r['achieved']:.0f} | "
                   f"{fmt(frac, '.2f')}{note} | {tr} | {cpu} | {c1} |")
    d2 = load(TAG, "cfg2")
    if d2 and d2["config"].get("hbm_resident"):
        h = d2["config"]["hbm_resident"]
        out.append("")
        out.append(f"`bench.py`'s default line (config 2) also carries the HBM-resident probe `config.hbm_resident`: Ns = 16, "
                   f"H·v {h['ms_hv']:.3f} ms = {h['achieved']:.0f} GB/s = **{h['frac']:.2f}**; the boundary product on the "
                   f"reference's layout (generic kernels) {h['ms_hv_reference_layout']:.3f} ms.")
    return "\n".join(out)


def main():
    s = open("DESIGN.md").read()
    a, b = s.index(BEGIN) + len(BEGIN), s.index(END)
    new = s[:a] + "\n" + table() + "\n" + s[b:]
    if "--check" in sys.argv:
        print("stale" if new != s else "ok")
        sys.exit(1 if new != s else 0)
    open("DESIGN.md", "w").write(new)
    print(table())


if __name__ == "__main__":
    main()
