#!/usr/bin/env python3
"""Rewrite the measurement rows of DESIGN.md section 6 (and the headline figures quoted elsewhere) from the bench lines
in profiles/r02_bench_*.json, so that the text always quotes the committed evidence.  python scripts/sync_design_numbers.py"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
NAMES = {"cfg2": "cfg2 normal 2orb Nbath6 (7,7)", "cfg3": "cfg3 normal 3orb hybrid 8 (5,6)", "cfg3_ns15": "cfg3 ladder Ns=15",
         "cfg3_ns16": "cfg3 ladder Ns=16", "cfg4": "cfg4 superc 2orb hybrid 8 Sz=0", "cfg4_ns12": "cfg4 ladder Ns=12",
         "cfg5": "cfg5 nonsu2 3orb hybrid 10 N=13, direct", "cfg5_stored": "cfg5 the same, stored",
         "cfg5_stored_ns11": "cfg5 structure, stored, Ns=11"}
R1 = {"cfg2": (5731, 0.130), "cfg3": (47850, 0.016), "cfg3_ns15": (682, 1.203), "cfg3_ns16": (209, 4.168), "cfg4": (44643, 0.013),
      "cfg4_ns12": (4808, 0.166), "cfg5": (565, 2.002), "cfg5_stored": (660, 1.346), "cfg5_stored_ns11": (13160, 0.063)}


def sp(x):
    return f"{x:,.0f}".replace(",", " ")


CHECK = "--check" in sys.argv     # report instead of rewriting; exit 1 when the text does not quote the bench lines
s = open("DESIGN.md").read()
s_before = s
out = []
for ln in s.split("\n"):
    hit = next((wl for wl, nm in NAMES.items() if ln.startswith(f"| {nm} |")), None)
    if hit:
        d = json.load(open(f"profiles/r02_bench_{hit}.json"))
        r, c = d["roofline"], d.get("cpu_baseline", {})
        dim = int(d["config"]["workload"].split("Dim=")[1].split(" ")[0])
        tf = f"{r['traffic'] / 1e6:.0f} MB = {r['traffic_frac']:.2f}" if r.get("traffic") else "—"
        cv = f"{c['value']:.3g}" if c.get("value") else "—"
        c1 = f"{c['single_thread_value']:.3g}" if c.get("single_thread_value") else "—"
        ln = (f"| {NAMES[hit]} | {sp(dim)} | {d['value']:.0f} ({R1[hit][0]}) | {r['ms_per_launch']:.3f} ({R1[hit][1]}) | "
              f"{r['achieved']:.0f} | {r['frac']:.2f} | {tf} | {cv} | {c1} |")
    out.append(ln)
s = "\n".join(out)
d17 = json.load(open("profiles/r02_bench_cfg3_ns17.json"))
s = re.sub(r"\| 590 976 100 \| [0-9.]+ \(27.3\) \| [0-9.]+ \(25.2\) \| [0-9]+ \| [0-9.]+ \|",
           f"| 590 976 100 | {d17['value']:.1f} (27.3) | {d17['roofline']['ms_per_launch']:.1f} (25.2) | "
           f"{d17['roofline']['achieved']:.0f} | {d17['roofline']['frac']:.2f} |", s)
d = json.load(open("profiles/r02_bench_cfg2.json"))
dh = json.load(open("profiles/r02_bench_cfg2_handover.json"))
s = re.sub(r"\*\*6 \d\d\d Lanczos iterations/s\*\*", f"**{sp(d['value'])} Lanczos iterations/s**", s)
s = re.sub(r"6 \d\d\d it/s is 8\d % of it\.", f"{sp(d['value'])} it/s is {100 * d['value'] / 7100:.0f} % of it.", s)
s = re.sub(r"the same rate \(6 \d\d\d it/s;", f"the same rate ({sp(dh['value'])} it/s;", s)
s = re.sub(r"`profiles/r02_bench_cfg2_handover.json`\): 6 \d\d\d it/s", f"`profiles/r02_bench_cfg2_handover.json`): {sp(dh['value'])} it/s", s)
t = open("INTEGRATION.md").read()
t_before = t
t = re.sub(r"Config 2 through this patch: 6 \d\d\d Lanczos it/s", f"Config 2 through this patch: {sp(dh['value'])} Lanczos it/s", t)
if CHECK:
    stale = [n for n, a, b in (("DESIGN.md", s_before, s), ("INTEGRATION.md", t_before, t)) if a != b]
    print("stale:", stale)
    sys.exit(1 if stale else 0)
open("DESIGN.md", "w").write(s)
open("INTEGRATION.md", "w").write(t)
print("cfg2", d["value"], "handover", dh["value"], "ns17", d17["value"])
