"""Collect the reference's own regression fixtures for the H*v path into one small JSON.

Reads the plain-text *.check DATA files of /root/reference/test/src/<BATH>_<MODE>/ (numbers only,
no source) plus the handful of input values that define each test (inputED.in) and writes
tests/golden/reference_checks.json.  Run in the build container only (the reference tree does
not exist on the GPU box):  python tests/golden/make_reference_checks.py
"""
import json
import os
import re

REF = "/root/reference/test/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_checks.json")
KEYS = ["NORB", "NBATH", "NSPIN", "BATH_TYPE", "ED_MODE", "ULOC", "UST", "JH", "JX", "JP", "XMU", "HFMODE",
        "ED_HW_BATH", "DELTASC", "DELTA", "MH", "LAMBDA", "BETA", "LANC_NGFITER", "LANC_DIM_THRESHOLD"]


def fnum(s):
    return float(s.lower().replace("d", "e"))


def read_input(path):
    out = {}
    for line in open(path):
        m = re.match(r"\s*([A-Z_0-9]+)\s*=\s*([^!]*)", line)
        if not m or m.group(1) not in KEYS:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k in ("BATH_TYPE", "ED_MODE"):
            out[k] = v
        elif k == "HFMODE":
            out[k] = v.upper().startswith("T")
        elif k == "ULOC":
            out[k] = [fnum(x) for x in v.split(",") if x.strip()]
        else:
            out[k] = fnum(v)
    return out


def read_check(path):
    """numbers as written by SciFortran save_array: reals, or (re,im) complex pairs"""
    txt = open(path).read()
    if "(" in txt:
        return [[fnum(a), fnum(b)] for a, b in re.findall(r"\(\s*([^,\s]+)\s*,\s*([^)\s]+)\s*\)", txt)]
    return [fnum(x) for x in txt.split()]


def main():
    res = {}
    for d in sorted(os.listdir(REF)):
        full = os.path.join(REF, d)
        if not os.path.isdir(full) or not os.path.exists(os.path.join(full, "evals.check")):
            continue
        entry = {"input": read_input(os.path.join(full, "inputED.in"))}
        for f in sorted(os.listdir(full)):
            if f.endswith(".check"):
                entry[f[:-6]] = read_check(os.path.join(full, f))
        res[d] = entry
    json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)
    print("wrote", OUT, "with", len(res), "test dirs")


if __name__ == "__main__":
    main()
