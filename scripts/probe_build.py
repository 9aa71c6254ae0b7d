#!/usr/bin/env python3
"""Build-time probe: builds a workload's sector several times (first build includes one-off HIP start-up)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--n", type=int, default=3)
    a = ap.parse_args()
    import torch  # noqa: F401
    from edipack_amd import capi
    from edipack_amd.synthetic import WORKLOADS, build_workload
    capi.init(0)
    w = WORKLOADS[a.workload]
    for i in range(a.n):
        t0 = time.perf_counter()
        h = build_workload(w)
        t1 = time.perf_counter()
        ms = h.time_apply(1, 3)
        t2 = time.perf_counter()
        h.destroy()
        print(f"{a.workload}: build {i}: {t1 - t0:.4f} s (dim={h.dim}); first products {t2 - t1:.4f} s, H*v {ms:.4f} ms")


if __name__ == "__main__":
    main()
