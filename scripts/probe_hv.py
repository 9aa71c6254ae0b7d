#!/usr/bin/env python3
"""Profiling probe: build one workload and run N plain H*v products as the device-resident Lanczos loops compute them
(the launches bench.py's roofline figure is measured on; no recurrence, no CPU leg).
Meant to sit behind `rocprofv3 ... -- python3 scripts/probe_hv.py --workload cfg2 --steps 20`."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--lanczos", action="store_true")
    ap.add_argument("--natural", action="store_true",
                    help="the boundary product (edigpu_apply_dev, reference layout) instead of the product of the Lanczos loops")
    a = ap.parse_args()
    import torch  # noqa: F401
    from edipack_amd import capi
    from edipack_amd.synthetic import WORKLOADS, build_workload
    capi.init(0)
    w = WORKLOADS[a.workload]
    h = build_workload(w)
    b_hv, b_step = h.algorithmic_bytes()
    if a.lanczos:
        ms_step, ms_hv = h.lanczos_bench(a.warmup, a.steps)
        print(f"{a.workload}: dim={h.dim} lanczos step {ms_step:.4f} ms ({b_step / ms_step / 1e6:.0f} GB/s) "
              f"H*v {ms_hv:.4f} ms ({b_hv / ms_hv / 1e6:.0f} GB/s alg)")
    else:
        ms = h.time_apply(a.warmup, a.steps, lanczos=0 if a.natural else 2)
        print(f"{a.workload}: dim={h.dim} H*v {ms:.4f} ms  {b_hv / ms / 1e6:.0f} GB/s algorithmic "
              f"({b_hv / 1e6:.1f} MB/launch)")
    h.destroy()


if __name__ == "__main__":
    main()
