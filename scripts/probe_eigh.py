#!/usr/bin/env python3
"""Wall time of the eigensolvers on a workload's largest sector: edigpu_lanczos_eigh (lowest pair, plain recurrence)
and edigpu_lanczos_eigh_multi (thick restart, ARPACK-equivalent) for 1, 2, 4 eigenpairs.

    python scripts/probe_eigh.py [--workload cfg2] [--tol 1e-10]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--tol", type=float, default=1e-10)
    ap.add_argument("--ncv", type=int, default=24)
    args = ap.parse_args()
    import torch  # noqa: F401
    from edipack_amd import capi
    from edipack_amd.synthetic import WORKLOADS, build_workload
    capi.init(0)
    w = WORKLOADS[args.workload]
    h = build_workload(w)
    out = {"workload": w.name, "dim": h.dim, "tol": args.tol, "ncv": args.ncv}
    h.lanczos_eigh(nitermax=20, tol=1e-3, want_vector=False)     # warm-up
    t0 = time.perf_counter()
    e0, _, nit = h.lanczos_eigh(nitermax=300, tol=args.tol, want_vector=False)
    out["eigh"] = {"e0": e0, "iterations": nit, "seconds": round(time.perf_counter() - t0, 4)}
    # the same with the Ritz vector left on the device (the form the Green's-function step consumes)
    import ctypes as C
    vec = C.c_void_p()
    capi.check(capi.lib().edigpu_dev_alloc(8 * h.dim * (2 if h.is_complex else 1), C.byref(vec)))
    ev, nd = C.c_double(), C.c_int(0)
    t0 = time.perf_counter()
    capi.check(capi.lib().edigpu_lanczos_eigh(h._h, 300, args.tol, 10, None, C.byref(ev),
                                                 C.cast(vec, C.POINTER(C.c_double)), C.byref(nd)))
    out["eigh_with_vector"] = {"e0": ev.value, "iterations": nd.value, "seconds": round(time.perf_counter() - t0, 4)}
    capi.check(capi.lib().edigpu_dev_free(vec))
    for ne in (1, 2, 4):
        t0 = time.perf_counter()
        ev, _, nconv, nmv = h.lanczos_eigh_multi(ne, ncv=args.ncv, tol=args.tol, want_vectors=False)
        out[f"eigh_multi_{ne}"] = {"evals": [float(x) for x in ev], "nconv": nconv, "matvecs": nmv,
                                   "seconds": round(time.perf_counter() - t0, 4)}
    h.destroy()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
