#!/usr/bin/env python3
"""Concurrent sectors: K host threads, each with its own sector handle (own HIP stream), run device-resident
tridiagonalisations at the same time -- the shape of a DMFT Green's-function step (2 * Norb * Nspin independent
Lanczos runs on small sectors).  Reports the aggregate Lanczos iterations/s against one sector at a time.

    python scripts/probe_concurrent.py [--workload cfg3] [--threads 1,2,4,8] [--nlanc 300]
"""
import argparse
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--threads", default="1,2,4,8")
    ap.add_argument("--nlanc", type=int, default=300)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import numpy as np
    import torch  # noqa: F401
    from edipack_amd import capi
    from edipack_amd.synthetic import WORKLOADS, build_workload
    capi.init(0)
    w = WORKLOADS[args.workload]
    kmax = max(int(x) for x in args.threads.split(","))
    hs = [build_workload(w) for _ in range(kmax)]
    rng = np.random.default_rng(1)
    v = rng.standard_normal(hs[0].dim)
    if hs[0].is_complex:
        v = v + 1j * rng.standard_normal(hs[0].dim)
    ref = hs[0].lanczos_tridiag(v, args.nlanc)
    out = {"workload": w.name, "dim": hs[0].dim, "nlanc": args.nlanc}
    for k in [int(x) for x in args.threads.split(",")]:
        res = [None] * k

        def work(i):
            capi.init(0)
            for _ in range(args.reps):
                res[i] = hs[i].lanczos_tridiag(v, args.nlanc)

        ths = [threading.Thread(target=work, args=(i,)) for i in range(k)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        for r in res:   # every thread got the same coefficients as the single-threaded run
            assert np.array_equal(r[0], ref[0]) and np.array_equal(r[1], ref[1])
        out[f"threads={k}"] = {"aggregate_it_per_s": round(k * args.reps * args.nlanc / dt, 1),
                               "wall_ms_per_run": round(dt * 1e3 / args.reps, 3)}
    for h in hs:
        h.destroy()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
