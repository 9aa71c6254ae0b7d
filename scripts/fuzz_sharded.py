#!/usr/bin/env python3
"""Randomized sweep of the N > 1 recurrence on padded panels (not part of the test suite): random normal-mode models and
sectors with the block image forced, ONE communicator for all of them (RCCL world of one with the collectives forced, so
every exchange buffer is really written and reused across geometries), the sharded product and tridiagonalisation against
the CPU oracle.   python scripts/fuzz_sharded.py --trials 300"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=300)
    ap.add_argument("--seed", type=int, default=4242)
    args = ap.parse_args()
    os.environ.update(EDIGPU_FORCE_COLLECTIVES="1", EDIGPU_IB="1", EDIGPU_IB_MIN="0")
    import numpy as np
    import torch  # noqa: F401
    from edipack_amd import capi
    from edipack_amd.sharding import LibraryComm, library_sharded_sector
    from oracle import oracle as O
    from tests.common import make_models, rel_err
    capi.init(0)
    rng = np.random.default_rng(args.seed)
    comm = LibraryComm(0, 1, unique_id=LibraryComm.unique_id())
    t0, worst, nblock, nother = time.time(), 0.0, 0, 0
    for trial in range(args.trials):
        norb = int(rng.integers(1, 4))
        bath = ["normal", "hybrid"][int(rng.integers(0, 2))]
        nbath = int(rng.integers(2, 6 if norb == 1 else (5 if norb == 2 else 4)))
        om, pm = make_models("normal", bath, norb, nbath, seed=int(rng.integers(0, 1 << 30)))
        ns = om.ns
        nup, ndw = int(rng.integers(1, ns)), int(rng.integers(1, ns))
        os.environ["EDIGPU_IB_ROWS"] = str(int(rng.choice([8, 16, 24, 64, 480])))
        ho = O.HNormal(om, nup, ndw)
        if ho.dim < 40 or ho.dim > 400000:
            continue
        v = rng.standard_normal(ho.dim)
        h, first, count = library_sharded_sector(pm, (nup, ndw), comm)
        kind = comm.shard_info(h)[0]
        nl = int(min(12, ho.dim - 1))
        e = rel_err(comm.apply(h, v), ho.matvec(v))
        a, b, nd, _ = comm.tridiag(h, v, nl)
        ar, br, _ = ho.lanc_tridiag(v, nl)
        k = min(nd, nl)
        # (a breakdown inside the first 12 steps ends both recurrences at the same step; compare what both computed)
        et = max(rel_err(a[:k], ar[:k]), rel_err(b[:k], br[:k])) if k > 0 else 0.0
        h.destroy()
        if kind == 2:
            nblock += 1
        else:
            nother += 1
        worst = max(worst, e, et)
        if e > 1e-11 or et > 1e-8:
            print(f"MISMATCH trial {trial}: norb={norb} bath={bath} nbath={nbath} sector=({nup},{ndw}) dim={ho.dim} kind={kind} "
                  f"rows={os.environ['EDIGPU_IB_ROWS']} H*v {e:.2e} tridiag {et:.2e} nd={nd}")
            return 1
        if (trial + 1) % 50 == 0:
            print(f"trial {trial + 1}: {nblock} on padded panels, {nother} on column blocks / whole, worst rel err {worst:.2e}, "
                  f"{time.time() - t0:.0f} s", flush=True)
    comm.destroy()
    print(f"sharded fuzz ok: {nblock} sectors on padded panels, {nother} others, worst rel err {worst:.2e}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
