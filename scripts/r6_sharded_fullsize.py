"""Full-size check of the sharded recurrence (round 4, late): W ranks share the one GPU through the library's shared-memory
transport and tridiagonalise a BASELINE-size sector; every rank's coefficients are compared with the single-GPU loop of
the same handle (which the parity suite pins on the oracle at the sizes the oracle reaches).  Usage:
    python scripts/r6_sharded_fullsize.py <workload> <world> [nlanc]      (env switches as for bench.py)
Prints one line per run; exit code 1 on a mismatch."""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rank_main(rank, world, name, wl, nlanc, q):
    try:
        import torch  # noqa: F401
        from edipack_amd import capi
        from edipack_amd.sharding import LibraryComm, library_sharded_sector
        from edipack_amd.synthetic import WORKLOADS, synthetic_model
        capi.init(0)
        w = WORKLOADS[wl]
        model = synthetic_model(w)
        comm = LibraryComm(rank, world, shm_name=name, slot_bytes=1 << 30)
        h, first, count = library_sharded_sector(model, w.sector, comm)
        info = comm.shard_info(h)
        v = np.random.default_rng(5).standard_normal(h.dim)
        dim_up = h.dim_up
        sl = v[first * dim_up:(first + count) * dim_up]
        a, b, nd, n2 = comm.tridiag(h, sl, nlanc)
        ref = None
        if rank == 0:
            ra, rb, _ = h.lanczos_tridiag(v, nlanc)
            ref = (ra, rb)
        h.destroy()
        comm.destroy()
        q.put((rank, info, a, b, nd, ref, None))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, None, None, None, 0, None, traceback.format_exc() + str(e)))


def main():
    wl, world = sys.argv[1], int(sys.argv[2])
    nlanc = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = f"edigpu_full_{os.getpid()}_{world}"
    procs = [ctx.Process(target=rank_main, args=(r, world, name, wl, nlanc, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    bad = [r[6] for r in res if r[6]]
    if bad:
        print(f"{wl} world={world}: FAILED\n{bad[0]}")
        return 1
    ref = [r[5] for r in res if r[5] is not None][0]
    worst = 0.0
    for r in res:
        ea = np.max(np.abs(r[2] - ref[0])) / np.max(np.abs(ref[0]))
        eb = np.max(np.abs(r[3] - ref[1])) / np.max(np.abs(ref[1]))
        worst = max(worst, ea, eb)
    kind = {0: "all-gather", 1: "column blocks", 2: "padded panels"}[res[0][1][0]]
    loop = os.environ.get("EDIGPU_SHARD_PANEL_LOOP", "1")
    ok = worst < 1e-10
    print(f"{wl} world={world} exchange={kind} panel_loop={loop} nlanc={nlanc}: max rel dev of alpha/beta from the single-GPU loop "
          f"{worst:.2e} {'OK' if ok else 'MISMATCH'}")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
