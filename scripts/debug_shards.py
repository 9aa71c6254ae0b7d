import multiprocessing as mp, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_library_shards import _reference, CASES

def rank_main(rank, world, name, case, q):
    import torch
    from edipack_amd import capi
    from edipack_amd.sharding import LibraryComm, library_sharded_sector
    capi.init(0)
    mode, bath, norb, nbath, sector, direct, exchange = case
    ho, pm, v = _reference(mode, bath, norb, nbath, sector)
    comm = LibraryComm(rank, world, shm_name=name, slot_bytes=1 << 22)
    h, first, count = library_sharded_sector(pm, sector, comm, direct=direct, exchange=exchange)
    ul = ho.dimup
    sl = slice(first * ul, (first + count) * ul)
    out = {}
    hv1 = comm.apply(h, v[sl])
    hv2 = comm.apply(h, v[sl])
    os.environ["EDIGPU_LANCZOS_EXACTBETA"] = "1"
    a, b, nd, n2 = comm.tridiag(h, v[sl], 10)
    del os.environ["EDIGPU_LANCZOS_EXACTBETA"]
    a2, b2, nd2, _ = comm.tridiag(h, v[sl], 10)
    hv3 = comm.apply(h, v[sl])
    q.put((rank, sl.start, sl.stop, hv1, hv2, hv3, a, a2, h.transpose_halo()))
    h.destroy(); comm.destroy()

if __name__ == "__main__":
    case = CASES[1]; world = 2
    ho, _, v = _reference(*case[:5])
    ref = ho.matvec(v); a_ref, b_ref, _ = ho.lanc_tridiag(v, 10)
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    ps = [ctx.Process(target=rank_main, args=(r, world, "edigpu_dbg", case, q)) for r in range(world)]
    [p.start() for p in ps]
    res = [q.get(timeout=300) for _ in range(world)]
    [p.join() for p in ps]
    print("dimup", ho.dimup, "dimdw", ho.dimdw)
    for rank, lo, hi, h1, h2, h3, a, a2, halo in sorted(res):
        r = ref[lo:hi]
        for nm, h in (("apply1", h1), ("apply2", h2), ("apply3", h3)):
            e = np.abs(h - r).reshape(-1, ho.dimup)
            print(rank, "halo", halo, nm, "max err", e.max(), "bad cols", np.nonzero(e.max(axis=0) > 1e-10)[0], "bad rows", np.nonzero(e.max(axis=1) > 1e-10)[0])
        print(rank, "exact tridiag err", np.abs(a - a_ref).max(), "fused", np.abs(a2 - a_ref).max())
