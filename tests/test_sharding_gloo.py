"""CPU tests of the multi-GPU host logic with world_size=2 over gloo: the shard plan, the
all-gather exchange with padded equal chunks, the local/remote split of H*v and the sharded
three-term recurrence (edipack_amd/sharding.py) reproduce the serial oracle.

The per-shard products are computed here with scipy on the ORACLE's matrices (test
infrastructure standing in for the HIP kernels, which need a GPU); what is under test is the
distributed data flow that replaces spMatVec_mpi_* / MPI_Allgatherv in the reference."""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from edipack_amd.sharding import ShardedLanczos, ShardPlan
from tests.common import make_models


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_plan_covers_everything():
    for units, ulen, world in [(3432, 3432, 8), (20, 15, 2), (5, 7, 8), (924, 1, 3), (1, 1, 4)]:
        plans = [ShardPlan(units, ulen, world, r) for r in range(world)]
        assert sum(p.count for p in plans) == units
        for r, p in enumerate(plans):
            assert p.first == min(r * p.q, units)
            assert p.row_first == p.first * ulen and p.nloc == p.count * ulen
            assert p.chunk * world >= units * ulen          # gathered buffer holds the whole vector
            assert p.counts()[r] == p.nloc


def _worker(rank, world, port, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        rng = np.random.default_rng(12345)
        if mode == "normal":
            om, _ = make_models("normal", "normal", 2, 2, seed=41)
            h = O.HNormal(om, 3, 3)
            plan = ShardPlan(h.dimdw, h.dimup, world, rank)
            hup = sp.csr_matrix((h.up[2], h.up[1], h.up[0]), shape=(h.dimup, h.dimup))
            hdw = sp.csr_matrix((h.dw[2], h.dw[1], h.dw[0]), shape=(h.dimdw, h.dimdw))
            hnd = sp.csr_matrix((h.nd[2], h.nd[1], h.nd[0]), shape=(h.dim, h.dim))
            rows = slice(plan.row_first, plan.row_first + plan.nloc)
            v0 = rng.standard_normal(h.dim)
            dtype = torch.float64

            def apply_local(vchunk, out):      # diagonal + up part: shard-local data only
                v = vchunk.numpy()[:plan.nloc].reshape(plan.count, h.dimup)
                res = h.hd[rows].reshape(plan.count, h.dimup) * v + (hup @ v.T).T
                out.numpy()[:plan.nloc] = res.reshape(-1)

            def apply_remote(vfull, out):      # down part + Hnd: need the gathered vector
                vf = vfull.numpy()[:h.dim]
                V = vf.reshape(h.dimdw, h.dimup)
                res = (hdw[plan.first:plan.first + plan.count] @ V).reshape(-1) + hnd[rows] @ vf
                out.numpy()[:plan.nloc] += res
        else:
            om, _ = make_models("superc", "normal", 2, 2, seed=42)
            h = O.HFlat(om, 0)
            plan = ShardPlan(h.dim, 1, world, rank)
            hm = sp.csr_matrix((h.csr[2], h.csr[1], h.csr[0]), shape=(h.dim, h.dim))
            lo, hi = plan.first, plan.first + plan.count
            hloc = hm[lo:hi, lo:hi]
            hnon = hm[lo:hi].tolil()
            hnon[:, lo:hi] = 0
            hnon = hnon.tocsr()
            v0 = rng.standard_normal(h.dim) + 1j * rng.standard_normal(h.dim)
            dtype = torch.complex128

            def apply_local(vchunk, out):      # `loc` block
                out.numpy()[:plan.nloc] = hloc @ vchunk.numpy()[:plan.nloc]

            def apply_remote(vfull, out):      # non-local block after the all-gather
                out.numpy()[:plan.nloc] += hnon @ vfull.numpy()[:h.dim]
        lz = ShardedLanczos(plan, apply_local, apply_remote, dtype=dtype, device="cpu")
        a, b, n = lz.tridiag(torch.from_numpy(v0[plan.row_first:plan.row_first + plan.nloc].copy()), 40)
        a_ref, b_ref, n_ref = h.lanc_tridiag(v0, 40)
        err = max(np.max(np.abs(a[:12] - a_ref[:12])) / np.max(np.abs(a_ref)),
                  np.max(np.abs(b[:12] - b_ref[:12])) / np.max(np.abs(b_ref)))
        q.put((rank, float(err), int(n), int(n_ref)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["normal", "superc"])
def test_sharded_lanczos_world2_gloo(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, n, n_ref in res:
        assert n == n_ref == 40
        assert err < 1e-10, f"rank {rank}: alpha/beta deviate from the serial oracle by {err}"
