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

from edipack_amd.sharding import ShardedLanczos, ShardPlan, TransposedLanczos
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


class _OracleTransposeKernels:
    """Stand-ins for the four device steps of the transposed exchange, on the ORACLE's matrices (numpy):
    what is under test is the data flow of TransposedLanczos (buffer layout, all-to-all pattern, halo)."""

    def __init__(self, h, plan):
        self.h, self.plan = h, plan
        self.dim_up, self.dim_dw = h.dimup, h.dimdw
        self.hup = sp.csr_matrix((h.up[2], h.up[1], h.up[0]), shape=(h.dimup, h.dimup))
        self.hdw = sp.csr_matrix((h.dw[2], h.dw[1], h.dw[0]), shape=(h.dimdw, h.dimdw))
        nd = sp.csr_matrix((h.nd[2], h.nd[1], h.nd[0]), shape=(h.dim, h.dim)).tocoo()
        self.nd_i, self.nd_j, self.nd_v = nd.row, nd.col, nd.data
        # Hnd reaches a column at most `halo` away (impurity-level moves inside one bath configuration)
        self.halo = int(np.max(np.abs(nd.row % h.dimup - nd.col % h.dimup))) if nd.nnz else 0

    def pack(self, lz, vin, send):
        pl, du = self.plan, self.dim_up
        v = np.zeros((pl.q, du))
        v[:pl.count] = vin.numpy()[:pl.nloc].reshape(pl.count, du)
        s = send.numpy().reshape(pl.world, pl.q, lz.pw)
        s[:] = 0.0
        for c in range(pl.world):
            lo, hi = c * lz.pcol - lz.halo, (c + 1) * lz.pcol + lz.halo
            a, b = max(lo, 0), min(hi, du)
            if b > a:
                s[c, :, a - lo:b - lo] = v[:, a:b]

    def rows(self, lz, vin, out):
        pl, du = self.plan, self.dim_up
        v = vin.numpy()[:pl.nloc].reshape(pl.count, du)
        hd = self.h.hd[pl.row_first:pl.row_first + pl.nloc].reshape(pl.count, du)
        out.numpy()[:pl.nloc] = (hd * v + (self.hup @ v.T).T).reshape(-1)

    def cols(self, lz, recv, hvc):
        du, dd, hl = self.dim_up, self.dim_dw, lz.halo
        w = recv.numpy().reshape(-1, lz.pw)
        o = hvc.numpy().reshape(-1, lz.pw)
        cf, cc = lz.col_first, lz.col_count
        o[:dd, hl:hl + cc] = self.hdw @ w[:dd, hl:hl + cc]
        iu, idw = self.nd_i % du, self.nd_i // du
        ju, jdw = self.nd_j % du, self.nd_j // du
        m = (iu >= cf) & (iu < cf + cc)
        assert np.all(np.abs(ju[m] - iu[m]) <= hl)
        np.add.at(o, (idw[m], iu[m] - cf + hl), self.nd_v[m] * w[jdw[m], ju[m] - cf + hl])

    def unpack_add(self, lz, back, out):
        pl, du = self.plan, self.dim_up
        b = back.numpy().reshape(pl.world, pl.q, lz.pw)
        res = out.numpy()[:pl.nloc].reshape(pl.count, du)
        for c in range(pl.world):
            a, e = c * lz.pcol, min((c + 1) * lz.pcol, du)
            if e > a:
                res[:, a:e] += b[c, :pl.count, lz.halo:lz.halo + e - a]


class _OracleTransposeKernelsFused(_OracleTransposeKernels):
    """+ the fused vector updates (edigpu_transpose_rotate_pack / _unpack_add_dot2)."""

    def rotate_pack(self, lz, first, vin, vout, ab_prev, send):
        if not first:
            a = float(ab_prev[0])
            b = float(np.sqrt(ab_prev[1] - a * a))
            t = vin.clone()
            torch.div(vout - a * t, b, out=vin)
            torch.mul(t, -b, out=vout)
        self.pack(lz, vin, send)

    def unpack_add_dot2(self, lz, vin, vout, tmp, back, out2):
        self.unpack_add(lz, back, tmp)
        vout.add_(tmp)
        out2[0] = torch.dot(vin, vout)
        out2[1] = torch.dot(vout, vout)


def _worker_transposed(rank, world, port, q, fused=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        om, _ = make_models("normal", "hybrid", 2, 2, seed=43)     # Jx, Jp != 0: Hnd with halo columns
        h = O.HNormal(om, 3, 2)
        plan = ShardPlan(h.dimdw, h.dimup, world, rank)
        k = (_OracleTransposeKernelsFused if fused else _OracleTransposeKernels)(h, plan)
        lz = TransposedLanczos(plan, k, device="cpu")
        assert lz.fused == fused
        v0 = np.random.default_rng(777).standard_normal(h.dim)
        a, b, n = lz.tridiag(torch.from_numpy(v0[plan.row_first:plan.row_first + plan.nloc].copy()), 40)
        a_ref, b_ref, n_ref = h.lanc_tridiag(v0, 40)
        err = max(np.max(np.abs(a[:12] - a_ref[:12])) / np.max(np.abs(a_ref)),
                  np.max(np.abs(b[:12] - b_ref[:12])) / np.max(np.abs(b_ref)))
        q.put((rank, float(err), int(n), int(n_ref), int(k.halo), int(lz.exchange_bytes)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("world", [2, 3])
def test_transposed_lanczos_gloo(world, fused):
    """Transposed exchange (two all-to-alls per product, Hnd through halo columns) against the serial oracle."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_transposed, args=(r, world, port, q, fused)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, n, n_ref, halo, xb in res:
        assert n == n_ref == 40 and halo > 0 and xb > 0
        assert err < 1e-10, f"rank {rank}: alpha/beta deviate from the serial oracle by {err}"
