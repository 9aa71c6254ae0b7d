"""CPU tests of the multi-GPU host logic with world_size 2 and 3 over gloo: the shard plan (edigpu_shard_plan), the index
maps of the transposed exchange AS THE LIBRARY'S KERNELS COMPUTE THEM (edigpu_exchange_send_map / _back_map: the same
inline functions, csrc/exchange_index.hpp, that transpose_pack_kernel, ks_rotate3 and the unpack kernels call), the
all-gather exchange with padded equal chunks, the local/remote split of H*v and the sharded three-term recurrence
reproduce the serial oracle.

The per-shard products are computed here with scipy on the ORACLE's matrices (test infrastructure standing in for the
HIP kernels, which need a GPU); what is under test is the distributed data flow that replaces spMatVec_mpi_* /
vector_transpose_MPI / MPI_Allgatherv in the reference.  The loops that drive it (tests/torch_sharded_loop.py) are the
torch-level second implementation; the buffers they exchange are packed and unpacked through the library's maps."""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from edipack_amd.sharding import ShardPlan
from tests.torch_sharded_loop import ShardedLanczos, TransposedLanczos
from tests.common import make_models


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _lib():
    """libedigpu.so without a GPU: only its host-only entry points are called here."""
    import ctypes as C
    from edipack_amd import capi
    return capi.lib(), C


def _exchange_maps(dim_up, nrows, q, world, pcol, halo):
    L, C = _lib()
    send = np.empty(world * q * (pcol + 2 * halo), dtype=np.int64)
    back = np.empty(nrows * dim_up, dtype=np.int64)
    p64 = C.POINTER(C.c_int64)
    assert L.edigpu_exchange_send_map(dim_up, nrows, q, world, pcol, halo, send.ctypes.data_as(p64)) == 0
    assert L.edigpu_exchange_back_map(dim_up, nrows, q, world, pcol, halo, back.ctypes.data_as(p64)) == 0
    return send, back


def test_shard_plan_covers_everything():
    L, C = _lib()
    for units, ulen, world in [(3432, 3432, 8), (20, 15, 2), (5, 7, 8), (924, 1, 3), (1, 1, 4)]:
        plans = [ShardPlan(units, ulen, world, r) for r in range(world)]
        for r, p in enumerate(plans):      # the Python mirror == the library's plan
            f, c, q = C.c_int64(), C.c_int64(), C.c_int64()
            assert L.edigpu_shard_plan(units, world, r, C.byref(f), C.byref(c), C.byref(q)) == 0
            assert (f.value, c.value, q.value) == (p.first, p.count, p.q)
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

    def _maps(self, lz):
        if getattr(self, "_m", None) is None:
            self._m = _exchange_maps(self.dim_up, self.plan.count, self.plan.q, self.plan.world, lz.pcol, lz.halo)
        return self._m

    def pack(self, lz, vin, send):
        # through the library's send map (what transpose_pack_kernel / ks_rotate3 compute on the device)
        src, _ = self._maps(lz)
        v = vin.numpy()[:self.plan.nloc]
        s = send.numpy()
        s[:src.size] = np.where(src >= 0, v[np.maximum(src, 0)], 0.0)

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
        # through the library's back map (transpose_unpack_add_kernel / kv_unpack_add_dot2 / ks_unpack...)
        _, slot = self._maps(lz)
        out.numpy()[:self.plan.nloc] += back.numpy()[slot]


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


def _worker_maps(rank, world, port, q, dim_up, dim_dw, halo):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan = ShardPlan(dim_dw, dim_up, world, rank)
        pcol = -(-dim_up // world)
        pw = pcol + 2 * halo
        v = np.random.default_rng(99).standard_normal((dim_dw, dim_up))            # the whole vector, on every rank
        mine = v[plan.first:plan.first + plan.count].reshape(-1)
        src, slot = _exchange_maps(dim_up, plan.count, plan.q, world, pcol, halo)
        send = torch.from_numpy(np.where(src >= 0, mine[np.maximum(src, 0)] if mine.size else 0.0, 0.0).copy())
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send)                                         # equal splits of q * pw
        # what vector_transpose_MPI delivers (ED_HAMILTONIAN_NORMAL_COMMON.f90:66-167): ALL rows of this rank's block of
        # columns -- here with `halo` columns on both sides, zeros outside the sector and in the tail ranks' padding
        got = recv.numpy().reshape(world * plan.q, pw)
        want = np.zeros((world * plan.q, pw))
        lo = rank * pcol - halo
        for j in range(pw):
            col = lo + j
            if 0 <= col < dim_up:
                for r in range(world):
                    pr = ShardPlan(dim_dw, dim_up, world, r)
                    want[r * plan.q:r * plan.q + pr.count, j] = v[pr.first:pr.first + pr.count, col]
        err = float(np.max(np.abs(got - want)))
        # the way back: every rank returns its block unchanged; the back map must pick exactly the own elements
        back = torch.empty_like(recv)
        dist.all_to_all_single(back, recv)
        err2 = float(np.max(np.abs(back.numpy()[slot] - mine))) if mine.size else 0.0
        q.put((rank, err, err2))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dim_up,dim_dw,halo", [(2, 20, 15, 2), (3, 35, 20, 3), (3, 7, 5, 0), (2, 9, 3, 1)])
def test_library_exchange_maps_transpose_between_gloo_ranks(world, dim_up, dim_dw, halo):
    """The library's own index maps (host-only entry points) move a row-sharded vector to column blocks and back between
    real processes: the received buffer is the transposed shard vector_transpose_MPI produces."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_maps, args=(r, world, port, q, dim_up, dim_dw, halo)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, err2 in res:
        assert err == 0.0 and err2 == 0.0, (rank, err, err2)
