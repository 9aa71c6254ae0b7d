"""Row shards of one sector state vector over the GPUs of a node: the shard plan and the ctypes front-end of the
library's communicator (csrc/edigpu_shard.hip), which owns the exchange and the whole sharded recurrence.

One process per GPU.  Replaces the reference's MPI data flow:

* normal mode: the down index is split (``ED_NORMAL/ED_HAMILTONIAN_NORMAL.f90:129-142``); the library runs the
  transposed exchange of ``spMatVec_mpi_normal_main`` (two all-to-alls per product) on whole-sector handles and the
  all-gather form on handed-over shards;
* superc / nonsu2: rows are split, ``MPI_Allgatherv`` of the vector
  (``ED_SUPERC/ED_HAMILTONIAN_SUPERC_STORED_HxV.f90:418-421``,
  ``ED_NONSU2/ED_HAMILTONIAN_NONSU2_STORED_HxV.f90:256-259``) becomes one all-gather beside the shard-local block.

Shard sizes: every rank owns q = ceil(units/P) units except the tail ranks, so that rank r starts at unit r*q
(``edigpu_shard_plan``; the reference puts the remainder on the first/last ranks instead -- not observable in the
results).

The torch-level loops of rounds 1-2 (``ShardedLanczos``, ``TransposedLanczos`` and their vector-op back-ends), which
drive the building-block entry points from Python, are test infrastructure now: ``tests/torch_sharded_loop.py``.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Callable

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class ShardPlan:
    units: int          # DimDw (normal) or Dim (flat)
    unit_len: int       # DimUp (normal) or 1 (flat): vector elements per unit
    world: int
    rank: int

    @property
    def q(self) -> int:
        return -(-self.units // self.world)

    @property
    def first(self) -> int:
        return min(self.rank * self.q, self.units)

    @property
    def count(self) -> int:
        return max(0, min(self.q, self.units - self.first))

    @property
    def nloc(self) -> int:            # vecDim_Hv_sector_* of this rank
        return self.count * self.unit_len

    @property
    def chunk(self) -> int:           # padded per-rank chunk exchanged by the all-gather
        return self.q * self.unit_len

    @property
    def row_first(self) -> int:
        return self.first * self.unit_len

    def counts(self):
        return [max(0, min(self.q, self.units - min(r * self.q, self.units))) * self.unit_len
                for r in range(self.world)]




class LibraryComm:
    """edigpu_comm: RCCL over xGMI (``unique_id`` = the 128 bytes of edigpu_comm_unique_id made on rank 0 and handed
    to every rank by the host) or, with ``shm_name``, the host-staged shared-memory transport for ranks that share
    a GPU.  The Fortran host gets the same object through fortran/edigpu_shim.f90 (gpu_comm_create)."""

    def __init__(self, rank: int, world: int, unique_id: bytes | None = None, shm_name: str | None = None,
                 slot_bytes: int = 1 << 26):
        import ctypes as C
        from . import capi
        self._capi, self.rank, self.world = capi, rank, world
        self._c = C.c_void_p()
        if shm_name is not None:
            capi.check(capi.lib().edigpu_comm_create_shm(C.byref(self._c), rank, world, shm_name.encode(), slot_bytes),
                       "edigpu_comm_create_shm")
        else:
            buf = C.create_string_buffer(unique_id, 128) if unique_id is not None else None
            capi.check(capi.lib().edigpu_comm_create(C.byref(self._c), rank, world, buf), "edigpu_comm_create")

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C
        from . import capi
        buf = C.create_string_buffer(128)
        capi.check(capi.lib().edigpu_comm_unique_id(buf), "edigpu_comm_unique_id")
        return buf.raw

    def plan(self, units: int):
        """(first, count, q) of this rank's shard of `units` (edigpu_shard_plan)."""
        import ctypes as C
        f, n, q = C.c_int64(), C.c_int64(), C.c_int64()
        self._capi.check(self._capi.lib().edigpu_shard_plan(units, self.world, self.rank, C.byref(f), C.byref(n),
                                                            C.byref(q)), "edigpu_shard_plan")
        return f.value, n.value, q.value

    def shard_info(self, h):
        """(exchange kind: 0 all-gather, 1 transposed on column blocks, 2 transposed on padded panels; q; columns or
        panels per rank; halo)."""
        import ctypes as C
        a = (C.c_int32 * 4)()
        self._capi.check(self._capi.lib().edigpu_shard_info(h._h, self._c, a), "edigpu_shard_info")
        return tuple(a)

    def apply(self, h, v_shard):
        """spMatVec_mpi_*: (Nloc, v, Hv) on this rank's shard, host arrays."""
        import numpy as np
        v = np.ascontiguousarray(v_shard, dtype=h.dtype)
        hv = np.empty_like(v)
        fn = self._capi.lib().edigpu_apply_sharded_z if h.is_complex else self._capi.lib().edigpu_apply_sharded_d
        self._capi.check(fn(h._h, self._c, v.shape[0], self._capi.pd(v.view(np.float64)),
                            self._capi.pd(hv.view(np.float64))), "edigpu_apply_sharded")
        return hv

    def tridiag(self, h, v_shard, nlanc: int, threshold: float = 1e-12):
        """sp_lanc_tridiag(MpiComm, ...) -> (alanc, blanc, niter, norm2); v_shard: host array of this rank's slice."""
        import ctypes as C
        import numpy as np
        v = np.ascontiguousarray(v_shard, dtype=h.dtype)
        a, b = np.zeros(nlanc), np.zeros(nlanc)
        nd, n2 = C.c_int(0), C.c_double(0.0)
        ptr = v.ctypes.data_as(C.c_void_p) if v.size else None
        self._capi.check(self._capi.lib().edigpu_lanczos_tridiag_sharded(
            h._h, self._c, ptr, nlanc, self._capi.pd(a), self._capi.pd(b), threshold, C.byref(nd), C.byref(n2)),
            "edigpu_lanczos_tridiag_sharded")
        return a, b, nd.value, n2.value

    def eigh_multi(self, h, neigen: int, nloc: int, v0_shard=None, ncv: int = 0, tol: float = 1e-12, maxrestart: int = 300,
                   vectors: bool = True):
        """sp_eigh(MpiComm, ...): lowest `neigen` eigenpairs with every vector a device-resident shard.
        Returns (evals, evecs[neigen, nloc] or None, nconv, nmatvec); nloc = this rank's elements."""
        import ctypes as C
        import numpy as np
        ev = np.zeros(neigen)
        x = np.zeros((neigen, nloc), dtype=h.dtype) if vectors else None
        v0 = None if v0_shard is None else np.ascontiguousarray(v0_shard, dtype=h.dtype)
        nc, nm = C.c_int(0), C.c_int(0)
        self._capi.check(self._capi.lib().edigpu_lanczos_eigh_multi_sharded(
            h._h, self._c, neigen, ncv, tol, maxrestart, None if v0 is None or v0.size == 0 else v0.ctypes.data_as(C.c_void_p),
            self._capi.pd(ev), None if x is None or x.size == 0 else x.ctypes.data_as(C.c_void_p), C.byref(nc), C.byref(nm)),
            "edigpu_lanczos_eigh_multi_sharded")
        return ev, x, nc.value, nm.value

    def eigh(self, h, nloc: int, v0_shard=None, nitermax: int = 300, tol: float = 1e-12):
        """sp_lanc_eigh(MpiComm, ...): (lowest eigenvalue, this rank's shard of its vector, products used)."""
        import ctypes as C
        import numpy as np
        e = np.zeros(1)
        x = np.zeros(nloc, dtype=h.dtype)
        v0 = None if v0_shard is None else np.ascontiguousarray(v0_shard, dtype=h.dtype)
        nm = C.c_int(0)
        self._capi.check(self._capi.lib().edigpu_lanczos_eigh_sharded(
            h._h, self._c, nitermax, tol, None if v0 is None or v0.size == 0 else v0.ctypes.data_as(C.c_void_p),
            self._capi.pd(e), x.ctypes.data_as(C.c_void_p) if x.size else None, C.byref(nm)), "edigpu_lanczos_eigh_sharded")
        return e[0], x, nm.value

    def apply_cops(self, src, dst, v_src_shard, nloc_dst: int, ops):
        """apply_Cops on shards: ops = [(coef, create, iorb, ispin)]; returns this rank's shard of the destination."""
        import ctypes as C
        import numpy as np
        v = np.ascontiguousarray(v_src_shard, dtype=src.dtype)
        out = np.zeros(nloc_dst, dtype=dst.dtype)
        n = len(ops)
        a = (C.c_double * (2 * n))(*[x for o in ops for x in (complex(o[0]).real, complex(o[0]).imag)])
        cr = (C.c_int32 * n)(*[1 if o[1] else -1 for o in ops])
        io = (C.c_int32 * n)(*[int(o[2]) for o in ops])
        sp = (C.c_int32 * n)(*[int(o[3]) for o in ops])
        self._capi.check(self._capi.lib().edigpu_apply_cops_sharded(
            src._h, dst._h, self._c, v.ctypes.data_as(C.c_void_p) if v.size else None,
            out.ctypes.data_as(C.c_void_p) if out.size else None, n, a, cr, io, sp), "edigpu_apply_cops_sharded")
        return out

    def bench(self, h, warmup: int, steps: int):
        """(ms per sharded Lanczos step, bytes this rank sends per product)."""
        import ctypes as C
        ms, xb = C.c_double(0.0), C.c_int64(0)
        self._capi.check(self._capi.lib().edigpu_lanczos_bench_sharded(h._h, self._c, warmup, steps, C.byref(ms),
                                                                       C.byref(xb)), "edigpu_lanczos_bench_sharded")
        return ms.value, xb.value

    def exchange_bench(self, h, steps: int = 20):
        """(ranks RCCL reports for this communicator: 0 = shared-memory transport, ms of one product's collectives)."""
        import ctypes as C
        n, ms = C.c_int32(0), C.c_double(0.0)
        self._capi.check(self._capi.lib().edigpu_exchange_bench(h._h, self._c, steps, C.byref(n), C.byref(ms)),
                         "edigpu_exchange_bench")
        return n.value, ms.value

    def destroy(self):
        if self._c:
            self._capi.lib().edigpu_comm_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def library_sharded_sector(model, sector, comm: LibraryComm, direct: bool = False, exchange: str = "auto",
                           cmplx: bool = False):
    """This rank's handle for the in-library N > 1 calls: normal mode -> the whole sector (transposed exchange) unless
    exchange == "allgather"; everything else -> this rank's row shard.  Returns (handle, first unit, unit count)."""
    import ctypes as C
    from . import capi
    from .hamiltonian import SectorHamiltonian
    L, cm = capi.lib(), model.to_c()
    if model.ed_mode == "normal":
        nup, ndw = sector
        d_dw = C.c_int64()
        capi.check(L.edigpu_sector_dim(C.byref(cm), 0, ndw, C.byref(d_dw)))
        first, count, _ = comm.plan(d_dw.value)
        if cmplx:       # _CMPLX_NORMAL: whole sector, served through its doubled real sector (transposed exchange)
            return SectorHamiltonian.normal_cmplx_from_model(model, nup, ndw), first, count
        if model.nph > 0:   # phonon blocks: whole sector, every block through the transposed exchange (density couplings)
            return SectorHamiltonian.normal_from_model(model, nup, ndw), first, count
        if exchange != "allgather":
            h = SectorHamiltonian.normal_from_model(model, nup, ndw)
            try:
                h.transpose_halo()
                return h, first, count
            except capi.EdigpuError:      # explicit spH0nd: all-gather form
                h.destroy()
        return SectorHamiltonian.normal_from_model(model, nup, ndw, dw_first=first, dw_count=count), first, count
    dim = C.c_int64()
    capi.check(L.edigpu_sector_dim(C.byref(cm), int(sector), 0, C.byref(dim)))
    first, count, _ = comm.plan(dim.value)
    build = SectorHamiltonian.direct_from_model if direct else SectorHamiltonian.flat_from_model
    return build(model, int(sector), row_first=first, row_count=count), first, count

