"""Row-sharding of one sector state vector over the GPUs of a node + the sharded Lanczos loop.

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI on ROCm; ``gloo``
in the CPU tests).  Replaces the reference's MPI data flow:

* normal mode: the down index is split (``ED_NORMAL/ED_HAMILTONIAN_NORMAL.f90:129-142``); here the
  whole vector is all-gathered (variant (B) of SURVEY.md 8e: required anyway for ``spH0nd``,
  ``..._STORED_HxV.f90:906-927``) and the down part reads the gathered copy;
* superc / nonsu2: rows are split, ``MPI_Allgatherv`` of the vector
  (``ED_SUPERC/ED_HAMILTONIAN_SUPERC_STORED_HxV.f90:418-421``,
  ``ED_NONSU2/ED_HAMILTONIAN_NONSU2_STORED_HxV.f90:256-259``) becomes one ``all_gather_into_tensor``.

The exchange is issued asynchronously and overlaps the shard-local part of H*v (diagonal + up
part / ``loc`` block), which needs no remote data; the non-local part runs after it.

Shard sizes: every rank owns q = ceil(units/P) units except the tail ranks, so that rank r starts
at unit r*q.  The gathered buffer therefore is the global vector followed by padding, and RCCL
gets equal chunks (the reference puts the remainder on the first/last ranks instead; the choice
is not observable in the results).
"""
from __future__ import annotations

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


class ShardedLanczos:
    """Three-term recurrence on a row-sharded vector (sp_lanc_tridiag semantics, MPI variant).

    ``apply_local(v_chunk, out)`` computes the shard-local part of H*v from the rank's own (padded)
    chunk and overwrites ``out``; ``apply_remote(v_full, out)`` adds the part that needs the
    gathered vector.  Both operate on torch tensors that live on ``device``.
    """

    def __init__(self, plan: ShardPlan, apply_local: Callable, apply_remote: Callable,
                 dtype=torch.float64, device="cpu", group=None):
        self.plan, self.apply_local, self.apply_remote = plan, apply_local, apply_remote
        self.dtype, self.device, self.group = dtype, device, group
        n = plan.chunk
        self.vin = torch.zeros(n, dtype=dtype, device=device)     # padded chunk; tail stays zero
        self.vout = torch.zeros(n, dtype=dtype, device=device)
        self.tmp = torch.zeros(n, dtype=dtype, device=device)
        self.vfull = torch.zeros(n * plan.world, dtype=dtype, device=device)

    # -- collectives --------------------------------------------------------------------------
    def _allreduce(self, t: torch.Tensor) -> torch.Tensor:
        if self.plan.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def _dot_real(self, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        if a.is_complex():
            a, b = torch.view_as_real(a), torch.view_as_real(b)
        return self._allreduce(torch.sum(a * b).reshape(1))

    def hv(self) -> None:
        """tmp <- (H vin) restricted to the local rows, exchange overlapped with the local part."""
        work = None
        if self.plan.world > 1:
            # complex vectors travel as (re,im) pairs: every backend handles real tensors
            src = torch.view_as_real(self.vin) if self.vin.is_complex() else self.vin
            dst = torch.view_as_real(self.vfull) if self.vfull.is_complex() else self.vfull
            work = dist.all_gather_into_tensor(dst, src, group=self.group, async_op=True)
        else:
            self.vfull.copy_(self.vin)
        self.apply_local(self.vin, self.tmp)
        if work is not None:
            work.wait()
        self.apply_remote(self.vfull, self.tmp)

    # -- sp_lanc_tridiag ------------------------------------------------------------------------
    def tridiag(self, v_local: torch.Tensor, nlanc: int, threshold: float = 0.0):
        """v_local: this rank's slice (length plan.nloc).  Returns (alanc, blanc, niter) on the host."""
        nl = self.plan.nloc
        self.vin.zero_()
        self.vin[:nl].copy_(v_local)
        self.vout.zero_()
        alphas, betas = [], []
        nrm = torch.sqrt(self._dot_real(self.vin, self.vin))
        self.vin.div_(nrm.to(self.vin.dtype) if self.vin.is_complex() else nrm)
        beta = None
        for it in range(nlanc):
            if it > 0:
                b = beta.to(self.vin.dtype) if self.vin.is_complex() else beta
                self.tmp.copy_(self.vin)
                torch.div(self.vout, b, out=self.vin)
                torch.mul(self.tmp, -b, out=self.vout)
            self.hv()
            self.vout.add_(self.tmp)
            alpha = self._dot_real(self.vin, self.vout)
            a = alpha.to(self.vin.dtype) if self.vin.is_complex() else alpha
            self.vout.addcmul_(self.vin, -a)
            beta = torch.sqrt(self._dot_real(self.vout, self.vout))
            alphas.append(alpha)
            betas.append(beta)
        al = torch.cat(alphas).cpu().numpy()
        be = torch.cat(betas).cpu().numpy()
        import numpy as np
        alanc = np.zeros(nlanc)
        blanc = np.zeros(nlanc)
        ndone = nlanc
        for k in range(nlanc):
            alanc[k] = al[k]
            if abs(be[k]) < threshold:
                ndone = k + 1
                break
            if k + 1 < nlanc:
                blanc[k + 1] = be[k]
        alanc[ndone:] = 0.0
        blanc[ndone:] = 0.0
        return alanc, blanc, ndone


def gpu_sharded_hamiltonian(model, workload_sector, world: int, rank: int, direct: bool = False):
    """Build this rank's shard on its GPU and return (plan, SectorHamiltonian, ShardedLanczos)."""
    from . import capi
    from .hamiltonian import SectorHamiltonian

    L = capi.lib()
    import ctypes as C
    cm = model.to_c()
    if model.ed_mode == "normal":
        nup, ndw = workload_sector
        d_up, d_dw = C.c_int64(), C.c_int64()
        # DimUp = dim(nup, 0), DimDw = dim(0, ndw): the library's own sector arithmetic
        capi.check(L.edigpu_sector_dim(C.byref(cm), nup, 0, C.byref(d_up)))
        capi.check(L.edigpu_sector_dim(C.byref(cm), 0, ndw, C.byref(d_dw)))
        plan = ShardPlan(units=d_dw.value, unit_len=d_up.value, world=world, rank=rank)
        h = SectorHamiltonian.normal_from_model(model, nup, ndw, dw_first=plan.first, dw_count=plan.count)
        dtype = torch.float64
    else:
        dim = C.c_int64()
        capi.check(L.edigpu_sector_dim(C.byref(cm), int(workload_sector), 0, C.byref(dim)))
        plan = ShardPlan(units=dim.value, unit_len=1, world=world, rank=rank)
        build = SectorHamiltonian.direct_from_model if direct else SectorHamiltonian.flat_from_model
        h = build(model, int(workload_sector), row_first=plan.first, row_count=plan.count)
        dtype = torch.complex128

    def apply_local(v_chunk, out):
        h.apply_local_dev(v_chunk.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)

    def apply_remote(v_full, out):
        h.apply_remote_dev(v_full.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)

    lz = ShardedLanczos(plan, apply_local, apply_remote, dtype=dtype, device="cuda")
    return plan, h, lz
