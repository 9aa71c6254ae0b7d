"""Test infrastructure: the sharded Lanczos loops of rounds 1-2 written on torch ops and torch.distributed, driving the
library's building-block entry points (edigpu_apply_local_dev / _remote_dev, edigpu_normal_apply_rows_dev / _cols_dev,
the fused vector kernels) from Python.  The product path is the in-library loop behind ``LibraryComm``
(edipack_amd/sharding.py, csrc/edigpu_shard.hip); these classes stay as an independent second implementation that the
parity tests compare it with, and as the host of the world-2/3 gloo tests."""
from __future__ import annotations

import os
from typing import Callable

import torch
import torch.distributed as dist

from edipack_amd.sharding import ShardPlan


class TorchVecOps:
    """Elementwise part of the recurrence with torch ops (CPU / gloo tests and any device)."""

    @staticmethod
    def _real(t):
        return torch.view_as_real(t) if t.is_complex() else t

    def rotate(self, vin, vout, beta2):
        b = torch.sqrt(beta2)
        b = b.to(vin.dtype) if vin.is_complex() else b
        t = vin.clone()
        torch.div(vout, b, out=vin)
        torch.mul(t, -b, out=vout)

    def add_dot(self, vin, vout, tmp, out):
        vout.add_(tmp)
        out.copy_(torch.sum(self._real(vin) * self._real(vout)).reshape(1))

    def axpy_nrm2(self, vin, vout, alpha, out):
        a = alpha.to(vin.dtype) if vin.is_complex() else alpha
        vout.addcmul_(vin, -a)
        r = self._real(vout)
        out.copy_(torch.sum(r * r).reshape(1))

    def nrm2(self, v, out):
        r = self._real(v)
        out.copy_(torch.sum(r * r).reshape(1))

    def scale(self, v, nrm2):
        b = torch.sqrt(nrm2)
        v.div_(b.to(v.dtype) if v.is_complex() else b)

    # one-reduction recurrence (see ShardedLanczos.tridiag)
    def rotate_lazy(self, vin, vout, ab):
        a = ab[0]
        b = torch.sqrt(ab[1] - a * a)
        if vin.is_complex():
            a, b = a.to(vin.dtype), b.to(vin.dtype)
        t = vin.clone()
        torch.div(vout - a * t, b, out=vin)
        torch.mul(t, -b, out=vout)

    def add_dot2(self, vin, vout, tmp, out2):
        vout.add_(tmp)
        r = self._real(vout)
        out2[0] = torch.sum(self._real(vin) * r)
        out2[1] = torch.sum(r * r)


class NativeVecOps:
    """The same through the library's fused vector kernels (edigpu_vec_*, include/edigpu.h): one
    launch per update instead of 3-4 torch ops, no temporaries, no host synchronisation."""

    def __init__(self, device="cuda"):
        from edipack_amd import capi
        self.L, self.check = capi.lib(), capi.check
        self.work = torch.zeros(self.L.edigpu_vec_work_doubles(), dtype=torch.float64, device=device)
        self.zero = None

    @staticmethod
    def _n(t):
        return t.numel() * (2 if t.is_complex() else 1)

    @staticmethod
    def _st():
        return torch.cuda.current_stream().cuda_stream

    def rotate(self, vin, vout, beta2):
        self.check(self.L.edigpu_vec_rotate(self._n(vin), vin.data_ptr(), vout.data_ptr(), beta2.data_ptr(),
                                            self._st()))

    def add_dot(self, vin, vout, tmp, out):
        self.check(self.L.edigpu_vec_add_dot(self._n(vin), vin.data_ptr(), vout.data_ptr(), tmp.data_ptr(),
                                             out.data_ptr(), self.work.data_ptr(), self._st()))

    def axpy_nrm2(self, vin, vout, alpha, out):
        self.check(self.L.edigpu_vec_axpy_nrm2(self._n(vin), vin.data_ptr(), vout.data_ptr(), alpha.data_ptr(),
                                               out.data_ptr(), self.work.data_ptr(), self._st()))

    def nrm2(self, v, out):
        # sum(v^2) = the norm partial of (v - 0*v)
        if self.zero is None:
            self.zero = torch.zeros(1, dtype=torch.float64, device=v.device)
        self.check(self.L.edigpu_vec_axpy_nrm2(self._n(v), v.data_ptr(), v.data_ptr(), self.zero.data_ptr(),
                                               out.data_ptr(), self.work.data_ptr(), self._st()))

    def scale(self, v, nrm2):
        self.check(self.L.edigpu_vec_scale(self._n(v), v.data_ptr(), nrm2.data_ptr(), self._st()))

    def rotate_lazy(self, vin, vout, ab):
        self.check(self.L.edigpu_vec_rotate_lazy(self._n(vin), vin.data_ptr(), vout.data_ptr(), ab.data_ptr(),
                                                 self._st()))

    def add_dot2(self, vin, vout, tmp, out2):
        self.check(self.L.edigpu_vec_add_dot2(self._n(vin), vin.data_ptr(), vout.data_ptr(), tmp.data_ptr(),
                                              out2.data_ptr(), self.work.data_ptr(), self._st()))


class ShardedLanczos:
    """Three-term recurrence on a row-sharded vector (sp_lanc_tridiag semantics, MPI variant).

    ``apply_local(v_chunk, out)`` computes the shard-local part of H*v from the rank's own (padded)
    chunk and overwrites ``out``; ``apply_remote(v_full, out)`` adds the part that needs the
    gathered vector.  Both operate on torch tensors that live on ``device``.  ``vec_ops`` supplies
    the elementwise updates (TorchVecOps by default, NativeVecOps on the GPU).
    """

    def __init__(self, plan: ShardPlan, apply_local: Callable, apply_remote: Callable,
                 dtype=torch.float64, device="cpu", group=None, vec_ops=None, gathered: bool = True):
        self.plan, self.apply_local, self.apply_remote = plan, apply_local, apply_remote
        self.dtype, self.device, self.group = dtype, device, group
        self.ops = vec_ops if vec_ops is not None else TorchVecOps()
        n = plan.chunk
        self.vin = torch.zeros(n, dtype=dtype, device=device)     # padded chunk; tail stays zero
        self.vout = torch.zeros(n, dtype=dtype, device=device)
        self.tmp = torch.zeros(n, dtype=dtype, device=device)
        self.vfull = torch.zeros(n * plan.world, dtype=dtype, device=device) if gathered else None

    # -- collectives --------------------------------------------------------------------------
    def _collectives(self) -> bool:
        # a single rank needs no exchange; EDIGPU_FORCE_COLLECTIVES=1 issues them anyway (exercises the RCCL calls
        # of the N > 1 path on a one-GPU box)
        return self.plan.world > 1 or bool(os.environ.get("EDIGPU_FORCE_COLLECTIVES"))

    def _allreduce(self, t: torch.Tensor) -> torch.Tensor:
        if self._collectives():
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def hv(self) -> None:
        """tmp <- (H vin) restricted to the local rows, exchange overlapped with the local part."""
        work = None
        if self._collectives():
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

    def step(self, it: int, alphas: torch.Tensor, beta2s: torch.Tensor) -> None:
        """One lanczos_iteration; alpha_it and beta_it^2 land in alphas[it], beta2s[it] (device)."""
        ops = self.ops
        if it > 0:
            ops.rotate(self.vin, self.vout, beta2s[it - 1:it])
        self.hv()
        ops.add_dot(self.vin, self.vout, self.tmp, alphas[it:it + 1])
        self._allreduce(alphas[it:it + 1])
        ops.axpy_nrm2(self.vin, self.vout, alphas[it:it + 1], beta2s[it:it + 1])
        self._allreduce(beta2s[it:it + 1])

    # -- sp_lanc_tridiag ------------------------------------------------------------------------
    def _start(self, v_local: torch.Tensor) -> None:
        nl = self.plan.nloc
        self.vin.zero_()
        self.vin[:nl].copy_(v_local)
        self.vout.zero_()
        nrm2 = torch.zeros(1, dtype=torch.float64, device=self.device)
        self.ops.nrm2(self.vin, nrm2)
        self._allreduce(nrm2)
        self.ops.scale(self.vin, nrm2)

    @staticmethod
    def _coefficients_from_ab(abh, nlanc: int, threshold: float):
        """alanc, blanc, niter from the history of (<v|w>, <w|w>); None when beta^2 = <w|w> - alpha^2 lost more than
        three digits somewhere (or the recurrence broke down): the caller repeats the run with the exact form."""
        import numpy as np
        al, qq = abh[0::2], abh[1::2]
        b2 = qq - al * al
        alanc, blanc, ndone = np.zeros(nlanc), np.zeros(nlanc), nlanc
        for k in range(nlanc):
            if not (b2[k] > 1e-3 * qq[k]):
                return None
            alanc[k] = al[k]
            be = np.sqrt(b2[k])
            if abs(be) < threshold:
                ndone = k + 1
                break
            if k + 1 < nlanc:
                blanc[k + 1] = be
        alanc[ndone:] = 0.0
        blanc[ndone:] = 0.0
        return alanc, blanc, ndone

    def tridiag(self, v_local: torch.Tensor, nlanc: int, threshold: float = 1e-12, exact: bool = False):
        """v_local: this rank's slice (length plan.nloc).  Returns (alanc, blanc, niter) on the host.

        Default: one all-reduce of (<v|w>, <w|w>) per step, beta^2 = <w|w> - alpha^2, the axpy folded into the next
        rotate (vec_ops.add_dot2 / rotate_lazy).  exact=True (or EDIGPU_LANCZOS_EXACTBETA, or vec_ops without these
        two methods, or cancellation seen in the history): the literal two-reduction recurrence."""
        import numpy as np
        exact = exact or bool(os.environ.get("EDIGPU_LANCZOS_EXACTBETA")) or not hasattr(self.ops, "add_dot2")
        if not exact:
            self._start(v_local)
            ab = torch.zeros(2 * nlanc, dtype=torch.float64, device=self.device)
            views = [ab[2 * i:2 * i + 2] for i in range(nlanc)]
            for it in range(nlanc):
                if it > 0:
                    self.ops.rotate_lazy(self.vin, self.vout, views[it - 1])
                self.hv()
                self.ops.add_dot2(self.vin, self.vout, self.tmp, views[it])
                self._allreduce(views[it])
            res = self._coefficients_from_ab(ab.cpu().numpy(), nlanc, threshold)
            if res is not None:
                return res
            # the same values on every rank, so every rank takes this branch together
        nl = self.plan.nloc
        self.vin.zero_()
        self.vin[:nl].copy_(v_local)
        self.vout.zero_()
        alphas = torch.zeros(nlanc, dtype=torch.float64, device=self.device)
        beta2s = torch.zeros(nlanc, dtype=torch.float64, device=self.device)
        nrm2 = torch.zeros(1, dtype=torch.float64, device=self.device)
        self.ops.nrm2(self.vin, nrm2)
        self._allreduce(nrm2)
        self.ops.scale(self.vin, nrm2)
        for it in range(nlanc):
            self.step(it, alphas, beta2s)
        al = alphas.cpu().numpy()
        be = np.sqrt(beta2s.cpu().numpy())
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


class TransposedKernels:
    """The four device steps of the transposed exchange through libedigpu.so (include/edigpu.h)."""

    def __init__(self, h, plan: ShardPlan):
        from edipack_amd import capi
        self.h, self.L, self.check, self.plan = h, capi.lib(), capi.check, plan
        self.halo = h.transpose_halo()
        self.dim_up, self.dim_dw = h.dim_up, h.dim_dw
        self.work = None

    @staticmethod
    def _st():
        return torch.cuda.current_stream().cuda_stream

    def pack(self, lz, vin, send):
        pl = self.plan
        self.check(self.L.edigpu_transpose_pack(self.dim_up, pl.count, pl.q, pl.world, lz.pcol, lz.halo,
                                                vin.data_ptr(), send.data_ptr(), self._st()))

    def rows(self, lz, vin, out):
        self.h.apply_rows_dev(self.plan.first, self.plan.count, vin.data_ptr(), out.data_ptr(), self._st())

    def cols(self, lz, recv, hvc):
        self.h.apply_cols_dev(lz.col_first, lz.col_count, lz.pw, lz.halo, recv.data_ptr(), hvc.data_ptr(), self._st())

    def unpack_add(self, lz, back, out):
        pl = self.plan
        self.check(self.L.edigpu_transpose_unpack_add(self.dim_up, pl.count, pl.q, pl.world, lz.pcol, lz.halo,
                                                      back.data_ptr(), out.data_ptr(), self._st()))

    # fused with the vector updates of the recurrence (one all-reduce per step)
    def rotate_pack(self, lz, first, vin, vout, ab_prev, send):
        pl = self.plan
        self.check(self.L.edigpu_transpose_rotate_pack(1 if first else 0, self.dim_up, pl.count, pl.q, pl.world,
                                                       lz.pcol, lz.halo, vin.data_ptr(), vout.data_ptr(),
                                                       None if first else ab_prev.data_ptr(), send.data_ptr(),
                                                       self._st()))

    def unpack_add_dot2(self, lz, vin, vout, tmp, back, out2):
        pl = self.plan
        if self.work is None:
            self.work = torch.zeros(self.L.edigpu_vec_work_doubles(), dtype=torch.float64, device=vin.device)
        self.check(self.L.edigpu_transpose_unpack_add_dot2(self.dim_up, pl.count, pl.q, pl.world, lz.pcol, lz.halo,
                                                           vin.data_ptr(), vout.data_ptr(), tmp.data_ptr(),
                                                           back.data_ptr(), out2.data_ptr(), self.work.data_ptr(),
                                                           self._st()))


def _bind_fused_kernels(k: "TransposedKernels", lz, ab: torch.Tensor):
    """The four kernel launches of one fused step with every argument converted once (ctypes objects, raw
    pointers, the stream): the per-step host cost of the N > 1 loop is launch-bound at 8 ranks, and argument
    marshalling was half of it.  Returns (rotate_pack(it), rows(), cols(), unpack_dot2(it))."""
    import ctypes as C
    from edipack_amd import capi
    L, pl, h = k.L, k.plan, k.h
    if k.work is None:
        k.work = torch.zeros(L.edigpu_vec_work_doubles(), dtype=torch.float64, device=lz.vin.device)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    vp = lambda t: C.c_void_p(t.data_ptr())                      # noqa: E731
    vin, vout, tmp, send, recv, hvc, back, work = map(vp, (lz.vin, lz.vout, lz.tmp, lz.send, lz.recv, lz.hvc,
                                                            lz.back, k.work))
    ab0 = ab.data_ptr()
    du, cnt, q = C.c_int64(k.dim_up), C.c_int64(pl.count), C.c_int64(pl.q)
    world, pcol, halo = C.c_int32(pl.world), C.c_int64(lz.pcol), C.c_int32(lz.halo)
    first, ncol, pw = C.c_int64(pl.first), C.c_int64(lz.col_count), C.c_int64(lz.pw)
    col_first, hnd, one, zero, null = C.c_int64(lz.col_first), h._h, C.c_int32(1), C.c_int32(0), C.c_void_p(None)
    f_rp, f_rows, f_cols, f_un = (L.edigpu_transpose_rotate_pack, L.edigpu_normal_apply_rows_dev,
                                  L.edigpu_normal_apply_cols_dev, L.edigpu_transpose_unpack_add_dot2)

    def rotate_pack(it):
        prev = null if it == 0 else C.c_void_p(ab0 + 16 * (it - 1))
        if f_rp(one if it == 0 else zero, du, cnt, q, world, pcol, halo, vin, vout, prev, send, st):
            capi.check(1, "edigpu_transpose_rotate_pack")

    def rows():
        if f_rows(hnd, first, cnt, vin, tmp, st):
            capi.check(1, "edigpu_normal_apply_rows_dev")

    def cols():
        if f_cols(hnd, col_first, ncol, pw, halo, recv, hvc, st):
            capi.check(1, "edigpu_normal_apply_cols_dev")

    def unpack_dot2(it):
        if f_un(du, cnt, q, world, pcol, halo, vin, vout, tmp, back, C.c_void_p(ab0 + 16 * it), work, st):
            capi.check(1, "edigpu_transpose_unpack_add_dot2")

    return rotate_pack, rows, cols, unpack_dot2


class TransposedLanczos(ShardedLanczos):
    """Normal mode with the transposed exchange (reference spMatVec_mpi_normal_main,
    ED_NORMAL/ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:834-866, with vector_transpose_MPI,
    ED_NORMAL/ED_HAMILTONIAN_NORMAL_COMMON.f90:66-167).

    The vector stays row-sharded (rank r owns down rows [r q, (r+1) q)).  Per product two equal-split
    all-to-alls move (N-1)/N of the shard each way instead of an all-gather of the whole vector: pack ->
    all-to-all (in flight while the row half (Hd + Hup) runs) -> column half (Hdw + Hnd) on this rank's
    up columns of all rows -> all-to-all back -> unpack-add.  No physical transpose: the column-panel
    kernel works on the received blocks as they arrive (they are already in row order), and Hnd is served
    by a few halo columns instead of the reference's all-gather (:906-927).

    ``kernels`` supplies pack / rows / cols / unpack_add (TransposedKernels on the GPU; the CPU tests pass
    stand-ins built on the oracle's matrices) and the attributes halo, dim_up, dim_dw.
    """

    def __init__(self, plan: ShardPlan, kernels, device="cuda", group=None, vec_ops=None, stage_host: bool = False):
        self.k = kernels
        self.halo, self.dim_up, self.dim_dw = kernels.halo, kernels.dim_up, kernels.dim_dw
        self.pcol = -(-self.dim_up // plan.world)
        self.col_first = min(plan.rank * self.pcol, self.dim_up)
        self.col_count = max(0, min(self.pcol, self.dim_up - self.col_first))
        self.pw = self.pcol + 2 * self.halo
        self.stage_host = stage_host          # gloo rehearsal on a GPU: all_to_all through host memory
        self.exact = bool(os.environ.get("EDIGPU_LANCZOS_EXACTBETA"))   # two all-reduces per step, beta from ||w - alpha v||
        ShardedLanczos.__init__(self, plan, None, None, dtype=torch.float64, device=device, group=group,
                                vec_ops=vec_ops, gathered=False)
        n = plan.world * plan.q * self.pw
        self.send = torch.zeros(n, dtype=torch.float64, device=device)
        self.recv = torch.zeros(n, dtype=torch.float64, device=device)   # column shard of v
        self.hvc = torch.zeros(n, dtype=torch.float64, device=device)    # column shard of the down half; padding rows stay 0
        self.back = torch.zeros(n, dtype=torch.float64, device=device)

    @property
    def exchange_bytes(self) -> int:
        """bytes this rank sends per product (both all-to-alls, own block excluded)"""
        return 2 * 8 * (self.plan.world - 1) * self.plan.q * self.pw

    def _a2a(self, dst, src, async_op=False):
        if not self._collectives():
            dst.copy_(src)
            return None
        if self.stage_host:
            s, d = src.cpu(), torch.empty(src.numel(), dtype=src.dtype)
            dist.all_to_all_single(d, s, group=self.group)
            dst.copy_(d)
            return None
        return dist.all_to_all_single(dst, src, group=self.group, async_op=async_op)

    def _exchange_and_apply(self) -> None:
        work = self._a2a(self.recv, self.send, async_op=True)
        self.k.rows(self, self.vin, self.tmp)
        if work is not None:
            work.wait()
        self.k.cols(self, self.recv, self.hvc)
        self._a2a(self.back, self.hvc)

    def hv(self) -> None:
        self.k.pack(self, self.vin, self.send)
        self._exchange_and_apply()
        self.k.unpack_add(self, self.back, self.tmp)

    # -- fused recurrence: one all-reduce per step ------------------------------------------------
    @property
    def fused(self) -> bool:
        return hasattr(self.k, "rotate_pack") and not self.exact

    def fused_step(self, it: int, ab: torch.Tensor) -> None:
        """ab[2 it : 2 it + 2] <- (<v|w>, <w|w>) of step `it`, summed over the ranks.  The axpy w - alpha v
        and the rotate of step it-1 are applied lazily at the start (rotate_pack)."""
        self.k.rotate_pack(self, it == 0, self.vin, self.vout, None if it == 0 else ab[2 * it - 2:2 * it], self.send)
        self._exchange_and_apply()
        self.k.unpack_add_dot2(self, self.vin, self.vout, self.tmp, self.back, ab[2 * it:2 * it + 2])
        self._allreduce(ab[2 * it:2 * it + 2])

    def tridiag(self, v_local: torch.Tensor, nlanc: int, threshold: float = 1e-12):
        if not self.fused:
            return super().tridiag(v_local, nlanc, threshold, exact=True)
        self._start(v_local)
        ab = torch.zeros(2 * nlanc, dtype=torch.float64, device=self.device)
        if isinstance(self.k, TransposedKernels) and type(self).fused_step is TransposedLanczos.fused_step \
                and "fused_step" not in self.__dict__:
            # library kernels: pre-bound launches (same sequence as fused_step, less host work per step)
            rotate_pack, rows, cols, unpack_dot2 = _bind_fused_kernels(self.k, self, ab)
            views = [ab[2 * i:2 * i + 2] for i in range(nlanc)]
            for it in range(nlanc):
                rotate_pack(it)
                work = self._a2a(self.recv, self.send, async_op=True)
                rows()
                if work is not None:
                    work.wait()
                cols()
                self._a2a(self.back, self.hvc)
                unpack_dot2(it)
                self._allreduce(views[it])
        else:
            for it in range(nlanc):
                self.fused_step(it, ab)
        res = self._coefficients_from_ab(ab.cpu().numpy(), nlanc, threshold)
        if res is not None:
            return res
        # beta^2 = <w|w> - alpha^2 lost more than three digits (or the recurrence broke down): the same values on
        # every rank, so every rank repeats the run with the exact two-reduction recurrence
        return super().tridiag(v_local, nlanc, threshold, exact=True)


def gpu_transposed_hamiltonian(model, sector, world: int, rank: int, group=None, stage_host: bool = False):
    """Normal mode, transposed exchange: every rank builds the whole sector (factored tables only) and runs
    TransposedLanczos on its row shard.  Raises RuntimeError when the sector cannot be served this way
    (explicit spH0nd, phonons): fall back to gpu_sharded_hamiltonian (all-gather form)."""
    from edipack_amd.hamiltonian import SectorHamiltonian
    nup, ndw = sector
    h = SectorHamiltonian.normal_from_model(model, nup, ndw)
    try:
        plan = ShardPlan(units=h.dim_dw, unit_len=h.dim_up, world=world, rank=rank)
        lz = TransposedLanczos(plan, TransposedKernels(h, plan), group=group, vec_ops=NativeVecOps(),
                               stage_host=stage_host)
    except Exception:
        h.destroy()
        raise
    return plan, h, lz


def gpu_sharded_hamiltonian(model, workload_sector, world: int, rank: int, direct: bool = False):
    """Build this rank's shard on its GPU and return (plan, SectorHamiltonian, ShardedLanczos)."""
    from edipack_amd import capi
    from edipack_amd.hamiltonian import SectorHamiltonian

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

    lz = ShardedLanczos(plan, apply_local, apply_remote, dtype=dtype, device="cuda", vec_ops=NativeVecOps())
    return plan, h, lz


# -----------------------------------------------------------------------------------------------
# the N > 1 path inside libedigpu.so (csrc/edigpu_shard.hip): communicator + sharded recurrence in C
# -----------------------------------------------------------------------------------------------
