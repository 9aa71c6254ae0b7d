"""Host-side mirror of EDIpack's ED_HAMILTONIAN interface on top of the C ABI.

Mirrors the four entry points per mode that the reference exports from
``ED_HAMILTONIAN.f90:10-25`` (paths relative to /root/reference/src/singlesite):

==============================  =====================================================
reference                       here
==============================  =====================================================
``build_Hv_sector_<mode>``      :func:`build_Hv_sector_normal` / ``_superc`` / ``_nonsu2``
``vecDim_Hv_sector_<mode>``     :func:`vecDim_Hv_sector`
``spHtimesV_p / spHtimesV_cc``  :func:`spHtimesV_p` / :func:`spHtimesV_cc`  (Nloc, v, Hv)
``tridiag_Hv_sector_<mode>``    :func:`tridiag_Hv_sector`
``delete_Hv_sector_<mode>``     :func:`delete_Hv_sector`
==============================  =====================================================

As in the reference exactly one sector is "live" at module level (the Fortran code keeps
it in module globals, ``ED_VARS_GLOBAL.f90:190-197``); :class:`SectorHamiltonian` is the
object form of the same thing and may be instantiated several times.

All numerics run in libedigpu.so (HIP, gfx950).  Nothing here computes H*v on the CPU.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import capi

ED_MODES = {"normal": 0, "superc": 1, "nonsu2": 2}
BATH_TYPES = {"normal": 0, "hybrid": 1, "replica": 2, "general": 3}


@dataclass
class ImpurityModel:
    """The module-global inputs the reference's builders read (``ED_INPUT_VARS``,
    ``dmft_bath``, ``impHloc``, the ``*_internal`` interaction tables)."""
    ed_mode: str = "normal"
    bath_type: str = "normal"
    norb: int = 1
    nbath: int = 1
    nspin: int = 1
    hfmode: bool = True
    xmu: float = 0.0
    uloc: np.ndarray = field(default_factory=lambda: np.zeros(1))
    ust: np.ndarray | float = 0.0   # scalar (Kanamori, off-diagonal) or [norb,norb] table
    jh: np.ndarray | float = 0.0
    jx: np.ndarray | float = 0.0
    jp: np.ndarray | float = 0.0
    hloc: np.ndarray | None = None  # impHloc(+mfHloc) [nspin,nspin,norb,norb] complex
    pair_field: np.ndarray | None = None
    be: np.ndarray | None = None    # dmft_bath%e [nspin, norb (1 for hybrid), nbath]
    bv: np.ndarray | None = None    # dmft_bath%v [nspin, norb, nbath]
    bd: np.ndarray | None = None    # dmft_bath%d (superc)
    bu: np.ndarray | None = None    # dmft_bath%u (nonsu2)
    # replica / general baths: hb[is, js, iorb, jorb, k] = build_Hreplica/Hgeneral(lambda_k) (is, js over
    # Nspin, or the Nambu index in superc); hybridisations in bv (replica: item(k)%v for all is, iorb;
    # general: item(k)%vg(iorb + Norb*(is-1)))
    hb: np.ndarray | None = None
    # phonons (normal mode): cut-off Nph, frequency, displacement field, coupling matrix g_ph[a, b] (density
    # couplings g_aa only)
    nph: int = 0
    w0_ph: float = 0.0
    a_ph: float = 0.0
    g_ph: np.ndarray | None = None
    # normal mode only: spin_field[iorb, xyz] (SPIN_FIELD_X/Y/Z; z is read), exc_field[4] (EXC_FIELD; (1), (4) are
    # read), coulomb_sundry lines as (U, (orb_i, spin_i), (orb_j, spin_j), (orb_k, spin_k), (orb_l, spin_l)) for
    # U cd_i cd_j c_k c_l with 0-based orbitals and spin 0 up / 1 down (ED_NORMAL/stored/H_sundry.f90)
    spin_field: np.ndarray | None = None
    exc_field: np.ndarray | None = None
    sundry: list | None = None

    @property
    def ns(self) -> int:
        """Levels per spin, ``ed_setup_dimensions`` (ED_SETUP.f90:118-126)."""
        return self.nbath + self.norb if self.bath_type == "hybrid" else (self.nbath + 1) * self.norb

    def _table(self, x) -> np.ndarray:
        no = self.norb
        if np.isscalar(x):
            return float(x) * (1.0 - np.eye(no))   # set_umatrix, ED_PARSE_UMATRIX.f90:136-142
        return np.asarray(x, dtype=float).reshape(no, no)

    def to_c(self) -> capi.EdigpuModel:
        m = capi.EdigpuModel()
        m.ed_mode = ED_MODES[self.ed_mode]
        m.bath_type = BATH_TYPES[self.bath_type]
        m.norb, m.nbath, m.nspin = self.norb, self.nbath, self.nspin
        m.hfmode = int(self.hfmode)
        m.xmu = float(self.xmu)
        no = self.norb
        if no > capi.MAXORB or self.nbath > capi.MAXBATH:
            raise capi.EdigpuError("ImpurityModel: norb/nbath exceed EDIGPU_MAXORB/EDIGPU_MAXBATH")
        np.ctypeslib.as_array(m.uloc)[:no] = np.asarray(self.uloc, dtype=float)[:no]
        for name in ("ust", "jh", "jx", "jp"):
            np.ctypeslib.as_array(getattr(m, name)).reshape(capi.MAXORB, capi.MAXORB)[:no, :no] = \
                self._table(getattr(self, name))
        if self.hloc is not None:
            h = np.asarray(self.hloc, dtype=complex)
            k = h.shape[0]
            v = np.ctypeslib.as_array(m.hloc).reshape(2, 2, capi.MAXORB, capi.MAXORB, 2)
            v[:k, :k, :no, :no, 0] = h.real
            v[:k, :k, :no, :no, 1] = h.imag
        if self.pair_field is not None:
            np.ctypeslib.as_array(m.pair_field)[:no] = np.asarray(self.pair_field, dtype=float)
        m.nph, m.w0_ph, m.a_ph = int(self.nph), float(self.w0_ph), float(self.a_ph)
        if self.g_ph is not None:
            np.ctypeslib.as_array(m.g_ph).reshape(capi.MAXORB, capi.MAXORB)[:no, :no] = \
                np.asarray(self.g_ph, dtype=float).reshape(no, no)
        if self.spin_field is not None:
            np.ctypeslib.as_array(m.spin_field).reshape(capi.MAXORB, 3)[:no, :] = \
                np.asarray(self.spin_field, dtype=float).reshape(no, 3)
        if self.exc_field is not None:
            np.ctypeslib.as_array(m.exc_field)[:] = np.asarray(self.exc_field, dtype=float).reshape(4)
        if self.sundry:
            if len(self.sundry) > capi.MAXSUNDRY:
                raise capi.EdigpuError("ImpurityModel: more than EDIGPU_MAXSUNDRY coulomb_sundry lines")
            m.nsundry = len(self.sundry)
            for il, (u, *ops) in enumerate(self.sundry):
                m.sundry_u[il] = float(u)
                for k, (orb, spin) in enumerate(ops):
                    m.sundry_op[il * 8 + 2 * k] = int(orb) + 1
                    m.sundry_op[il * 8 + 2 * k + 1] = int(spin) + 1
        if self.bath_type in ("replica", "general"):
            if self.hb is None or self.bv is None:
                raise capi.EdigpuError("ImpurityModel: replica/general baths need hb and bv")
            hb = np.asarray(self.hb, dtype=complex)
            k = hb.shape[0]
            v = np.ctypeslib.as_array(m.hb).reshape(2, 2, capi.MAXORB, capi.MAXORB, capi.MAXBATH, 2)
            v[:k, :k, :no, :no, : self.nbath, 0] = hb.real
            v[:k, :k, :no, :no, : self.nbath, 1] = hb.imag
        for name in ("be", "bv", "bd", "bu"):
            arr = getattr(self, name)
            if arr is None:
                continue
            a = np.asarray(arr, dtype=float)
            np.ctypeslib.as_array(getattr(m, name)).reshape(2, capi.MAXORB, capi.MAXBATH)[
                : a.shape[0], : a.shape[1], : a.shape[2]] = a
        return m


def sector_map(model: "ImpurityModel", q1: int, q2: int = 0, which: int = 0) -> np.ndarray:
    """build_sector's map (ED_SECTOR.f90:165-373) as int32: normal mode -> up (which=0) / down (which=1) map of the
    sector (q1, q2) = (Nup, Ndw); superc / nonsu2 -> the map of sector q1."""
    cm = model.to_c()
    n = C.c_int64(0)
    capi.check(capi.lib().edigpu_sector_map(C.byref(cm), q1, q2, which, None, C.byref(n)), "edigpu_sector_map")
    out = np.zeros(n.value, np.int32)
    capi.check(capi.lib().edigpu_sector_map(C.byref(cm), q1, q2, which, capi.pi32(out), C.byref(n)), "edigpu_sector_map")
    return out


def sector_map_jz(model: "ImpurityModel", ntot: int, twojz: int) -> np.ndarray:
    """build_sector's map of the nonsu2 sector (Ntot, twoJz) of JZ_BASIS=T (ED_SECTOR.f90:289-350)."""
    cm = model.to_c()
    n = C.c_int64(0)
    capi.check(capi.lib().edigpu_sector_map_jz(C.byref(cm), ntot, twojz, None, C.byref(n)), "edigpu_sector_map_jz")
    out = np.zeros(n.value, np.int32)
    capi.check(capi.lib().edigpu_sector_map_jz(C.byref(cm), ntot, twojz, capi.pi32(out), C.byref(n)), "edigpu_sector_map_jz")
    return out


def _csr_args(rowptr, col, val, cplx=False):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.complex128 if cplx else np.float64)
    return rowptr, col, val


class SectorHamiltonian:
    """One sector Hamiltonian resident on the GPU (an ``edigpu_handle``)."""

    def __init__(self, handle: C.c_void_p, owned: bool = True):
        self._h = handle
        self._owned = owned        # False: the handle belongs to a SectorCache
        info = (C.c_int64 * 10)()
        capi.check(capi.lib().edigpu_info(self._h, info), "edigpu_info")
        (self.dim, self.nloc, self.row_first, cplx, self.kind, self.dim_up, self.dim_dw,
         self.nnz_a, self.nnz_b, self.device) = [int(x) for x in info]
        self.is_complex = bool(cplx)
        self.dtype = np.complex128 if self.is_complex else np.float64

    # ---- constructors -------------------------------------------------------------------
    @classmethod
    def normal_from_model(cls, model: ImpurityModel, nup: int, ndw: int, dw_first: int = 0,
                          dw_count: int = -1) -> "SectorHamiltonian":
        h = C.c_void_p()
        cm = model.to_c()
        capi.check(capi.lib().edigpu_normal_build(C.byref(h), C.byref(cm), nup, ndw, dw_first, dw_count),
                   "edigpu_normal_build")
        return cls(h)

    @classmethod
    def normal_cmplx_from_model(cls, model: ImpurityModel, nup: int, ndw: int) -> "SectorHamiltonian":
        """ed_mode=normal with complex algebra (the reference's -D_CMPLX_NORMAL build): complex impHloc / replica
        bath matrices, complex vectors."""
        h = C.c_void_p()
        cm = model.to_c()
        capi.check(capi.lib().edigpu_normal_build_z(C.byref(h), C.byref(cm), nup, ndw), "edigpu_normal_build_z")
        return cls(h)

    @classmethod
    def flat_from_model(cls, model: ImpurityModel, sector: int, row_first: int = 0,
                        row_count: int = -1) -> "SectorHamiltonian":
        h = C.c_void_p()
        cm = model.to_c()
        capi.check(capi.lib().edigpu_flat_build(C.byref(h), C.byref(cm), sector, row_first, row_count),
                   "edigpu_flat_build")
        return cls(h)

    @classmethod
    def flat_jz_from_model(cls, model: ImpurityModel, ntot: int, twojz: int, row_first: int = 0,
                           row_count: int = -1) -> "SectorHamiltonian":
        """nonsu2 sector (Ntot, twoJz) of JZ_BASIS=T (build_sector, ED_SECTOR.f90:289-350)."""
        h = C.c_void_p()
        cm = model.to_c()
        capi.check(capi.lib().edigpu_flat_build_jz(C.byref(h), C.byref(cm), ntot, twojz, row_first, row_count),
                   "edigpu_flat_build_jz")
        return cls(h)

    @classmethod
    def direct_jz_from_model(cls, model: ImpurityModel, ntot: int, twojz: int, row_first: int = 0,
                             row_count: int = -1) -> "SectorHamiltonian":
        """on-the-fly form of the nonsu2 sector (Ntot, twoJz) of JZ_BASIS=T."""
        h = C.c_void_p()
        cm = model.to_c()
        capi.check(capi.lib().edigpu_direct_build_jz(C.byref(h), C.byref(cm), ntot, twojz, row_first, row_count),
                   "edigpu_direct_build_jz")
        return cls(h)

    @classmethod
    def direct_from_model(cls, model: ImpurityModel, sector: int, row_first: int = 0,
                          row_count: int = -1) -> "SectorHamiltonian":
        """ed_sparse_H=F: on-the-fly H*v, nothing stored (directMatVec_*_main)."""
        h = C.c_void_p()
        cm = model.to_c()
        capi.check(capi.lib().edigpu_direct_build(C.byref(h), C.byref(cm), sector, row_first, row_count),
                   "edigpu_direct_build")
        return cls(h)

    @classmethod
    def orbs_from_model(cls, model: ImpurityModel, nups, ndws, row_first: int = 0,
                        row_count: int = -1) -> "SectorHamiltonian":
        """ed_total_ud=F: sector with per-orbital (Nup_a, Ndw_a) (build_Hv_sector_normal -> ed_buildh_normal_orbs);
        row_first / row_count: a row shard (spMatVec_mpi_normal_orbs, all-gather form)."""
        no = model.norb
        a = np.ascontiguousarray(nups, dtype=np.int32)
        b = np.ascontiguousarray(ndws, dtype=np.int32)
        if a.shape != (no,) or b.shape != (no,):
            raise capi.EdigpuError("orbs_from_model: nups/ndws need Norb entries each")
        h = C.c_void_p()
        cm = model.to_c()
        capi.check(capi.lib().edigpu_orbs_build_rows(C.byref(h), C.byref(cm), capi.pi32(a), capi.pi32(b), row_first,
                                                     row_count), "edigpu_orbs_build")
        return cls(h)

    @classmethod
    def orbs_from_arrays(cls, dims, hd, factors) -> "SectorHamiltonian":
        """Hand over spH0d and spH0ups(1:Norb), spH0dws(1:Norb) (each a (rowptr, col, val) triple, in that order)."""
        dims = np.ascontiguousarray(dims, dtype=np.int64)
        hd = np.ascontiguousarray(hd, dtype=np.float64)
        rps, cols, vals, base = [np.zeros(1, np.int64)], [], [], 0
        for (rp, col, val) in factors:
            rp = np.asarray(rp, dtype=np.int64)
            rps.append(rp[1:] + base)
            base += int(rp[-1])
            cols.append(np.asarray(col, dtype=np.int32))
            vals.append(np.asarray(val, dtype=np.float64))
        rp = np.ascontiguousarray(np.concatenate(rps))
        col = np.ascontiguousarray(np.concatenate(cols)) if cols else np.zeros(0, np.int32)
        val = np.ascontiguousarray(np.concatenate(vals)) if vals else np.zeros(0, np.float64)
        h = C.c_void_p()
        capi.check(capi.lib().edigpu_orbs_create(C.byref(h), len(dims), capi.pi64(dims), capi.pd(hd), capi.pi64(rp),
                                                 capi.pi32(col), capi.pd(val)), "edigpu_orbs_create")
        return cls(h)

    @classmethod
    def normal_from_arrays(cls, dim_up, dim_dw, hd, up, dw, nd=None, dw_first=0, dw_count=None):
        """Hand over spH0d / spH0ups(1) / spH0dws(1) / spH0nd as (rowptr, col, val) triples."""
        if dw_count is None:
            dw_count = dim_dw
        hd = np.ascontiguousarray(hd, dtype=np.float64)
        ur, uc, uv = _csr_args(*up)
        dr, dc, dv = _csr_args(*dw)
        h = C.c_void_p()
        if nd is not None:
            nr, nc, nv = _csr_args(*nd)
            args = (capi.pi64(nr), capi.pi32(nc), capi.pd(nv))
        else:
            args = (None, None, None)
        capi.check(capi.lib().edigpu_normal_create(
            C.byref(h), dim_up, dim_dw, dw_first, dw_count, capi.pd(hd),
            capi.pi64(ur), capi.pi32(uc), capi.pd(uv), capi.pi64(dr), capi.pi32(dc), capi.pd(dv), *args),
            "edigpu_normal_create")
        return cls(h)

    @classmethod
    def csr_from_arrays(cls, rowptr, col, val, ncol_global=None, row_first=0):
        """Hand over spH0 (local rows, global columns); real or complex by dtype of val."""
        cplx = np.iscomplexobj(val)
        rowptr, col, val = _csr_args(rowptr, col, val, cplx)
        nrow = rowptr.shape[0] - 1
        if ncol_global is None:
            ncol_global = nrow
        h = C.c_void_p()
        fn = capi.lib().edigpu_csr_create_z if cplx else capi.lib().edigpu_csr_create_d
        capi.check(fn(C.byref(h), nrow, ncol_global, row_first, capi.pi64(rowptr), capi.pi32(col),
                      capi.pd(val.view(np.float64))), "edigpu_csr_create")
        return cls(h)

    # ---- queries --------------------------------------------------------------------------
    def vecDim(self) -> int:
        """vecDim_Hv_sector_* (ED_NORMAL/ED_HAMILTONIAN_NORMAL.f90:286-313)."""
        return self.nloc

    def image_info(self):
        """(factored, Hnd terms, diagonal classes, panel variant, panel-major width, 0) of a normal-mode handle."""
        a = (C.c_int32 * 6)()
        capi.check(capi.lib().edigpu_image_info(self._h, a), "edigpu_image_info")
        return tuple(a)

    def algorithmic_bytes(self):
        a, b = C.c_double(), C.c_double()
        capi.check(capi.lib().edigpu_algorithmic_bytes(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def export_normal(self):
        nup_nnz_dw = None
        L = capi.lib()
        # sizes: up/dw nnz are not separately in info -> query row pointers first
        ur = np.zeros(self.dim_up + 1, np.int64)
        dr = np.zeros(self.dim_dw + 1, np.int64)
        nr = np.zeros(self.nloc + 1, np.int64)
        capi.check(L.edigpu_normal_export(self._h, None, capi.pi64(ur), None, None, capi.pi64(dr), None, None,
                                          capi.pi64(nr), None, None))
        hd = np.zeros(self.nloc)
        uc, uv = np.zeros(ur[-1], np.int32), np.zeros(ur[-1])
        dc, dv = np.zeros(dr[-1], np.int32), np.zeros(dr[-1])
        nc, nv = np.zeros(nr[-1], np.int32), np.zeros(nr[-1])
        capi.check(L.edigpu_normal_export(self._h, capi.pd(hd), None, capi.pi32(uc), capi.pd(uv), None,
                                          capi.pi32(dc), capi.pd(dv), None, capi.pi32(nc), capi.pd(nv)))
        del nup_nnz_dw
        return hd, (ur, uc, uv), (dr, dc, dv), (nr, nc, nv)

    def export_csr(self):
        L = capi.lib()
        rp = np.zeros(self.nloc + 1, np.int64)
        nnz = self.nnz_a + self.nnz_b
        col = np.zeros(nnz, np.int32)
        val = np.zeros(nnz, self.dtype)
        capi.check(L.edigpu_csr_export(self._h, capi.pi64(rp), capi.pi32(col), capi.pd(val.view(np.float64))))
        return rp, col, val

    # ---- H*v --------------------------------------------------------------------------------
    def apply(self, v: np.ndarray, hv: np.ndarray | None = None) -> np.ndarray:
        """Callback-compatible product on host arrays (dd_sparse_HxV / cc_sparse_HxV)."""
        v = np.ascontiguousarray(v, dtype=self.dtype)
        if hv is None:
            hv = np.empty_like(v)
        fn = capi.lib().edigpu_apply_z if self.is_complex else capi.lib().edigpu_apply_d
        capi.check(fn(self._h, v.shape[0], capi.pd(v.view(np.float64)), capi.pd(hv.view(np.float64))),
                   "edigpu_apply")
        return hv

    def apply_dev(self, v_full_ptr: int, hv_ptr: int, stream: int = 0) -> None:
        capi.check(capi.lib().edigpu_apply_dev(self._h, v_full_ptr, hv_ptr, stream if stream else None), "edigpu_apply_dev")

    def apply_local_dev(self, v_local_ptr: int, hv_ptr: int, stream: int = 0) -> None:
        capi.check(capi.lib().edigpu_apply_local_dev(self._h, v_local_ptr, hv_ptr, stream if stream else None),
                   "edigpu_apply_local_dev")

    def apply_remote_dev(self, v_full_ptr: int, hv_ptr: int, stream: int = 0) -> None:
        capi.check(capi.lib().edigpu_apply_remote_dev(self._h, v_full_ptr, hv_ptr, stream if stream else None),
                   "edigpu_apply_remote_dev")

    def apply_cops_to(self, dst: "SectorHamiltonian", v_src_ptr: int, v_dst_ptr: int, coefs, creates, iorbs, ispins,
                      stream: int = 0) -> None:
        """apply_Cops on device vectors: |dst> = sum_s coefs[s] * c^(+)_{iorbs[s], ispins[s]} |src> (normal mode;
        creates[s] truthy = c^+)."""
        n = len(coefs)
        a = (C.c_double * n)(*[float(x) for x in coefs])
        o = (C.c_int32 * n)(*[1 if x else -1 for x in creates])
        io = (C.c_int32 * n)(*[int(x) for x in iorbs])
        sp = (C.c_int32 * n)(*[int(x) for x in ispins])
        capi.check(capi.lib().edigpu_apply_cops_normal(self._h, dst._h, v_src_ptr, v_dst_ptr, n, a, o, io, sp,
                                                       stream if stream else None), "edigpu_apply_cops_normal")

    def apply_cops_flat_to(self, dst: "SectorHamiltonian", v_src_ptr: int, v_dst_ptr: int, coefs, creates, iorbs, ispins,
                           stream: int = 0) -> None:
        """apply_Cops on device vectors of superc / nonsu2 sectors, complex coefficients."""
        n = len(coefs)
        a = (C.c_double * (2 * n))(*[x for c in coefs for x in (complex(c).real, complex(c).imag)])
        o = (C.c_int32 * n)(*[1 if x else -1 for x in creates])
        io = (C.c_int32 * n)(*[int(x) for x in iorbs])
        sp = (C.c_int32 * n)(*[int(x) for x in ispins])
        capi.check(capi.lib().edigpu_apply_cops_flat(self._h, dst._h, v_src_ptr, v_dst_ptr, n, a, o, io, sp,
                                                     stream if stream else None), "edigpu_apply_cops_flat")

    # ---- transposed exchange (normal mode, N > 1; include/edigpu.h) ---------------------------
    def transpose_halo(self) -> int:
        h = C.c_int32(0)
        capi.check(capi.lib().edigpu_normal_transpose_info(self._h, C.byref(h)), "edigpu_normal_transpose_info")
        return h.value

    def apply_rows_dev(self, dw_first: int, dw_count: int, v_rows_ptr: int, hv_rows_ptr: int, stream: int = 0):
        capi.check(capi.lib().edigpu_normal_apply_rows_dev(self._h, dw_first, dw_count, v_rows_ptr, hv_rows_ptr,
                                                           stream if stream else None),
                   "edigpu_normal_apply_rows_dev")

    def apply_cols_dev(self, col_first: int, col_count: int, row_stride: int, halo: int, w_ptr: int, hv_ptr: int,
                       stream: int = 0):
        capi.check(capi.lib().edigpu_normal_apply_cols_dev(self._h, col_first, col_count, row_stride, halo, w_ptr, hv_ptr,
                                                           stream if stream else None),
                   "edigpu_normal_apply_cols_dev")

    # ---- Lanczos ------------------------------------------------------------------------------
    def lanczos_tridiag(self, vin: np.ndarray, nlanc: int, threshold: float = 1e-12):
        """sp_lanc_tridiag semantics, vector resident on the device."""
        vin = np.ascontiguousarray(vin, dtype=self.dtype)
        a = np.zeros(nlanc)
        b = np.zeros(nlanc)
        nd = C.c_int(0)
        capi.check(capi.lib().edigpu_lanczos_tridiag(self._h, capi.pd(vin.view(np.float64)), nlanc, capi.pd(a),
                                                     capi.pd(b), threshold, C.byref(nd)), "edigpu_lanczos_tridiag")
        return a, b, nd.value

    def lanczos_eigh_multi(self, neigen: int, ncv: int = 0, tol: float = 1e-12, maxrestart: int = 300,
                           v0: np.ndarray | None = None, want_vectors: bool = True):
        """sp_eigh (ARPACK) semantics: the lowest `neigen` eigenpairs -> (evals, evecs[neigen, nloc] | None,
        nconv, nmatvec)."""
        neigen = min(int(neigen), self.nloc)
        ev = np.zeros(neigen)
        vec = np.zeros((neigen, self.nloc), dtype=self.dtype) if want_vectors else None
        nc, nmv = C.c_int(0), C.c_int(0)
        v0p = None
        if v0 is not None:
            v0 = np.ascontiguousarray(v0, dtype=self.dtype)
            v0p = v0.ctypes.data_as(C.c_void_p)
        capi.check(capi.lib().edigpu_lanczos_eigh_multi(
            self._h, neigen, ncv, tol, maxrestart, v0p, capi.pd(ev),
            vec.ctypes.data_as(C.c_void_p) if want_vectors else None, C.byref(nc), C.byref(nmv)),
            "edigpu_lanczos_eigh_multi")
        return ev, vec, nc.value, nmv.value

    def lanczos_tridiag_dev(self, vin_ptr: int, nlanc: int, threshold: float = 1e-12):
        """tridiag_Hv_sector_* with the seed already on the device: -> (alanc, blanc, niter, norm2)."""
        a = np.zeros(nlanc)
        b = np.zeros(nlanc)
        nd = C.c_int(0)
        n2 = C.c_double(0.0)
        capi.check(capi.lib().edigpu_lanczos_tridiag_dev(self._h, vin_ptr, nlanc, capi.pd(a), capi.pd(b), threshold,
                                                         C.byref(nd), C.byref(n2)), "edigpu_lanczos_tridiag_dev")
        return a, b, nd.value, n2.value

    def apply_op_to(self, dst: "SectorHamiltonian", v_src_ptr: int, v_dst_ptr: int, iorb: int, ispin: int,
                    create: bool, stream: int = 0) -> None:
        """apply_op_C / apply_op_CDG on device vectors: |dst> = c^(+)_{iorb,ispin} |src> (self = source sector)."""
        fn = capi.lib().edigpu_apply_op_normal if self.kind == 0 else capi.lib().edigpu_apply_op_flat
        capi.check(fn(self._h, dst._h, v_src_ptr, v_dst_ptr, iorb, ispin, int(create),
                      stream if stream else None), "edigpu_apply_op")

    def lanczos_eigh(self, nitermax: int = 512, tol: float = 1e-12, check_every: int = 10,
                     v0: np.ndarray | None = None, want_vector: bool = True):
        """sp_lanc_eigh semantics (lowest eigenpair)."""
        ev = C.c_double()
        nd = C.c_int(0)
        vec = np.zeros(self.nloc, dtype=self.dtype) if want_vector else None
        v0p = None
        if v0 is not None:
            v0 = np.ascontiguousarray(v0, dtype=self.dtype)
            v0p = capi.pd(v0.view(np.float64))
        capi.check(capi.lib().edigpu_lanczos_eigh(
            self._h, nitermax, tol, check_every, v0p, C.byref(ev),
            capi.pd(vec.view(np.float64)) if want_vector else None, C.byref(nd)), "edigpu_lanczos_eigh")
        return ev.value, vec, nd.value

    def time_apply(self, warmup: int, steps: int, lanczos: int = 0) -> float:
        """ms per step: lanczos = 0 the boundary product (edigpu_apply_dev), 1 full Lanczos steps, 2 the plain product
        as the device-resident loops compute it (panel-major vectors where they are used)."""
        ms = C.c_double()
        capi.check(capi.lib().edigpu_time_apply(self._h, warmup, steps, int(lanczos), C.byref(ms)),
                   "edigpu_time_apply")
        return ms.value

    def lanczos_bench(self, warmup: int, steps: int):
        """(ms wall per Lanczos step, ms per H*v launch from HIP events)."""
        a, b = C.c_double(), C.c_double()
        capi.check(capi.lib().edigpu_lanczos_bench(self._h, warmup, steps, C.byref(a), C.byref(b)),
                   "edigpu_lanczos_bench")
        return a.value, b.value

    def destroy(self) -> None:
        if self._h:
            if self._owned:
                capi.lib().edigpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


# -----------------------------------------------------------------------------------------------
# module-level singleton API, named after the reference's procedures
# -----------------------------------------------------------------------------------------------
_live: SectorHamiltonian | None = None


def _set_live(h: SectorHamiltonian) -> SectorHamiltonian:
    global _live
    if _live is not None:
        raise capi.EdigpuError("build_Hv_sector: a sector is already allocated (call delete_Hv_sector first)")
    _live = h
    return h


def build_Hv_sector_normal(model: ImpurityModel, nup: int, ndw: int) -> SectorHamiltonian:
    """build_Hv_sector_normal (ED_NORMAL/ED_HAMILTONIAN_NORMAL.f90:31-206), ed_sparse_H=T."""
    return _set_live(SectorHamiltonian.normal_from_model(model, nup, ndw))


def build_Hv_sector_superc(model: ImpurityModel, sz: int) -> SectorHamiltonian:
    """build_Hv_sector_superc (ED_SUPERC/ED_HAMILTONIAN_SUPERC.f90:33-135)."""
    return _set_live(SectorHamiltonian.flat_from_model(model, sz))


def build_Hv_sector_nonsu2(model: ImpurityModel, ntot: int) -> SectorHamiltonian:
    """build_Hv_sector_nonsu2 (ED_NONSU2/ED_HAMILTONIAN_NONSU2.f90:31-128)."""
    return _set_live(SectorHamiltonian.flat_from_model(model, ntot))


def _need_live() -> SectorHamiltonian:
    if _live is None:
        # the reference: stop "... Hsector NOT allocated" (ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:67)
        raise capi.EdigpuError("Hsector NOT allocated")
    return _live


def vecDim_Hv_sector() -> int:
    return _need_live().vecDim()


def spHtimesV_p(Nloc: int, v: np.ndarray, Hv: np.ndarray) -> None:
    """dd_sparse_HxV(Nloc, v, Hv) (ED_VARS_GLOBAL.f90:111-122): Hv is overwritten in place."""
    h = _need_live()
    if h.is_complex:
        raise capi.EdigpuError("spHtimesV_p: live sector is complex, use spHtimesV_cc")
    if v.shape[0] != Nloc or Hv.shape[0] != Nloc:
        raise capi.EdigpuError("spHtimesV_p: Nloc does not match the vectors")
    h.apply(v, Hv)


def spHtimesV_cc(Nloc: int, v: np.ndarray, Hv: np.ndarray) -> None:
    """cc_sparse_HxV(Nloc, v, Hv) (ED_VARS_GLOBAL.f90:125-132)."""
    h = _need_live()
    if not h.is_complex:
        raise capi.EdigpuError("spHtimesV_cc: live sector is real, use spHtimesV_p")
    if v.shape[0] != Nloc or Hv.shape[0] != Nloc:
        raise capi.EdigpuError("spHtimesV_cc: Nloc does not match the vectors")
    h.apply(v, Hv)


def delete_Hv_sector() -> None:
    """delete_Hv_sector_* (ED_NORMAL/ED_HAMILTONIAN_NORMAL.f90:212-279)."""
    global _live
    if _live is not None:
        _live.destroy()
        _live = None


def tridiag_Hv_sector(build, vvinit: np.ndarray, nlanc: int | None = None, lanc_ngfiter: int = 200):
    """tridiag_Hv_sector_* (ED_NORMAL/ED_HAMILTONIAN_NORMAL.f90:321-369): norm2 = <v|v>,
    build the sector (``build`` is a zero-argument callable doing the build_Hv_sector_* call),
    tridiagonalise with Nlanc = min(Dim, lanc_ngfiter) steps, delete the sector.
    Returns (alanc, blanc, norm2)."""
    vvinit = np.asarray(vvinit)
    norm2 = float(np.real(np.vdot(vvinit, vvinit)))
    h = build()
    try:
        n = min(h.dim, lanc_ngfiter) if nlanc is None else nlanc
        alanc = np.zeros(n)
        blanc = np.zeros(n)
        if norm2 != 0.0:
            alanc, blanc, _ = h.lanczos_tridiag(vvinit / np.sqrt(norm2), n)
    finally:
        delete_Hv_sector()
    return alanc, blanc, norm2


class SectorCache:
    """Per-solve cache of sector handles (``edigpu_cache_*``): what replaces the reference's rebuild
    of the sector Hamiltonian in every ``tridiag_Hv_sector_*`` call.  ``get`` returns a SectorHamiltonian that borrows
    the cached handle (its ``destroy`` is a no-op); the two most recently returned ones are never evicted."""
    KINDS = {"normal": 0, "stored": 1, "direct": 2, "normal_cmplx": 3}

    def __init__(self, max_device_bytes: int = 8 << 30):
        self._c = C.c_void_p()
        capi.check(capi.lib().edigpu_cache_create(C.byref(self._c), int(max_device_bytes)), "edigpu_cache_create")

    def get(self, model: ImpurityModel, kind: str, q1: int, q2: int = 0) -> SectorHamiltonian:
        h = C.c_void_p()
        cm = model.to_c()
        capi.check(capi.lib().edigpu_cache_get(self._c, C.byref(cm), self.KINDS[kind], q1, q2, C.byref(h)),
                   "edigpu_cache_get")
        return SectorHamiltonian(h, owned=False)

    def stats(self) -> dict:
        a = (C.c_int64 * 5)()
        capi.check(capi.lib().edigpu_cache_stats(self._c, a), "edigpu_cache_stats")
        return dict(zip(("hits", "misses", "evictions", "bytes", "entries"), [int(x) for x in a]))

    def clear(self) -> None:
        capi.check(capi.lib().edigpu_cache_clear(self._c), "edigpu_cache_clear")

    def destroy(self) -> None:
        if self._c:
            capi.lib().edigpu_cache_destroy(self._c)
            self._c = None
