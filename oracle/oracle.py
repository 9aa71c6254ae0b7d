"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of the CPU oracle (oracle/edipack_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product (edipack_amd/) never does.

Besides loading the C restatement it holds the small host-side pieces that are
needed to reproduce the reference's own regression fixtures
(/root/reference/test/src/*/*.check):

* ``init_dmft_bath``   -- ED_BATH/ED_BATH_DMFT.f90:178-307 (deterministic start bath)
* ``kanamori_tables``  -- ED_PARSE_UMATRIX.f90:136-142 (ed_use_kanamori=T branch)
* ``ground_state``     -- dense LAPACK spectrum of every sector, as the reference does
                          for dim <= lanc_dim_threshold (ED_NORMAL/ED_DIAG_NORMAL.f90:226-236)

Parity status: pinned by tests/test_oracle_golden.py against evals/dens/docc.check.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libedipack_oracle.so")

MAXORB = 5
MAXBATH = 16

ED_MODES = {"normal": 0, "superc": 1, "nonsu2": 2}
BATH_TYPES = {"normal": 0, "hybrid": 1, "replica": 2, "general": 3}


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (idempotent)."""
    srcs = [os.path.join(HERE, f) for f in ("edipack_oracle.c", "edipack_oracle_flat.inc", "edipack_oracle_orbs.inc", "edipack_oracle.h")]
    if (not force and os.path.exists(LIB_PATH)
            and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return LIB_PATH
    subprocess.check_call(["make", "-s", "-C", HERE, "-B"])
    return LIB_PATH


_D = C.c_double


MAXSUNDRY = 64


class OrcModel(C.Structure):
    _fields_ = [
        ("ed_mode", C.c_int), ("bath_type", C.c_int),
        ("norb", C.c_int), ("nbath", C.c_int), ("nspin", C.c_int),
        ("hfmode", C.c_int),
        ("xmu", _D),
        ("uloc", _D * MAXORB),
        ("ust", (_D * MAXORB) * MAXORB),
        ("jh", (_D * MAXORB) * MAXORB),
        ("jx", (_D * MAXORB) * MAXORB),
        ("jp", (_D * MAXORB) * MAXORB),
        ("hloc_re", (((_D * MAXORB) * MAXORB) * 2) * 2),
        ("hloc_im", (((_D * MAXORB) * MAXORB) * 2) * 2),
        ("mfh_re", (((_D * MAXORB) * MAXORB) * 2) * 2),
        ("mfh_im", (((_D * MAXORB) * MAXORB) * 2) * 2),
        ("anom_re", (_D * MAXORB) * MAXORB),
        ("anom_im", (_D * MAXORB) * MAXORB),
        ("pair_field", _D * MAXORB),
        ("spin_field", (_D * 3) * MAXORB),
        ("exc_field", _D * 4),
        ("be", ((_D * MAXBATH) * MAXORB) * 2),
        ("bv", ((_D * MAXBATH) * MAXORB) * 2),
        ("bd", ((_D * MAXBATH) * MAXORB) * 2),
        ("bu", ((_D * MAXBATH) * MAXORB) * 2),
        ("hb_re", ((((_D * MAXBATH) * MAXORB) * MAXORB) * 2) * 2),
        ("hb_im", ((((_D * MAXBATH) * MAXORB) * MAXORB) * 2) * 2),
        ("vr", _D * MAXBATH),
        ("vg", (_D * MAXBATH) * (2 * MAXORB)),
        ("nph", C.c_int),
        ("w0_ph", _D), ("a_ph", _D),
        ("g_ph", (_D * MAXORB) * MAXORB),
        ("nsundry", C.c_int),
        ("sundry_op", (C.c_int * 8) * MAXSUNDRY),
        ("sundry_u", _D * MAXSUNDRY),
    ]


def _np_view(struct: C.Structure, name: str) -> np.ndarray:
    """numpy view (no copy) onto an array member of a ctypes struct."""
    f = getattr(struct, name)
    return np.ctypeslib.as_array(f)


@dataclass
class Model:
    """Python-side description of one impurity problem (the reference's input globals)."""
    ed_mode: str = "normal"
    bath_type: str = "normal"
    norb: int = 1
    nbath: int = 1
    nspin: int = 1
    hfmode: bool = True
    xmu: float = 0.0
    uloc: tuple = (2.0,)
    ust: float = 0.0
    jh: float = 0.0
    jx: float = 0.0
    jp: float = 0.0
    ed_hw_bath: float = 2.0
    deltasc: float = 0.02
    # impHloc[ispin, jspin, iorb, jorb] complex
    hloc: np.ndarray | None = None
    pair_field: tuple | None = None
    # bath arrays e,v,d,u [nspin, norb(or 1), nbath]; None -> init_dmft_bath
    be: np.ndarray | None = None
    bv: np.ndarray | None = None
    bd: np.ndarray | None = None
    bu: np.ndarray | None = None
    # replica / general baths: hbath_tmp[is, js, iorb, jorb, k] = build_Hreplica/Hgeneral(lambda_k)
    # (is, js over Nspin, or over the Nambu index in superc), hybridisations item(k)%v (replica, [nbath])
    # or item(k)%vg (general, [nspin*norb, nbath])
    hb: np.ndarray | None = None
    vr: np.ndarray | None = None
    vg: np.ndarray | None = None
    # phonons: cut-off Nph (DimPh = Nph + 1), frequency, displacement field, coupling matrix g_ph[a, b]
    nph: int = 0
    w0_ph: float = 0.0
    a_ph: float = 0.0
    g_ph: np.ndarray | None = None
    # spin_field[iorb, xyz] (SPIN_FIELD_X/Y/Z), exc_field[4] (EXC_FIELD), coulomb_sundry: list of
    # (U, (orb_i, spin_i), (orb_j, spin_j), (orb_k, spin_k), (orb_l, spin_l)) for U cd_i cd_j c_k c_l, 0-based orbitals,
    # spin 0 up / 1 down
    spin_field: np.ndarray | None = None
    exc_field: np.ndarray | None = None
    sundry: list | None = None

    @property
    def ns(self) -> int:
        # ED_SETUP.f90:118-126
        return self.nbath + self.norb if self.bath_type == "hybrid" else (self.nbath + 1) * self.norb


def init_dmft_bath(m: Model) -> None:
    """ED_BATH/ED_BATH_DMFT.f90:211-244 -- deterministic initial bath for normal/hybrid."""
    nb, hw = m.nbath, m.ed_hw_bath
    nfoo = 1 if m.bath_type == "hybrid" else m.norb
    e = np.zeros(nb)
    e[0] = -hw
    e[nb - 1] = hw
    nh = nb // 2
    if nb % 2 == 0 and nb >= 4:
        de = hw / max(nh - 1, 1)
        e[nh - 1] = -1.0e-1
        e[nh] = 1.0e-1
        for i in range(2, nh):
            e[i - 1] = -hw + (i - 1) * de
            e[nb - i] = hw - (i - 1) * de
    elif nb % 2 != 0 and nb >= 3:
        de = hw / nh
        e[nh] = 0.0
        for i in range(2, nh + 1):
            e[i - 1] = -hw + (i - 1) * de
            e[nb - i] = hw - (i - 1) * de
    m.be = np.broadcast_to(e, (m.nspin, nfoo, nb)).copy()
    m.bv = np.full((m.nspin, m.norb, nb), max(0.1, 1.0 / np.sqrt(float(nb))))
    m.bd = np.full((m.nspin, nfoo, nb), m.deltasc) if m.ed_mode == "superc" else None
    m.bu = m.bv.copy() if m.ed_mode == "nonsu2" else None


def to_struct(m: Model) -> OrcModel:
    s = OrcModel()
    s.ed_mode = ED_MODES[m.ed_mode]
    s.bath_type = BATH_TYPES[m.bath_type]
    s.norb, s.nbath, s.nspin = m.norb, m.nbath, m.nspin
    s.hfmode = int(m.hfmode)
    s.xmu = m.xmu
    no = m.norb
    s.nph, s.w0_ph, s.a_ph = int(m.nph), float(m.w0_ph), float(m.a_ph)
    if m.g_ph is not None:
        _np_view(s, "g_ph")[:no, :no] = np.asarray(m.g_ph, dtype=float).reshape(no, no)
    assert no <= MAXORB and m.nbath <= MAXBATH
    if m.spin_field is not None:
        _np_view(s, "spin_field")[:no, :] = np.asarray(m.spin_field, dtype=float).reshape(no, 3)
    if m.exc_field is not None:
        _np_view(s, "exc_field")[:] = np.asarray(m.exc_field, dtype=float)
    if m.sundry:
        assert len(m.sundry) <= MAXSUNDRY
        s.nsundry = len(m.sundry)
        for il, (u, *ops) in enumerate(m.sundry):
            s.sundry_u[il] = float(u)
            for k, (orb, spin) in enumerate(ops):
                s.sundry_op[il][2 * k] = int(orb) + 1
                s.sundry_op[il][2 * k + 1] = int(spin) + 1
    _np_view(s, "uloc")[:no] = np.asarray(m.uloc, dtype=float)[:no]
    # ED_PARSE_UMATRIX.f90:136-142 (ed_use_kanamori): off-diagonal constants
    off = 1.0 - np.eye(no)
    _np_view(s, "ust")[:no, :no] = m.ust * off
    _np_view(s, "jh")[:no, :no] = m.jh * off
    _np_view(s, "jx")[:no, :no] = m.jx * off
    _np_view(s, "jp")[:no, :no] = m.jp * off
    if m.hloc is not None:
        h = np.asarray(m.hloc, dtype=complex)
        nsn = h.shape[0]
        _np_view(s, "hloc_re")[:nsn, :nsn, :no, :no] = h.real
        _np_view(s, "hloc_im")[:nsn, :nsn, :no, :no] = h.imag
    if m.pair_field is not None:
        _np_view(s, "pair_field")[:no] = np.asarray(m.pair_field, dtype=float)
    if m.bath_type in ("replica", "general"):
        assert m.hb is not None, "replica/general bath: hb (the per-replica matrices) is required"
        hb = np.asarray(m.hb, dtype=complex)
        n1 = hb.shape[0]
        _np_view(s, "hb_re")[:n1, :n1, :no, :no, : m.nbath] = hb.real
        _np_view(s, "hb_im")[:n1, :n1, :no, :no, : m.nbath] = hb.imag
        if m.bath_type == "replica":
            _np_view(s, "vr")[: m.nbath] = np.asarray(m.vr, dtype=float)
        else:
            vg = np.asarray(m.vg, dtype=float)
            _np_view(s, "vg")[: vg.shape[0], : m.nbath] = vg
        return s
    if m.be is None:
        init_dmft_bath(m)
    for name, arr in (("be", m.be), ("bv", m.bv), ("bd", m.bd), ("bu", m.bu)):
        if arr is None:
            continue
        a = np.asarray(arr, dtype=float)
        _np_view(s, name)[: a.shape[0], : a.shape[1], : a.shape[2]] = a
    return s


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB_PATH)
    vp, i64p, i32p, dp = C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_double)
    L.orc_ns.restype = C.c_int
    L.orc_ns.argtypes = [C.POINTER(OrcModel)]
    L.orc_binomial.restype = C.c_int64
    L.orc_binomial.argtypes = [C.c_int, C.c_int]
    L.orc_build_sector_normal.argtypes = [C.c_int, C.c_int, C.c_int, i32p, i32p]
    L.orc_build_sector_superc.restype = C.c_int64
    L.orc_build_sector_superc.argtypes = [C.c_int, C.c_int, i32p]
    L.orc_build_sector_nonsu2.restype = C.c_int64
    L.orc_build_sector_nonsu2.argtypes = [C.c_int, C.c_int, i32p]
    L.orc_buildh_normal_main.restype = vp
    L.orc_buildh_normal_main.argtypes = [C.POINTER(OrcModel), C.c_int, C.c_int]
    L.orc_hnormal_free.argtypes = [vp]
    L.orc_hnormal_sizes.argtypes = [vp, i64p]
    L.orc_spmatvec_normal_main.argtypes = [vp, dp, dp]
    L.orc_spmatvec_normal_ph.argtypes = [vp, C.POINTER(OrcModel), dp, dp]
    L.orc_hnormal_dense.argtypes = [vp, dp]
    L.orc_spmatvec_flat_ph.argtypes = [vp, C.POINTER(OrcModel), dp, dp]
    L.orc_lanc_tridiag_flat_ph.restype = C.c_int
    L.orc_lanc_tridiag_flat_ph.argtypes = [vp, C.POINTER(OrcModel), dp, C.c_int, dp, dp, C.c_double]
    L.orc_lanc_tridiag_normal_ph.restype = C.c_int
    L.orc_lanc_tridiag_normal_ph.argtypes = [vp, C.POINTER(OrcModel), dp, C.c_int, dp, dp, C.c_double]
    L.orc_lanc_tridiag_normal.restype = C.c_int
    L.orc_lanc_tridiag_normal.argtypes = [vp, dp, C.c_int, dp, dp, C.c_double]
    ip = C.POINTER(C.c_int)
    L.orc_buildh_normal_orbs.restype = vp
    L.orc_buildh_normal_orbs.argtypes = [C.POINTER(OrcModel), ip, ip]
    L.orc_horbs_free.argtypes = [vp]
    L.orc_horbs_sizes.argtypes = [vp, i64p]
    L.orc_spmatvec_normal_orbs.argtypes = [vp, dp, dp]
    L.orc_horbs_dense.argtypes = [vp, dp]
    L.orc_lanc_tridiag_orbs.restype = C.c_int
    L.orc_lanc_tridiag_orbs.argtypes = [vp, dp, C.c_int, dp, dp, C.c_double]
    L.orc_buildh_superc_main.restype = vp
    L.orc_buildh_superc_main.argtypes = [C.POINTER(OrcModel), C.c_int]
    L.orc_buildh_nonsu2_main.restype = vp
    L.orc_buildh_nonsu2_main.argtypes = [C.POINTER(OrcModel), C.c_int]
    L.orc_buildh_nonsu2_jz.restype = vp
    L.orc_buildh_nonsu2_jz.argtypes = [C.POINTER(OrcModel), C.c_int, C.c_int]
    L.orc_build_sector_nonsu2_jz.restype = C.c_int64
    L.orc_build_sector_nonsu2_jz.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, i32p]
    L.orc_directmatvec_nonsu2_main.restype = C.c_int
    L.orc_directmatvec_nonsu2_main.argtypes = [C.POINTER(OrcModel), C.c_int, dp, dp]
    L.orc_directmatvec_nonsu2_rows.argtypes = [C.POINTER(OrcModel), i32p, C.c_int64, C.c_int64, C.c_int64, dp, dp]
    L.orc_hflat_free.argtypes = [vp]
    L.orc_hflat_sizes.argtypes = [vp, i64p]
    L.orc_spmatvec_flat_z.argtypes = [vp, dp, dp]
    L.orc_hflat_dense.argtypes = [vp, dp]
    L.orc_lanc_tridiag_flat.restype = C.c_int
    L.orc_lanc_tridiag_flat.argtypes = [vp, dp, C.c_int, dp, dp, C.c_double]
    L.orc_spmatvec_normal_arrays.argtypes = [C.c_int64, C.c_int64, dp, i64p, i32p, dp, i64p, i32p, dp,
                                             i64p, i32p, dp, dp, dp]
    L.orc_csr_matvec_d.argtypes = [vp, dp, dp]
    L.orc_csr_matvec_z.argtypes = [vp, dp, dp]
    _lib = L
    return L


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class _Csr(C.Structure):
    _fields_ = [("nrow", C.c_int64), ("ncol", C.c_int64), ("nnz", C.c_int64), ("is_complex", C.c_int),
                ("rowptr", C.POINTER(C.c_int64)), ("col", C.POINTER(C.c_int32)), ("val", C.POINTER(C.c_double))]


class _HNormal(C.Structure):
    _fields_ = [("ns", C.c_int), ("nup", C.c_int), ("ndw", C.c_int),
                ("dimup", C.c_int64), ("dimdw", C.c_int64), ("dim", C.c_int64),
                ("mapup", C.POINTER(C.c_int32)), ("mapdw", C.POINTER(C.c_int32)),
                ("hd", C.POINTER(C.c_double)),
                ("up", _Csr), ("dw", _Csr), ("nd", _Csr), ("has_nd", C.c_int)]


class _HFlat(C.Structure):
    _fields_ = [("ns", C.c_int), ("dim", C.c_int64), ("map", C.POINTER(C.c_int32)), ("h", _Csr)]


def _csr_arrays(c: _Csr):
    """Copy a C orc_csr into (rowptr int64, col int32, val float64|complex128)."""
    if c.nrow == 0:
        return np.zeros(1, np.int64), np.zeros(0, np.int32), np.zeros(0, np.float64)
    rowptr = np.ctypeslib.as_array(c.rowptr, shape=(c.nrow + 1,)).copy()
    if c.nnz == 0:
        return rowptr, np.zeros(0, np.int32), np.zeros(0, np.complex128 if c.is_complex else np.float64)
    col = np.ctypeslib.as_array(c.col, shape=(c.nnz,)).copy()
    if c.is_complex:
        v = np.ctypeslib.as_array(c.val, shape=(2 * c.nnz,)).copy().view(np.complex128)
    else:
        v = np.ctypeslib.as_array(c.val, shape=(c.nnz,)).copy()
    return rowptr, col, v


class HNormal:
    """Oracle-built normal-mode sector Hamiltonian (Kronecker pieces)."""

    def __init__(self, model: Model, nup: int, ndw: int):
        self._L = lib()
        self.model = model
        self._s = to_struct(model)
        self._h = self._L.orc_buildh_normal_main(C.byref(self._s), nup, ndw)
        if not self._h:
            raise RuntimeError("oracle: normal build failed")
        hs = C.cast(self._h, C.POINTER(_HNormal)).contents
        self.ns, self.nup, self.ndw = hs.ns, nup, ndw
        self.dimup, self.dimdw, self.dim = hs.dimup, hs.dimdw, hs.dim
        self.mapup = np.ctypeslib.as_array(hs.mapup, shape=(hs.dimup,)).copy()
        self.mapdw = np.ctypeslib.as_array(hs.mapdw, shape=(hs.dimdw,)).copy()
        self.hd = np.ctypeslib.as_array(hs.hd, shape=(hs.dim,)).copy()
        self.up = _csr_arrays(hs.up)
        self.dw = _csr_arrays(hs.dw)
        self.has_nd = bool(hs.has_nd)
        self.nd = _csr_arrays(hs.nd) if self.has_nd else None
        self.dim_el = self.dim
        if model.nph > 0:           # phonon branches: vectors of length dim_el * (Nph + 1)
            self.dim = self.dim_el * (model.nph + 1)

    def matvec(self, v: np.ndarray) -> np.ndarray:
        v = np.ascontiguousarray(v, dtype=np.float64)
        hv = np.empty_like(v)
        if self.model.nph > 0:
            self._L.orc_spmatvec_normal_ph(self._h, C.byref(self._s), _dp(v), _dp(hv))
        else:
            self._L.orc_spmatvec_normal_main(self._h, _dp(v), _dp(hv))
        return hv

    def dense(self) -> np.ndarray:
        if self.model.nph > 0:
            out = np.empty((self.dim, self.dim))
            e = np.zeros(self.dim)
            for j in range(self.dim):
                e[j] = 1.0
                out[:, j] = self.matvec(e)
                e[j] = 0.0
            return out
        out = np.empty((self.dim, self.dim))
        self._L.orc_hnormal_dense(self._h, _dp(out))
        return out

    def lanc_tridiag(self, vin: np.ndarray, nitermax: int, threshold: float = 1e-12):
        v = np.array(vin, dtype=np.float64, copy=True)
        a = np.zeros(nitermax)
        b = np.zeros(nitermax)
        if self.model.nph > 0:
            n = self._L.orc_lanc_tridiag_normal_ph(self._h, C.byref(self._s), _dp(v), nitermax, _dp(a), _dp(b), threshold)
        else:
            n = self._L.orc_lanc_tridiag_normal(self._h, _dp(v), nitermax, _dp(a), _dp(b), threshold)
        return a, b, n

    def close(self):
        if self._h:
            self._L.orc_hnormal_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _HOrbs(C.Structure):
    _fields_ = [("norb", C.c_int), ("nsorb", C.c_int), ("nups", C.c_int * MAXORB), ("ndws", C.c_int * MAXORB),
                ("dims", C.c_int64 * (2 * MAXORB)), ("dim", C.c_int64),
                ("maps", C.POINTER(C.c_int32) * (2 * MAXORB)), ("hd", C.POINTER(C.c_double)),
                ("fac", _Csr * (2 * MAXORB))]


class HOrbs:
    """Oracle-built ed_total_ud=F sector (per-orbital quantum numbers): diagonal + one small factor per
    (orbital, spin); index = [iup_1..iup_Norb, idw_1..idw_Norb], first fastest."""

    def __init__(self, model: Model, nups, ndws):
        self._L = lib()
        self.model = model
        self._s = to_struct(model)
        no = model.norb
        self.nups, self.ndws = tuple(int(x) for x in nups), tuple(int(x) for x in ndws)
        a = (C.c_int * no)(*self.nups)
        b = (C.c_int * no)(*self.ndws)
        self._h = self._L.orc_buildh_normal_orbs(C.byref(self._s), a, b)
        if not self._h:
            raise RuntimeError("oracle: orbs build failed (needs bath_type=normal)")
        hs = C.cast(self._h, C.POINTER(_HOrbs)).contents
        self.norb, self.nsorb, self.dim = hs.norb, hs.nsorb, hs.dim
        self.dims = [int(hs.dims[k]) for k in range(2 * no)]
        self.maps = [np.ctypeslib.as_array(hs.maps[k], shape=(self.dims[k],)).copy() for k in range(2 * no)]
        self.hd = np.ctypeslib.as_array(hs.hd, shape=(max(self.dim, 1),)).copy()[: self.dim]
        self.fac = [_csr_arrays(hs.fac[k]) for k in range(2 * no)]

    def matvec(self, v: np.ndarray) -> np.ndarray:
        v = np.ascontiguousarray(v, dtype=np.float64)
        hv = np.empty_like(v)
        self._L.orc_spmatvec_normal_orbs(self._h, _dp(v), _dp(hv))
        return hv

    def dense(self) -> np.ndarray:
        out = np.empty((self.dim, self.dim))
        self._L.orc_horbs_dense(self._h, _dp(out))
        return out

    def lanc_tridiag(self, vin: np.ndarray, nitermax: int, threshold: float = 1e-12):
        v = np.array(vin, dtype=np.float64, copy=True)
        a = np.zeros(nitermax)
        b = np.zeros(nitermax)
        n = self._L.orc_lanc_tridiag_orbs(self._h, _dp(v), nitermax, _dp(a), _dp(b), threshold)
        return a, b, n

    def close(self):
        if self._h:
            self._L.orc_horbs_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HNormalCmplx:
    """Normal mode with complex algebra (the reference's -D_CMPLX_NORMAL build: the same builder and product with
    complex(8) spH0ups / spH0dws and vectors; ED_NORMAL/stored/H_up.f90:8-22,26-50 insert impHloc(1,1,iorb,jorb) and
    hbath_tmp(1,1,iorb,jorb,kp) as they are).  The builder is linear in the hop amplitudes, so H = S + iA with S the
    restated real build of the real parts and A = 1 (x) A_up + A_dw (x) 1 the same build applied to the imaginary
    parts of the off-diagonal one-body terms (everything else zero).  PARITY UNPINNED against fixtures (the
    reference ships none for this build); cross-checked against the fixture-pinned nonsu2 restatement in
    tests/test_oracle_golden.py."""

    def __init__(self, model: Model, nup: int, ndw: int):
        import copy
        self.model = model
        self.S = HNormal(model, nup, ndw)
        mi = copy.deepcopy(model)
        hl = np.asarray(model.hloc)
        im = np.zeros_like(hl)
        for a in range(model.norb):
            for b in range(model.norb):
                if a != b:
                    im[:, :, a, b] = hl[:, :, a, b].imag
        mi.hloc = im
        mi.hfmode, mi.xmu = False, 0.0
        mi.uloc = tuple([0.0] * model.norb)
        mi.ust = mi.jh = mi.jx = mi.jp = 0.0
        for name in ("be", "bv", "vr", "vg"):
            x = getattr(mi, name, None)
            if x is not None:
                setattr(mi, name, np.zeros_like(np.asarray(x)))
        if getattr(model, "hb", None) is not None:
            hb = np.asarray(model.hb)
            hbi = np.zeros_like(hb)
            for a in range(model.norb):
                for b in range(model.norb):
                    if a != b:
                        hbi[:, :, a, b, :] = hb[:, :, a, b, :].imag
            mi.hb = hbi
        self.A = HNormal(mi, nup, ndw)
        assert not self.A.has_nd and np.max(np.abs(self.A.hd)) == 0.0
        self.dim, self.dimup, self.dimdw = self.S.dim, self.S.dimup, self.S.dimdw

    def matvec(self, v: np.ndarray) -> np.ndarray:
        v = np.asarray(v, dtype=np.complex128)
        re, im = np.ascontiguousarray(v.real), np.ascontiguousarray(v.imag)
        return (self.S.matvec(re) - self.A.matvec(im)) + 1j * (self.S.matvec(im) + self.A.matvec(re))

    def dense(self) -> np.ndarray:
        return self.S.dense() + 1j * self.A.dense()

    def lanc_tridiag(self, vin: np.ndarray, nitermax: int, threshold: float = 1e-12):
        """sp_lanc_tridiag on complex vectors (the recurrence of lanc_tridiag in edipack_oracle.c, w = 2)."""
        v = np.array(vin, dtype=np.complex128)
        a, b = np.zeros(nitermax), np.zeros(nitermax)
        vout = np.zeros_like(v)
        done, b_ = 0, 0.0
        for it in range(1, nitermax + 1):
            if it == 1:
                nrm = np.sqrt(np.vdot(v, v).real)
                if nrm == 0.0:
                    break
                v = v / nrm
            else:
                v, vout = vout / b_, -b_ * v
            vout = vout + self.matvec(v)
            a_ = np.vdot(v, vout).real
            vout = vout - a_ * v
            b_ = np.sqrt(np.vdot(vout, vout).real)
            a[it - 1] = a_
            done = it
            if abs(b_) < threshold:
                break
            if it < nitermax:
                b[it] = b_
        return a, b, done


def direct_matvec_nonsu2(model: Model, ntot: int, v: np.ndarray) -> np.ndarray:
    """directMatVec_nonsu2_main (ED_NONSU2/ED_HAMILTONIAN_NONSU2_DIRECT_HxV.f90:22-126): the elements regenerated row by
    row and applied to v; nothing stored."""
    L = lib()
    s = to_struct(model)
    v = np.ascontiguousarray(v, dtype=np.complex128)
    hv = np.empty_like(v)
    if L.orc_directmatvec_nonsu2_main(C.byref(s), ntot, _dp(v.view(np.float64)), _dp(hv.view(np.float64))):
        raise RuntimeError("oracle: direct nonsu2 product: unsupported terms")
    return hv


class DirectNonsu2:
    """The on-the-fly nonsu2 product on a fixed sector, threaded over row ranges (bench.py's cpu_baseline for the on-the-fly
    workload: the reference's MPI ranks each take a range of rows, directMatVec_MPI_nonsu2_main)."""

    def __init__(self, model: Model, ntot: int):
        self._L = lib()
        self._s = to_struct(model)
        ns = self._L.orc_ns(C.byref(self._s))
        self._L.orc_build_sector_nonsu2.restype = C.c_int64
        self.dim = int(self._L.orc_build_sector_nonsu2(ns, ntot, None))
        self.map = np.zeros(max(self.dim, 1), dtype=np.int32)
        self._L.orc_build_sector_nonsu2(ns, ntot, self.map.ctypes.data_as(C.POINTER(C.c_int32)))

    def matvec(self, v: np.ndarray, hv: np.ndarray, threads: int = 1, rows: int | None = None) -> None:
        """hv[:rows] = (H v)[:rows] (rows=None: all); ctypes releases the GIL, so Python threads run the ranges in parallel"""
        import threading
        n = self.dim if rows is None else min(rows, self.dim)
        mp = self.map.ctypes.data_as(C.POINTER(C.c_int32))
        vp_, hp_ = _dp(v.view(np.float64)), _dp(hv.view(np.float64))
        bounds = [n * t // threads for t in range(threads + 1)]
        if threads == 1:
            self._L.orc_directmatvec_nonsu2_rows(C.byref(self._s), mp, self.dim, 0, n, vp_, hp_)
            return
        ts = [threading.Thread(target=self._L.orc_directmatvec_nonsu2_rows,
                               args=(C.byref(self._s), mp, self.dim, bounds[t], bounds[t + 1], vp_, hp_)) for t in range(threads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()


class HFlat:
    """Oracle-built flat-CSR sector Hamiltonian (superc: sector=Sz, nonsu2: sector=Ntot)."""

    def __init__(self, model: Model, sector: int, twojz: int | None = None):
        self._L = lib()
        self.model = model
        self._s = to_struct(model)
        if model.ed_mode == "superc":
            self._h = self._L.orc_buildh_superc_main(C.byref(self._s), sector)
        elif model.ed_mode == "nonsu2" and twojz is not None:       # Jz_basis=T sector (Ntot, twoJz)
            self._h = self._L.orc_buildh_nonsu2_jz(C.byref(self._s), sector, twojz)
        elif model.ed_mode == "nonsu2":
            self._h = self._L.orc_buildh_nonsu2_main(C.byref(self._s), sector)
        else:
            raise ValueError(model.ed_mode)
        if not self._h:
            raise RuntimeError("oracle: flat build failed / unsupported terms")
        hs = C.cast(self._h, C.POINTER(_HFlat)).contents
        self.ns, self.dim = hs.ns, hs.dim
        self.map = np.ctypeslib.as_array(hs.map, shape=(hs.dim,)).copy() if hs.dim else np.zeros(0, np.int32)
        self.csr = _csr_arrays(hs.h)
        self.dim_el = self.dim
        if model.nph > 0:           # phonon branches: vectors of dim_el * (Nph + 1) elements
            self.dim = self.dim_el * (model.nph + 1)

    def matvec(self, v: np.ndarray) -> np.ndarray:
        v = np.ascontiguousarray(v, dtype=np.complex128)
        hv = np.empty_like(v)
        if self.model.nph > 0:
            self._L.orc_spmatvec_flat_ph(self._h, C.byref(self._s), _dp(v.view(np.float64)), _dp(hv.view(np.float64)))
        else:
            self._L.orc_spmatvec_flat_z(self._h, _dp(v.view(np.float64)), _dp(hv.view(np.float64)))
        return hv

    def dense(self) -> np.ndarray:
        if self.model.nph > 0:
            out = np.empty((self.dim, self.dim), dtype=np.complex128)
            e = np.zeros(self.dim, dtype=np.complex128)
            for j in range(self.dim):
                e[j] = 1.0
                out[:, j] = self.matvec(e)
                e[j] = 0.0
            return out
        out = np.empty((self.dim, self.dim), dtype=np.complex128)
        self._L.orc_hflat_dense(self._h, _dp(out.view(np.float64)))
        return out

    def lanc_tridiag(self, vin: np.ndarray, nitermax: int, threshold: float = 1e-12):
        v = np.array(vin, dtype=np.complex128, copy=True)
        a = np.zeros(nitermax)
        b = np.zeros(nitermax)
        if self.model.nph > 0:
            n = self._L.orc_lanc_tridiag_flat_ph(self._h, C.byref(self._s), _dp(v.view(np.float64)), nitermax, _dp(a),
                                                 _dp(b), threshold)
        else:
            n = self._L.orc_lanc_tridiag_flat(self._h, _dp(v.view(np.float64)), nitermax, _dp(a), _dp(b), threshold)
        return a, b, n

    def close(self):
        if self._h:
            self._L.orc_hflat_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def normal_matvec_arrays(dimup, dimdw, hd, up, dw, nd, v, hv=None):
    """spMatVec_normal_main (reference loop order) on caller-owned arrays; nd may be None."""
    L = lib()
    i64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))  # noqa: E731
    i32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))  # noqa: E731
    if hv is None:
        hv = np.empty_like(v)
    ndargs = (i64(nd[0]), i32(nd[1]), _dp(nd[2])) if nd is not None else (None, None, None)
    L.orc_spmatvec_normal_arrays(dimup, dimdw, _dp(hd), i64(up[0]), i32(up[1]), _dp(up[2]),
                                 i64(dw[0]), i32(dw[1]), _dp(dw[2]), *ndargs, _dp(v), _dp(hv))
    return hv


def normal_matvec_arrays_mt(dimup, dimdw, hd, up, dw, nd, v, hv, nthreads):
    """spMatVec_mpi_normal_main's decomposition (blocks of down indices) on `nthreads` OpenMP threads."""
    L = lib()
    i64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))  # noqa: E731
    i32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))  # noqa: E731
    ndargs = (i64(nd[0]), i32(nd[1]), _dp(nd[2])) if nd is not None else (None, None, None)
    L.orc_spmatvec_normal_arrays_mt(C.c_int64(dimup), C.c_int64(dimdw), _dp(hd), i64(up[0]), i32(up[1]), _dp(up[2]),
                                    i64(dw[0]), i32(dw[1]), _dp(dw[2]), *ndargs, _dp(v), _dp(hv), C.c_int(nthreads))
    return hv


def csr_matvec_z_mt(rowptr, col, val, x, y, nthreads):
    """row-partitioned complex CSR product on `nthreads` OpenMP threads (arrays must be contiguous)."""
    L = lib()
    c = _Csr(rowptr.shape[0] - 1, x.shape[0], col.shape[0], 1, rowptr.ctypes.data_as(C.POINTER(C.c_int64)),
             col.ctypes.data_as(C.POINTER(C.c_int32)), val.view(np.float64).ctypes.data_as(C.POINTER(C.c_double)))
    L.orc_csr_matvec_z_mt(C.byref(c), _dp(x.view(np.float64)), _dp(y.view(np.float64)), C.c_int(nthreads))
    return y


# ----------------------------------------------------------------------------
# generic CSR products in the reference loop order (checker for the flat kernels)
# ----------------------------------------------------------------------------
def csr_matvec(rowptr: np.ndarray, col: np.ndarray, val: np.ndarray, x: np.ndarray) -> np.ndarray:
    L = lib()
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int32)
    cplx = np.iscomplexobj(val) or np.iscomplexobj(x)
    dt = np.complex128 if cplx else np.float64
    val = np.ascontiguousarray(val, dtype=dt)
    x = np.ascontiguousarray(x, dtype=dt)
    nrow = rowptr.shape[0] - 1
    y = np.empty(nrow, dtype=dt)
    c = _Csr(nrow, x.shape[0], col.shape[0], int(cplx),
             rowptr.ctypes.data_as(C.POINTER(C.c_int64)), col.ctypes.data_as(C.POINTER(C.c_int32)),
             val.view(np.float64).ctypes.data_as(C.POINTER(C.c_double)))
    fn = L.orc_csr_matvec_z if cplx else L.orc_csr_matvec_d
    fn(C.byref(c), _dp(x.view(np.float64)), _dp(y.view(np.float64)))
    return y


# ----------------------------------------------------------------------------
# reference-fixture level checks
# ----------------------------------------------------------------------------
def sectors(m: Model):
    """Sector labels scanned by diagonalize_impurity (ED_SETUP.f90:137-165)."""
    ns = m.ns
    if m.ed_mode == "normal":
        return [(nu, nd) for nu in range(ns + 1) for nd in range(ns + 1)]
    if m.ed_mode == "superc":
        return list(range(-ns, ns + 1))
    return list(range(0, 2 * ns + 1))


def hbuild(m: Model, sector):
    if m.ed_mode == "normal":
        return HNormal(m, sector[0], sector[1])
    return HFlat(m, sector)


def ground_state(m: Model, gs_threshold: float = 1e-9):
    """Lowest energy over all sectors + T=0 impurity observables averaged over the
    degenerate ground-state manifold (dens, docc): what evals/dens/docc.check pin.

    Spectrum by dense LAPACK as the reference does below lanc_dim_threshold
    (ED_NORMAL/ED_DIAG_NORMAL.f90:226-236); observables as in
    ED_NORMAL/ED_OBSERVABLES_NORMAL.f90 (dens=<nup+ndw>, docc=<nup ndw>)."""
    found = []  # (energy, sector, vector, H object)
    for sec in sectors(m):
        h = hbuild(m, sec)
        if h.dim == 0:
            continue
        w, v = np.linalg.eigh(h.dense())
        found.append((w, v, sec, h))
    e0 = min(w[0] for w, _, _, _ in found)
    dens = np.zeros(m.norb)
    docc = np.zeros(m.norb)
    ngs = 0
    ns = m.ns
    for w, v, sec, h in found:
        for k in range(len(w)):
            if w[k] - e0 > gs_threshold:
                break
            ngs += 1
            p = np.abs(v[:, k]) ** 2
            if m.ed_mode == "normal":
                iup = np.arange(h.dim) % h.dimup
                idw = np.arange(h.dim) // h.dimup
                mu, md = h.mapup[iup], h.mapdw[idw]
            else:
                mu, md = h.map & ((1 << ns) - 1), h.map >> ns
            for io in range(m.norb):
                nu = (mu >> io) & 1
                nd = (md >> io) & 1
                dens[io] += np.sum(p * (nu + nd))
                docc[io] += np.sum(p * (nu * nd))
    return e0, dens / ngs, docc / ngs, ngs
