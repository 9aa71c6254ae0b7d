"""The host builders of libedigpu (csrc/host_build.cpp) against the oracle, on the CPU: the sector images the kernels
consume (explicit arrays and factored tables) as dense matrices vs the oracle's dense H of the same seeded model.
Covers the terms of ED_NORMAL/stored/ that only some inputs switch on: coulomb_sundry (H_sundry.f90), spin_field z
(H_local.f90:38-42), exc_field(1),(4) (H_up.f90:87-104, H_dw.f90).  tests/host_image.cpp is the test-only shim."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from edipack_amd import capi
from oracle import oracle as O
from tests.common import make_models

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# U cd_i cd_j c_k c_l lines (0-based orbital, spin 0 up / 1 down), Hermitian-conjugate pairs so that H stays symmetric
SUNDRY3 = [
    (0.40, (0, 0), (1, 1), (2, 1), (1, 0)), (0.40, (1, 0), (2, 1), (1, 1), (0, 0)),     # mixed-spin correlated hop
    (0.25, (0, 0), (2, 0), (2, 0), (1, 0)), (0.25, (1, 0), (2, 0), (2, 0), (0, 0)),     # same-spin, density assisted
    (-0.30, (0, 1), (1, 1), (1, 1), (0, 1)),                                            # n_0dw n_1dw (diagonal)
    (0.15, (2, 0), (0, 1), (1, 1), (1, 0)), (0.15, (1, 0), (1, 1), (0, 1), (2, 0)),
]
SUNDRY2 = [(0.35, (0, 0), (1, 1), (0, 1), (1, 0)), (0.35, (1, 0), (0, 1), (1, 1), (0, 0)),
           (0.2, (0, 0), (0, 1), (0, 1), (0, 0))]


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    so = str(tmp_path_factory.mktemp("host_image") / "host_image.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "edipack_amd", "csrc"), "-o", so,
                           os.path.join(ROOT, "tests", "host_image.cpp"),
                           os.path.join(ROOT, "edipack_amd", "csrc", "host_build.cpp")])
    lib = C.CDLL(so)
    lib.host_image_error.restype = C.c_char_p
    return lib


def _host_dense(lib, pm, nup, ndw, form, dim):
    out = np.zeros((dim, dim))
    m = pm.to_c()
    rc = lib.host_normal_dense(C.byref(m), nup, ndw, form, out.ctypes.data_as(C.c_void_p), C.c_int64(dim))
    assert rc == 0, lib.host_image_error().decode()
    return out


CASES = [
    # bath, norb, nbath, sector, extra
    ("hybrid", 3, 3, (3, 3), dict(sundry=SUNDRY3)),
    ("hybrid", 3, 2, (2, 3), dict(sundry=SUNDRY3, jxp=0.0)),
    ("normal", 2, 2, (3, 2), dict(sundry=SUNDRY2)),
    ("normal", 2, 2, (2, 2), dict(spin_field=np.array([[0.3, 0.1, 0.2], [0.0, 0.0, -0.15]]))),
    ("hybrid", 3, 2, (2, 2), dict(exc_field=np.array([0.12, 0.5, 0.5, 0.07]))),
    ("replica", 2, 2, (3, 3), dict(exc_field=np.array([0.1, 0.0, 0.0, -0.2]), sundry=SUNDRY2,
                                   spin_field=np.array([[0, 0, 0.1], [0, 0, 0.2]]))),
    ("normal", 1, 4, (2, 3), dict(sundry=[(0.6, (0, 0), (0, 1), (0, 1), (0, 0))])),
    ("hybrid", 3, 3, (3, 3), dict()),
]


@pytest.mark.parametrize("bath,norb,nbath,sec,extra", CASES)
def test_normal_builder_images_equal_oracle(shim, bath, norb, nbath, sec, extra):
    extra = dict(extra)
    jxp = extra.pop("jxp", 0.25)
    om, pm = make_models("normal", bath, norb, nbath, seed=3, jxp=jxp, **extra)
    h = O.HNormal(om, *sec)
    ref = h.dense()
    assert np.abs(ref - ref.T).max() < 1e-14
    if extra:
        om0, _ = make_models("normal", bath, norb, nbath, seed=3, jxp=jxp)
        assert np.abs(ref - O.HNormal(om0, *sec).dense()).max() > 1e-3   # the switch does something
    for form in (0, 1):
        got = _host_dense(shim, pm, sec[0], sec[1], form, h.dim)
        assert np.abs(got - ref).max() < 1e-13, (form, np.abs(got - ref).max())


def test_sundry_lines_that_flip_a_spin_are_refused(shim):
    """H_sundry.f90:24-34 STOPs on a line that changes N_up - N_dw; the builder returns an error."""
    _, pm = make_models("normal", "normal", 2, 2, seed=1, sundry=[(0.3, (0, 0), (1, 0), (1, 1), (0, 1))])
    m = pm.to_c()
    out = np.zeros((36, 36))
    rc = shim.host_normal_dense(C.byref(m), 2, 2, 0, out.ctypes.data_as(C.c_void_p), C.c_int64(36))
    assert rc == 1 and "change the total spin" in shim.host_image_error().decode()
    _, pm = make_models("normal", "normal", 2, 2, seed=1, sundry=[(0.3, (0, 0), (2, 1), (1, 1), (0, 0))])
    m = pm.to_c()
    rc = shim.host_normal_dense(C.byref(m), 2, 2, 0, out.ctypes.data_as(C.c_void_p), C.c_int64(36))
    assert rc == 1 and "out of range" in shim.host_image_error().decode()


# a spin-flipping line and its conjugate (nonsu2 only: N is conserved, Sz is not)
SUNDRY_FLIP = [(0.2, (0, 0), (1, 0), (1, 1), (0, 0)), (0.2, (0, 0), (1, 1), (1, 0), (0, 0))]
FLAT_CASES = [
    # mode, bath, norb, nbath, sector, extra
    ("superc", "hybrid", 3, 1, 0, dict(sundry=SUNDRY3)),
    ("superc", "normal", 2, 2, 1, dict(sundry=SUNDRY2)),
    ("superc", "replica", 2, 2, -1, dict(sundry=SUNDRY2)),
    ("nonsu2", "hybrid", 3, 1, 4, dict(sundry=SUNDRY3)),
    ("nonsu2", "normal", 2, 2, 5, dict(sundry=SUNDRY2 + SUNDRY_FLIP)),
    ("nonsu2", "normal", 2, 2, 6, dict(exc_field=np.array([0.12, 0.3, -0.2, 0.07]))),
    ("nonsu2", "hybrid", 3, 1, 4, dict(spin_field=np.array([[0.3, 0.1, 0.2], [0.0, -0.25, -0.15], [0.05, 0.0, 0.0]]))),
    ("nonsu2", "replica", 2, 2, 5, dict(exc_field=np.array([0.1, 0.2, 0.0, -0.2]), sundry=SUNDRY2 + SUNDRY_FLIP,
                                        spin_field=np.array([[0.0, 0.4, 0.1], [0.2, 0.0, 0.2]]))),
]


@pytest.mark.parametrize("mode,bath,norb,nbath,sec,extra", FLAT_CASES)
def test_flat_builder_images_equal_oracle(shim, mode, bath, norb, nbath, sec, extra):
    """coulomb_sundry in both flat modes (Hint.f90:127-181), exc_field / spin_field in nonsu2 (Himp.f90:113-296): the
    stored rows and the on-the-fly term list (levels met twice included) against the oracle's matrix."""
    om, pm = make_models(mode, bath, norb, nbath, seed=5, **extra)
    h = O.HFlat(om, sec)
    ref = h.dense()
    assert np.abs(ref - ref.conj().T).max() < 1e-14
    om0, _ = make_models(mode, bath, norb, nbath, seed=5)
    assert np.abs(ref - O.HFlat(om0, sec).dense()).max() > 1e-3           # the switch does something
    m = pm.to_c()
    for fn in (shim.host_flat_dense, shim.host_direct_dense):
        out = np.zeros((h.dim, h.dim), dtype=np.complex128)
        rc = fn(C.byref(m), sec, out.ctypes.data_as(C.c_void_p), C.c_int64(h.dim))
        assert rc == 0, shim.host_image_error().decode()
        assert np.abs(out - ref).max() < 1e-13, np.abs(out - ref).max()


def test_flat_builders_refusals(shim):
    """superc: the reference's files hold no spin_field / exc_field terms, and a coulomb_sundry line that changes Sz
    STOPs there ("impossible operator"): both are errors here, in the builders and in the oracle."""
    buf = np.zeros(8)
    for extra, msg in ((dict(spin_field=np.array([[0, 0, 0.1], [0, 0, 0.0]])), "no terms in the superc"),
                       (dict(exc_field=np.array([0.1, 0, 0, 0])), "no terms in the superc"),
                       (dict(sundry=SUNDRY_FLIP), "changes Sz")):
        om, pm = make_models("superc", "normal", 2, 1, seed=1, **extra)
        m = pm.to_c()
        assert shim.host_flat_dense(C.byref(m), 0, buf.ctypes.data_as(C.c_void_p), C.c_int64(1)) == 1
        assert msg in shim.host_image_error().decode()
        assert shim.host_direct_refuses(C.byref(m), 0) == 1
        if "sundry" not in extra:           # (the oracle aborts like the reference on a line that leaves the sector)
            with pytest.raises(Exception):
                O.HFlat(om, 0)
    _, pm = make_models("nonsu2", "normal", 2, 1, seed=1, sundry=[(0.3, (0, 0), (2, 1), (1, 1), (0, 0))])
    m = pm.to_c()
    assert shim.host_flat_dense(C.byref(m), 3, buf.ctypes.data_as(C.c_void_p), C.c_int64(1)) == 1
    assert "out of range" in shim.host_image_error().decode()


def test_model_struct_sizes_agree(shim):
    assert C.sizeof(capi.EdigpuModel) == capi.lib().edigpu_model_sizeof()


# ---- hand-over images: edigpu_normal_create recovers the factored tables from the reference's arrays ----
def _handover(shim, ho, first, cnt):
    du, dd = ho.dimup, ho.dimdw
    r0, r1 = first * du, (first + cnt) * du
    ndr, ndc, ndv = ho.nd if ho.has_nd else (np.zeros(ho.dim + 1, np.int64), np.zeros(0, np.int32), np.zeros(0))
    rp = np.ascontiguousarray(ndr[r0:r1 + 1] - ndr[r0], dtype=np.int64)
    col = np.ascontiguousarray(ndc[ndr[r0]:ndr[r1]], dtype=np.int32)
    val = np.ascontiguousarray(ndv[ndr[r0]:ndr[r1]], dtype=np.float64)
    hd = np.ascontiguousarray(ho.hd[r0:r1])
    out = np.zeros((r1 - r0, ho.dim))
    ncls = C.c_int(0)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    nt = shim.host_handover_dense(C.c_int64(du), C.c_int64(dd), C.c_int64(first), C.c_int64(cnt), P(hd), P(rp), P(col),
                                  P(val), P(out), C.byref(ncls))
    ref = np.zeros_like(out)
    ref[np.arange(r1 - r0), np.arange(r0, r1)] = hd
    for i in range(r1 - r0):
        for k in range(rp[i], rp[i + 1]):
            ref[i, col[k]] += val[k]
    return nt, ncls.value, out, ref


@pytest.mark.parametrize("bath,norb,nbath,sec,jxp,extra,nterms,ncls", [
    ("normal", 2, 2, (3, 3), 0.25, {}, 2, 4),             # Jx = Jp: one term per down move
    ("hybrid", 3, 2, (2, 3), 0.25, {}, 6, 8),
    ("hybrid", 3, 2, (3, 2), 0.0, {}, 0, 8),              # no Hnd
    ("normal", 1, 4, (2, 3), 0.0, {}, 0, 2),
    ("hybrid", 3, 3, (3, 3), 0.0, dict(sundry=SUNDRY3), None, 8),
    ("replica", 2, 2, (3, 3), 0.25, dict(spin_field=np.array([[0, 0, 0.1], [0, 0, 0.2]])), 2, 4),
])
def test_handover_arrays_factor_exactly(shim, bath, norb, nbath, sec, jxp, extra, nterms, ncls):
    om, _ = make_models("normal", bath, norb, nbath, seed=12, jxp=jxp, **extra)
    ho = O.HNormal(om, *sec)
    for first, cnt in ((0, ho.dimdw), (1, ho.dimdw - 2), (ho.dimdw // 2, 1)):
        nt, nc, got, ref = _handover(shim, ho, first, cnt)
        assert nt >= 0, "the arrays of an impurity model must factor"
        if nterms is not None and cnt == ho.dimdw:
            assert nt <= nterms and nc <= ncls      # equal operators merge (Jx = Jp: both down moves share one)
        # Hnd entries are reproduced exactly, the diagonal within 8 ulp of max|Hd|
        off = ~np.eye(ho.dim, dtype=bool)[first * ho.dimup:(first + cnt) * ho.dimup]
        assert np.array_equal(got[off], ref[off])
        assert np.abs(got - ref).max() <= 8 * np.finfo(float).eps * np.abs(ho.hd).max()


def test_handover_arrays_that_do_not_factor_fall_back(shim):
    """A diagonal with no class structure and an Hnd with unrelated entries: the attempt must say no."""
    om, _ = make_models("normal", "normal", 2, 2, seed=12)
    ho = O.HNormal(om, 3, 3)
    rng = np.random.default_rng(0)

    class Fake:
        pass
    f = Fake()
    f.dimup, f.dimdw, f.dim, f.has_nd = ho.dimup, ho.dimdw, ho.dim, True
    f.hd = rng.standard_normal(ho.dim)                 # every dw row its own profile: > 64 classes? (DimDw = 20: no)
    f.nd = ho.nd
    nt, nc, got, ref = _handover(shim, f, 0, ho.dimdw)
    assert nt >= 0 and nc == ho.dimdw                  # 20 classes: still representable, and exact
    assert np.abs(got - ref).max() <= 8 * np.finfo(float).eps * np.abs(f.hd).max()
    # random Hnd values: every (idw -> jdw) pair its own up-operator, more than 16 terms
    ndr, ndc, ndv = ho.nd
    f.hd = ho.hd
    f.nd = (ndr, ndc, rng.standard_normal(ndv.shape))
    assert _handover(shim, f, 0, ho.dimdw)[0] == -1


# ---- nonsu2 sectors of JZ_BASIS=T ----
@pytest.mark.parametrize("nbath,ntot,twojz", [(1, 6, 0), (1, 5, 1), (1, 3, -3), (2, 9, 1), (2, 4, 0)])
def test_jz_sector_builder_equals_oracle(shim, nbath, ntot, twojz):
    from edipack_amd.hamiltonian import sector_map_jz
    from tests.common import make_jz_models
    om, pm = make_jz_models(nbath, seed=4)
    ho = O.HFlat(om, ntot, twojz=twojz)
    assert ho.dim > 0
    assert np.array_equal(sector_map_jz(pm, ntot, twojz), ho.map)          # bit-exact index data
    out = np.zeros((ho.dim, ho.dim, 2))
    m = pm.to_c()
    rc = shim.host_flat_jz_dense(C.byref(m), ntot, twojz, out.ctypes.data_as(C.c_void_p), C.c_int64(ho.dim))
    assert rc == 0, shim.host_image_error().decode()
    got = out[..., 0] + 1j * out[..., 1]
    assert np.abs(got - ho.dense()).max() < 1e-13
    # the on-the-fly description the device forms evaluate (two-table rank inside an (occupation, Lz) class)
    out2 = np.zeros((ho.dim, ho.dim, 2))
    rc = shim.host_direct_jz_dense(C.byref(m), ntot, twojz, out2.ctypes.data_as(C.c_void_p), C.c_int64(ho.dim))
    assert rc == 0, shim.host_image_error().decode()
    assert np.abs(out2[..., 0] + 1j * out2[..., 1] - ho.dense()).max() < 1e-13


def test_jz_sector_builder_refusals(shim):
    from tests.common import make_jz_models
    _, pm = make_jz_models(1, seed=4)
    buf = np.zeros(8)
    pm.jp = 0.3          # pair hopping moves a pair between orbitals of different Lz: Jz is not conserved
    m = pm.to_c()
    n = len(O.HFlat(make_jz_models(1, seed=4)[0], 6, twojz=0).map)
    big = np.zeros(2 * n * n)
    assert shim.host_flat_jz_dense(C.byref(m), 6, 0, big.ctypes.data_as(C.c_void_p), C.c_int64(n)) == 1
    assert "does not conserve Jz" in shim.host_image_error().decode()
    assert shim.host_direct_jz_dense(C.byref(m), 6, 0, big.ctypes.data_as(C.c_void_p), C.c_int64(n)) == 1
    assert "does not conserve Jz" in shim.host_image_error().decode()
    _, p2 = make_models("nonsu2", "normal", 2, 2, seed=1)                   # two orbitals
    m = p2.to_c()
    assert shim.host_flat_jz_dense(C.byref(m), 4, 0, buf.ctypes.data_as(C.c_void_p), C.c_int64(1)) == 1
    assert "Norb = 3" in shim.host_image_error().decode()
    _, p3 = make_models("nonsu2", "normal", 3, 2, seed=1)                   # orbital-major bath levels
    m = p3.to_c()
    assert shim.host_flat_jz_dense(C.byref(m), 4, 0, buf.ctypes.data_as(C.c_void_p), C.c_int64(1)) == 1
    assert "iorb + Norb * ibath" in shim.host_image_error().decode()


# ---- _CMPLX_NORMAL as one real sector on the doubled up index ----
@pytest.mark.parametrize("bath,norb,nbath,sec", [
    ("normal", 2, 2, (3, 2)),
    ("hybrid", 3, 3, (3, 4)),      # Hnd terms of S + three imaginary down-hop terms
    ("replica", 2, 2, (3, 3)),     # complex bath matrices: imaginary hops between bath levels as well
    ("general", 2, 3, (4, 4)),
    ("normal", 1, 3, (2, 2)),      # one orbital: nothing imaginary, H' = S (x) 1
])
def test_cmplx_normal_doubled_real_image_equals_oracle(shim, bath, norb, nbath, sec):
    """build_normal_doubled: H = S + iA read as the real operator [[S, -A], [A, S]] on interleaved (re, im) components
    must be the oracle's complex matrix, entry by entry."""
    from tests.test_gpu_parity import _complexify
    om, pm = make_models("normal", bath, norb, nbath, seed=95)
    _complexify(om, pm, 96)
    ho = O.HNormalCmplx(om, *sec)
    hc = ho.dense()
    n = ho.dim
    out = np.zeros((2 * n, 2 * n))
    nt = C.c_int(0)
    m = pm.to_c()
    rc = shim.host_normal_doubled_dense(C.byref(m), sec[0], sec[1], out.ctypes.data_as(C.c_void_p), C.c_int64(2 * n),
                                        C.byref(nt))
    assert rc == 0, shim.host_image_error().decode()
    # interleaved layout: real index 2 i + c for complex index i
    ref = np.zeros((2 * n, 2 * n))
    ref[0::2, 0::2] = hc.real
    ref[1::2, 1::2] = hc.real
    ref[0::2, 1::2] = -hc.imag
    ref[1::2, 0::2] = hc.imag
    assert np.abs(out - ref).max() < 1e-13 and np.abs(out - out.T).max() < 1e-13
    assert nt.value <= 16


@pytest.mark.parametrize("mode", ["superc", "nonsu2"])
def test_flat_random_sundry_lines(shim, mode):
    """Random coulomb_sundry lines (any of the 4^4 orbital/spin patterns the sector family allows: levels met twice,
    identically vanishing lines, diagonal lines) through the stored rows and the on-the-fly term encoding vs the oracle."""
    rng = np.random.default_rng(2024 if mode == "superc" else 2025)
    norb, nbath = 2, 2
    checked = 0
    for trial in range(40):
        lines = []
        for _ in range(int(rng.integers(1, 5))):
            while True:
                ops = [(int(rng.integers(0, norb)), int(rng.integers(0, 2))) for _ in range(4)]
                dsz = sum((1 if k < 2 else -1) * (1 if sp == 0 else -1) for k, (_, sp) in enumerate(ops))
                if mode == "nonsu2" or dsz == 0:
                    break
            lines.append((float(rng.uniform(-0.5, 0.5)), *ops))
        om, pm = make_models(mode, "hybrid" if trial % 2 else "normal", norb, nbath, seed=100 + trial, sundry=lines)
        ns = om.ns
        sec = int(rng.integers(-1, 2)) if mode == "superc" else int(rng.integers(2, 2 * ns - 1))
        h = O.HFlat(om, sec)
        if h.dim == 0:
            continue
        ref = h.dense()
        m = pm.to_c()
        for fn in (shim.host_flat_dense, shim.host_direct_dense):
            out = np.zeros((h.dim, h.dim), dtype=np.complex128)
            rc = fn(C.byref(m), sec, out.ctypes.data_as(C.c_void_p), C.c_int64(h.dim))
            assert rc == 0, shim.host_image_error().decode()
            assert np.abs(out - ref).max() < 1e-13, (trial, lines, np.abs(out - ref).max())
        checked += 1
    assert checked >= 30
