"""CPU tests: pin the oracle on the reference's own regression fixtures.

tests/golden/reference_checks.json holds the *.check data files of
/root/reference/test/src/<BATH>_<MODE>/ (made by tests/golden/make_reference_checks.py).
The reference asserts them with abs tol 1e-9 (test/src/ASSERTING.f90:74-80); same here.

What is pinned: the three sector-Hamiltonian builders (normal Kronecker pieces, superc and
nonsu2 flat CSR), the sector bases, and the H*v restatements (through dense/H*v consistency).
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.common import make_models, rel_err, replica_golden_models

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_checks.json")))

DIRS = ["NORMAL_NORMAL", "HYBRID_NORMAL", "NORMAL_SUPERC", "HYBRID_SUPERC", "NORMAL_NONSU2", "HYBRID_NONSU2"]
# replica / general baths: the per-replica matrices sum_i lambda_i(k) Hsym_i are formed in the test from the
# basis and lambdas the reference's test programs set (input preparation, outside the H*v path)
REPLICA_DIRS = ["REPLICA_NORMAL", "GENERAL_NORMAL", "REPLICA_SUPERC", "GENERAL_SUPERC", "REPLICA_NONSU2",
                "GENERAL_NONSU2"]


def golden_models(mode, bath, norb, nbath, par):
    """The impurity problem of test/src/<BATH>_<MODE>: deterministic init_dmft_bath start bath,
    Hloc = Delta*sigma_z (normal, superc: ed_normal_normal.f90:60-66) or Mh*Gamma5 (nonsu2:
    ed_normal_nonsu2.f90:76-77), Kanamori couplings from inputED.in."""
    nspin = 2 if mode == "nonsu2" else 1
    hl = np.zeros((nspin, nspin, norb, norb), complex)
    amp = 1.0 if mode == "nonsu2" else 0.5
    for s in range(nspin):
        hl[s, s, 0, 0] = amp
        hl[s, s, 1, 1] = -amp
    om, pm = make_models(mode, bath, norb, nbath, reference_bath=True, **par)
    om.hloc = hl
    pm.hloc = hl
    return om, pm


def _from_dir(name):
    inp = GOLD[name]["input"]
    par = dict(uloc=tuple(inp["ULOC"]), ust=inp["UST"], jh=inp["JH"], jx=inp["JX"], jp=inp["JP"],
               ed_hw_bath=inp["ED_HW_BATH"], deltasc=inp["DELTASC"])
    return inp, par


@pytest.mark.parametrize("name", DIRS)
def test_oracle_reproduces_reference_fixture(name):
    inp, par = _from_dir(name)
    pm_par = {k: v for k, v in par.items() if k not in ("ed_hw_bath", "deltasc")}
    om, _ = golden_models(inp["ED_MODE"], inp["BATH_TYPE"], int(inp["NORB"]), int(inp["NBATH"]), pm_par)
    e0, dens, docc, ngs = O.ground_state(om)
    g = GOLD[name]
    assert abs(e0 - g["evals"][0]) < 1e-9
    # HYBRID_SUPERC has a second state 1.5e-6 above the ground state in the same sector: any
    # eigensolver's vector carries an admixture ~ eps*|H|/gap ~ 1e-9..1e-8 of it, which shows up at
    # first order in dens/docc (the energy is second order and still agrees to 1e-14).
    tol = 5e-8 if name == "HYBRID_SUPERC" else 1e-9
    assert np.max(np.abs(dens - np.array(g["dens"]))) < tol
    assert np.max(np.abs(docc - np.array(g["docc"]))) < tol


@pytest.mark.parametrize("name", ["NORMAL_NORMAL", "HYBRID_NORMAL"])
def test_oracle_tridiag_reproduces_sigma_momenta(name):
    """Sigma_momenta.check: the fixture that goes through tridiag_Hv_sector_normal + sp_lanc_tridiag (one
    tridiagonalisation of up to lanc_ngfiter=200 steps per orbital, channel and ground state) -- pins the
    restated three-term recurrence at the level of the alpha/beta it returns.  Measured agreement: 5e-15."""
    from tests.gf_normal import sigma_momenta_normal
    inp, par = _from_dir(name)
    pm_par = {k: v for k, v in par.items() if k not in ("ed_hw_bath", "deltasc")}
    om, _ = golden_models(inp["ED_MODE"], inp["BATH_TYPE"], int(inp["NORB"]), int(inp["NBATH"]), pm_par)
    O.to_struct(om)   # fills the init_dmft_bath start bath

    def tridiag(sec, v, nl):
        a, b, _ = O.HNormal(om, *sec).lanc_tridiag(v.copy(), nl)
        return a, b

    m = sigma_momenta_normal(om, tridiag, beta=inp["BETA"], ngfiter=int(inp["LANC_NGFITER"]))
    g = np.array(GOLD[name]["Sigma_momenta"]).reshape(m.shape)
    assert np.max(np.abs(m - g) / np.abs(g)) < 1e-11


@pytest.mark.parametrize("name", ["REPLICA_NORMAL", "GENERAL_NORMAL"])
def test_oracle_tridiag_reproduces_replica_sigma_momenta(name):
    """Sigma_momenta.check of the replica / general normal-mode directories: diagonal and mixed channels (the bath
    couples the orbitals), G0 from the replica matrices -- the tridiagonalisation of sectors built with inter-orbital
    bath hops, pinned like the normal / hybrid ones."""
    from tests.gf_normal import sigma_momenta_normal
    g = GOLD[name]
    om, _ = replica_golden_models(g["input"])

    def tridiag(sec, v, nl):
        a, b, _ = O.HNormal(om, *sec).lanc_tridiag(v.copy(), nl)
        return a, b

    m = sigma_momenta_normal(om, tridiag, beta=g["input"]["BETA"], ngfiter=int(g["input"]["LANC_NGFITER"]))
    gold = np.array(g["Sigma_momenta"]).reshape(m.shape)
    assert np.max(np.abs(m - gold) / np.abs(gold)) < 1e-10


@pytest.mark.parametrize("name", REPLICA_DIRS)
def test_oracle_reproduces_replica_general_fixture(name):
    g = GOLD[name]
    om, _ = replica_golden_models(g["input"])
    e0, dens, docc, ngs = O.ground_state(om)
    assert abs(e0 - g["evals"][0]) < 1e-9
    assert np.max(np.abs(dens - np.array(g["dens"]))) < 1e-9
    assert np.max(np.abs(docc - np.array(g["docc"]))) < 1e-9


@pytest.mark.parametrize("mode,bath,norb,nbath,sec", [
    ("normal", "normal", 2, 2, (3, 3)),
    ("normal", "hybrid", 3, 3, (3, 2)),
    ("normal", "general", 2, 2, (3, 3)),
    ("superc", "replica", 2, 2, 0),
    ("nonsu2", "general", 2, 2, 6),
    ("superc", "normal", 2, 2, 0),
    ("nonsu2", "hybrid", 2, 3, 5),
])
def test_oracle_matvec_equals_dense(mode, bath, norb, nbath, sec):
    """spMatVec_* restatement == dense dump of the same stored matrices (the reference's own
    stored-vs-dense consistency), and H is Hermitian."""
    om, _ = make_models(mode, bath, norb, nbath, seed=2)
    h = O.hbuild(om, sec)
    d = h.dense()
    assert np.allclose(d, d.conj().T, atol=1e-14)
    rng = np.random.default_rng(0)
    v = rng.standard_normal(h.dim) + (1j * rng.standard_normal(h.dim) if mode != "normal" else 0)
    assert rel_err(h.matvec(v), d @ v) < 1e-13


def test_oracle_sector_maps():
    """build_sector: ascending maps with the right popcounts and dimensions (ED_SECTOR.f90:217-281)."""
    om, _ = make_models("normal", "normal", 2, 2, seed=0)
    h = O.HNormal(om, 2, 4)
    assert h.dimup == 15 and h.dimdw == 15
    assert np.all(np.diff(h.mapup) > 0) and all(bin(int(x)).count("1") == 2 for x in h.mapup)
    assert all(bin(int(x)).count("1") == 4 for x in h.mapdw)
    om, _ = make_models("superc", "normal", 2, 2, seed=0)
    h = O.HFlat(om, 0)
    assert h.dim == 924 and np.all(np.diff(h.map) > 0)
    ns = om.ns
    assert all(bin(int(x) & 63).count("1") == bin(int(x) >> ns).count("1") for x in h.map)


def test_oracle_lanczos_tridiag_reproduces_spectrum():
    """Full-length tridiagonalisation of a small sector reproduces the dense spectrum edge and the
    resolvent <v|(z-H)^-1|v> (the quantity the GF builder derives from alanc/blanc)."""
    om, _ = make_models("normal", "normal", 2, 2, seed=1)
    h = O.HNormal(om, 3, 3)
    d = h.dense()
    w, z = np.linalg.eigh(d)
    v = np.random.default_rng(3).standard_normal(h.dim)
    a, b, n = h.lanc_tridiag(v, 120)
    t = np.diag(a[:n]) + np.diag(b[1:n], 1) + np.diag(b[1:n], -1)
    assert abs(np.linalg.eigvalsh(t)[0] - w[0]) < 1e-10
    vn = v / np.linalg.norm(v)
    zz = 40.0 + 0.1j      # outside the spectrum: the truncated fraction has converged
    exact = np.sum(np.abs(z.T @ vn) ** 2 / (zz - w))
    g = 0.0
    for k in range(n - 1, -1, -1):
        b2 = b[k + 1] ** 2 if k + 1 < n else 0.0
        g = 1.0 / (zz - a[k] - b2 * g)
    assert abs(g - exact) / abs(exact) < 1e-10


def test_oracle_orbs_spectrum_equals_total_ud():
    """ed_total_ud=F restatement (no reference fixture exists for it) pinned through the fixture-pinned
    ed_total_ud=T oracle: the spectrum of a (Nup,Ndw) sector is the union of the spectra of its
    orbital-resolved sectors, up to the Hartree constant in which the reference's two builders differ
    (stored/H_local.f90:63-64 adds 0.5 per orbital pair, stored/Orbs/H_local.f90:61-62 adds 0.25)."""
    import itertools
    om, _ = make_models("normal", "normal", 2, 2, seed=5, jxp=0.0)
    hl = np.zeros_like(om.hloc)
    for a in range(2):
        hl[0, 0, a, a] = om.hloc[0, 0, a, a].real
    om.hloc = hl
    nso = om.nbath + 1
    shift = 0.25 * om.ust + 0.25 * (om.ust - om.jh)
    for nup, ndw in [(3, 3), (2, 4), (1, 1)]:
        w_t = np.linalg.eigvalsh(O.HNormal(om, nup, ndw).dense())
        ws = []
        for nups in itertools.product(range(nso + 1), repeat=2):
            for ndws in itertools.product(range(nso + 1), repeat=2):
                if sum(nups) != nup or sum(ndws) != ndw:
                    continue
                h = O.HOrbs(om, nups, ndws)
                d = h.dense()
                assert np.allclose(d, d.T, atol=1e-14)
                v = np.random.default_rng(1).standard_normal(h.dim)
                assert rel_err(h.matvec(v), d @ v) < 1e-13
                ws.append(np.linalg.eigvalsh(d))
        ws = np.sort(np.concatenate(ws))
        assert len(ws) == len(w_t) and np.max(np.abs(ws + shift - w_t)) < 1e-12


def test_oracle_threaded_baselines_equal_serial():
    """bench.py's multi-core CPU baseline (the reference's MPI row decomposition on OpenMP threads) computes
    the same product as the serial restatement."""
    om, _ = make_models("normal", "normal", 2, 3, seed=1)
    h = O.HNormal(om, 4, 4)
    v = np.random.default_rng(0).standard_normal(h.dim)
    hv = np.empty_like(v)
    O.normal_matvec_arrays_mt(h.dimup, h.dimdw, h.hd, h.up, h.dw, h.nd, v, hv, 3)
    assert rel_err(hv, h.matvec(v)) < 1e-14
    om, _ = make_models("superc", "normal", 2, 2, seed=1)
    f = O.HFlat(om, 0)
    rp, col, val = (np.ascontiguousarray(a) for a in f.csr)
    x = np.random.default_rng(1).standard_normal(f.dim) + 1j * np.random.default_rng(2).standard_normal(f.dim)
    y = np.empty_like(x)
    O.csr_matvec_z_mt(rp, col, val, x, y, 3)
    assert rel_err(y, f.matvec(x)) < 1e-14


def test_oracle_phonon_limits():
    """The phonon branches of the restatement have no reference fixture (PARITY UNPINNED against fixtures): exact
    limits instead -- Hermiticity, E = E_el + w0 n at g = A = 0, and the Lang-Firsov atomic limit."""
    om, _ = make_models("normal", "normal", 2, 2, seed=3)
    om.nph, om.w0_ph = 3, 0.7
    h = O.HNormal(om, 3, 2)
    d = h.dense()
    assert h.dim == 4 * h.dim_el and np.max(np.abs(d - d.T)) < 1e-14
    om0, _ = make_models("normal", "normal", 2, 2, seed=3)
    w0 = np.linalg.eigvalsh(O.HNormal(om0, 3, 2).dense())
    ref = np.sort(np.concatenate([w0 + 0.7 * n for n in range(4)]))
    assert np.max(np.abs(np.linalg.eigvalsh(d) - ref)) < 1e-12
    om.g_ph, om.a_ph = np.array([[0.3, 0.1], [0.1, 0.5]]), 0.2    # off-diagonal coupling + displacement field
    d2 = O.HNormal(om, 3, 2).dense()
    assert np.max(np.abs(d2 - d2.T)) < 1e-14
    om1, _ = make_models("normal", "normal", 1, 1, seed=1)
    om1.bv = np.zeros_like(om1.bv)
    om1.be = np.full_like(om1.be, 5.0)
    om1.hfmode, om1.uloc, om1.hloc = False, (0.0,), np.zeros((1, 1, 1, 1), complex)
    om1.nph, om1.w0_ph, om1.g_ph = 40, 1.0, np.array([[0.5]])
    assert abs(np.linalg.eigvalsh(O.HNormal(om1, 1, 1).dense())[0] + 1.0) < 1e-10


def test_oracle_flat_phonon_limits():
    """Phonon branches of the superc / nonsu2 restatement (orc_spmatvec_flat_ph; PARITY UNPINNED against
    fixtures, untested upstream): Hermiticity, E = E_el + w0 n at g = A = 0, and -- with the spin-flip terms
    switched off -- the nonsu2 N sector reproduces the union of the normal-mode (nup, ndw) sectors, phonons
    and density couplings included (two independent restatements of the same physics)."""
    for mode, sec in (("superc", 0), ("nonsu2", 4)):
        om, _ = make_models(mode, "hybrid", 2, 2, seed=5)
        om.nph, om.w0_ph = 2, 0.7
        h = O.HFlat(om, sec)
        d = h.dense()
        assert h.dim == 3 * h.dim_el and np.max(np.abs(d - d.conj().T)) < 1e-14
        om0, _ = make_models(mode, "hybrid", 2, 2, seed=5)
        w0 = np.linalg.eigvalsh(O.HFlat(om0, sec).dense())
        ref = np.sort(np.concatenate([w0 + 0.7 * n for n in range(3)]))
        assert np.max(np.abs(np.linalg.eigvalsh(d) - ref)) < 1e-12
        om.g_ph, om.a_ph = np.diag([0.3, 0.5]), 0.2
        d2 = h.__class__(om, sec).dense()
        assert np.max(np.abs(d2 - d2.conj().T)) < 1e-14 and np.max(np.abs(d2 - d)) > 0.1
        x = np.random.default_rng(4).standard_normal(h.dim) + 1j * np.random.default_rng(5).standard_normal(h.dim)
        assert rel_err(O.HFlat(om, sec).matvec(x), d2 @ x) < 1e-13
    # nonsu2 without spin mixing == union of normal sectors
    on, _ = make_models("normal", "normal", 2, 2, seed=6)
    on.nph, on.w0_ph, on.a_ph, on.g_ph = 2, 0.9, 0.1, np.diag([0.4, 0.2])
    o2, _ = make_models("nonsu2", "normal", 2, 2, seed=6)
    hl = np.zeros((2, 2, 2, 2), complex)
    hl[0, 0] = hl[1, 1] = on.hloc[0, 0]
    o2.hloc, o2.bu = hl, np.zeros_like(o2.bu)
    o2.be, o2.bv = np.stack([on.be[0]] * 2), np.stack([on.bv[0]] * 2)
    o2.nph, o2.w0_ph, o2.a_ph, o2.g_ph = on.nph, on.w0_ph, on.a_ph, on.g_ph
    ntot, ns = 5, 6
    w_flat = np.linalg.eigvalsh(O.HFlat(o2, ntot).dense())
    w_norm = np.sort(np.concatenate([np.linalg.eigvalsh(O.HNormal(on, nu, ntot - nu).dense())
                                     for nu in range(ntot + 1) if nu <= ns and ntot - nu <= ns]))
    assert w_flat.shape == w_norm.shape and np.max(np.abs(w_flat - w_norm)) < 1e-11


def test_oracle_cmplx_normal_matches_nonsu2():
    """Normal mode with complex algebra (-D_CMPLX_NORMAL; PARITY UNPINNED against fixtures): Hermiticity, and with
    a complex Hermitian impHloc the nonsu2 N sector without spin mixing (fixture-pinned restatement, complex
    natively) must reproduce the union of the complex normal-mode (nup, ndw) sectors -- this fixes the sign
    convention of the imaginary parts.  With a real impHloc the complex build equals the real one."""
    on, _ = make_models("normal", "normal", 2, 2, seed=7)
    hc = O.HNormalCmplx(on, 3, 2)
    assert np.max(np.abs(hc.A.dense())) == 0.0
    t = np.array([[0.0, 0.3 - 0.45j], [0.3 + 0.45j, 0.0]])
    on.hloc = np.asarray(on.hloc, complex).copy()
    on.hloc[0, 0] += t
    o2, _ = make_models("nonsu2", "normal", 2, 2, seed=7)
    hl = np.zeros((2, 2, 2, 2), complex)
    hl[0, 0] = hl[1, 1] = on.hloc[0, 0]
    o2.hloc, o2.bu = hl, np.zeros_like(o2.bu)
    o2.be, o2.bv = np.stack([on.be[0]] * 2), np.stack([on.bv[0]] * 2)
    ntot, ns = 5, 6
    w_flat = np.linalg.eigvalsh(O.HFlat(o2, ntot).dense())
    parts = []
    for nu in range(ntot + 1):
        if nu <= ns and ntot - nu <= ns:
            d = O.HNormalCmplx(on, nu, ntot - nu).dense()
            assert np.max(np.abs(d - d.conj().T)) < 1e-14
            parts.append(np.linalg.eigvalsh(d))
    w_norm = np.sort(np.concatenate(parts))
    assert w_flat.shape == w_norm.shape and np.max(np.abs(w_flat - w_norm)) < 1e-11
    # the imaginary part matters: the real-part-only build has a different spectrum
    w_real = np.sort(np.concatenate([np.linalg.eigvalsh(O.HNormal(on, nu, ntot - nu).dense())
                                     for nu in range(ntot + 1) if nu <= ns and ntot - nu <= ns]))
    assert np.max(np.abs(w_real - w_norm)) > 1e-3
    h = O.HNormalCmplx(on, 3, 2)
    x = np.random.default_rng(8).standard_normal(h.dim) + 1j * np.random.default_rng(9).standard_normal(h.dim)
    assert rel_err(h.matvec(x), h.dense() @ x) < 1e-13
    a, b, n = h.lanc_tridiag(x, 12)
    w = np.linalg.eigvalsh(np.diag(a) + np.diag(b[1:], 1) + np.diag(b[1:], -1))
    assert n == 12 and w[0] >= np.linalg.eigvalsh(h.dense())[0] - 1e-10


def test_oracle_cmplx_normal_replica_matches_nonsu2():
    """The same identity with complex replica bath matrices hbath_tmp(1,1,a,b,k) and a complex impHloc."""
    on, _ = make_models("normal", "replica", 2, 2, seed=9)
    o2, _ = make_models("nonsu2", "replica", 2, 2, seed=9)
    hb = np.asarray(on.hb, complex).copy()
    rng = np.random.default_rng(1)
    for k in range(hb.shape[-1]):
        x = rng.uniform(-0.3, 0.3)
        hb[0, 0, 0, 1, k] += 1j * x
        hb[0, 0, 1, 0, k] -= 1j * x
    on.hb = hb
    on.hloc = np.asarray(on.hloc, complex).copy()
    on.hloc[0, 0, 0, 1] += 0.2j
    on.hloc[0, 0, 1, 0] -= 0.2j
    hb2 = np.zeros_like(np.asarray(o2.hb, complex))
    hb2[0, 0] = hb2[1, 1] = hb[0, 0]
    hl = np.zeros((2, 2, 2, 2), complex)
    hl[0, 0] = hl[1, 1] = on.hloc[0, 0]
    o2.hb, o2.hloc, o2.vr = hb2, hl, on.vr.copy()
    ntot, ns = 5, 6
    w_flat = np.linalg.eigvalsh(O.HFlat(o2, ntot).dense())
    w_norm = np.sort(np.concatenate([np.linalg.eigvalsh(O.HNormalCmplx(on, nu, ntot - nu).dense())
                                     for nu in range(ntot + 1) if nu <= ns and ntot - nu <= ns]))
    assert w_flat.shape == w_norm.shape and np.max(np.abs(w_flat - w_norm)) < 1e-11


# --------------------------------------------------------------------------------------------
# the COMPLEX three-term recurrence pinned on fixtures: Sigma / Self moments of the superc
# directory and Sigma11 / Sigma12 moments of the nonsu2 directory with bath_type = normal can only be reproduced
# through tridiag_Hv_sector_superc / _nonsu2 + sp_lanc_tridiag on complex seeds (c, c^+ and the mixed
# c^+_up + c_dw, c^+_s + i c^+_s' channels)
# --------------------------------------------------------------------------------------------
def _flat_golden(name):
    inp, par = _from_dir(name)
    pm_par = {k: v for k, v in par.items() if k not in ("ed_hw_bath", "deltasc")}
    om, pm = golden_models(inp["ED_MODE"], inp["BATH_TYPE"], int(inp["NORB"]), int(inp["NBATH"]), pm_par)
    O.to_struct(om)   # fills the init_dmft_bath start bath
    pm.be, pm.bv, pm.bd, pm.bu = om.be, om.bv, om.bd, om.bu
    return inp, om, pm


def flat_momenta(name, tridiag):
    """(first, second) moment tables of a *_SUPERC / *_NONSU2 fixture directory through `tridiag`, with the fixture's
    own values: superc -> Sigma_momenta, Self_momenta; nonsu2 -> Sigma11_momenta, Sigma12_momenta."""
    from tests.gf_flat import momenta_nonsu2, momenta_superc
    inp, om, _ = _flat_golden(name)
    if inp["ED_MODE"] == "superc":
        a, b = momenta_superc(om, tridiag(om), beta=inp["BETA"], lmats=4096, ngfiter=int(inp["LANC_NGFITER"]))
        keys = ("Sigma_momenta", "Self_momenta")
    else:
        # LMATS=2000 in test/src/NORMAL_NONSU2/inputED.in
        a, b = momenta_nonsu2(om, tridiag(om), beta=inp["BETA"], lmats=2000, ngfiter=int(inp["LANC_NGFITER"]))
        keys = ("Sigma11_momenta", "Sigma12_momenta")
    ga, gb = (np.array(GOLD[name][k]).reshape(a.shape) for k in keys)
    return (a, ga), (b, gb)


@pytest.mark.parametrize("name,tol", [("NORMAL_SUPERC", 1e-8), ("NORMAL_NONSU2", 1e-11), ("HYBRID_NONSU2", 1e-11),
                                      ("HYBRID_SUPERC", 1e-7)])
def test_oracle_complex_tridiag_reproduces_flat_momenta(name, tol):
    """Measured agreement: nonsu2 3e-13; superc 4e-9 -- the superc fixture itself carries that much noise (its two
    equivalent orbitals differ in the 10th digit; the reference asserts these moments at 1e-8)."""
    def tridiag(om):
        def run(sec, v, nl):
            a, b, _ = O.HFlat(om, sec).lanc_tridiag(v.copy(), nl)
            return a, b
        return run

    for got, gold in flat_momenta(name, tridiag):
        assert np.max(np.abs(got / gold - 1.0)) < tol


@pytest.mark.parametrize("name", ["REPLICA_NONSU2", "GENERAL_NONSU2"])
def test_oracle_complex_tridiag_reproduces_replica_nonsu2_momenta(name):
    """Sigma_momenta.check of the replica / general NONSU2 directories: all sixteen Sigma_{ab}^{ss'} (ED_ALL_G = T), four
    moments each -- every diagonal and mixed channel of build_impG_nonsu2 through the complex recurrence, Delta from the
    replica matrices.  Measured agreement 1e-10."""
    from tests.gf_flat import momenta_nonsu2
    g = GOLD[name]
    om, _ = replica_golden_models(g["input"])
    O.to_struct(om)

    def run(sec, v, nl):
        a, b, _ = O.HFlat(om, sec).lanc_tridiag(v.copy(), nl)
        return a, b

    m = momenta_nonsu2(om, run, beta=g["input"]["BETA"], lmats=2000, ngfiter=int(g["input"]["LANC_NGFITER"]),
                       all_components=True)
    gold = np.array(g["Sigma_momenta"]).reshape(m.shape)
    assert np.max(np.abs(m / gold - 1.0)) < 1e-8


@pytest.mark.parametrize("name", ["REPLICA_SUPERC", "GENERAL_SUPERC"])
def test_oracle_complex_tridiag_reproduces_replica_superc_momenta(name):
    """Sigma_momenta.check (diagonal) and Self_momenta.check (every orbital pair) of the replica / general SUPERC
    directories: G_ab, barG, F_ab through the complex recurrence, the 4 x 4 Nambu-orbital inverse, Delta / Fdelta from the
    Nambu replica matrices.  Measured agreement 4e-13 / 8e-12."""
    from tests.gf_flat import momenta_superc
    g = GOLD[name]
    om, _ = replica_golden_models(g["input"])
    O.to_struct(om)

    def run(sec, v, nl):
        a, b, _ = O.HFlat(om, sec).lanc_tridiag(v.copy(), nl)
        return a, b

    sig, slf = momenta_superc(om, run, beta=g["input"]["BETA"], lmats=4096, ngfiter=int(g["input"]["LANC_NGFITER"]))
    assert np.max(np.abs(sig / np.array(g["Sigma_momenta"]).reshape(sig.shape) - 1.0)) < 1e-9
    assert np.max(np.abs(slf / np.array(g["Self_momenta"]).reshape(slf.shape) - 1.0)) < 1e-9


def test_oracle_jz_sectors_partition_the_ntot_sectors():
    """JZ_BASIS=T (ED_SECTOR.f90:289-350): the (Ntot, twoJz) maps are disjoint, ascending, and their union is the Ntot
    map; for a Jz-conserving model the sector Hamiltonians are the diagonal blocks of the Ntot Hamiltonian, so the union
    of their spectra is its spectrum (the Ntot builder is fixture-pinned)."""
    from tests.common import make_jz_models
    om, _ = make_jz_models(1, seed=2)
    L = O.lib()
    ns = 6
    for ntot in (3, 6):
        full = O.HFlat(om, ntot)
        maps, evs = [], []
        for twojz in range(-12, 13):
            n = L.orc_build_sector_nonsu2_jz(3, 1, ntot, twojz, None)
            if n <= 0:
                continue
            h = O.HFlat(om, ntot, twojz=twojz)
            assert h.dim == n and np.all(np.diff(h.map) > 0)
            maps.append(h.map)
            d = h.dense()
            assert np.abs(d - d.conj().T).max() < 1e-14
            # the block of the full matrix on these states
            idx = np.searchsorted(full.map, h.map)
            assert np.array_equal(full.map[idx], h.map)
            assert np.abs(full.dense()[np.ix_(idx, idx)] - d).max() < 1e-14
            evs.append(np.linalg.eigvalsh(d))
        allm = np.concatenate(maps)
        assert len(np.unique(allm)) == len(allm) == full.dim and np.array_equal(np.sort(allm), full.map)
        assert np.abs(np.sort(np.concatenate(evs)) - np.linalg.eigvalsh(full.dense())).max() < 1e-10
    # odd twoJz + even Ntot etc.: empty
    assert L.orc_build_sector_nonsu2_jz(3, 1, 6, 1, None) == 0


# ---------------------------------------------------------------------------------------------------------
# the remaining ground-state fixtures: doubles / energy / imp of all twelve directories, phisc of the *_SUPERC and magX
# of the NONSU2 ones (tests/observables.py).  They pin the eigenVECTORS (off-diagonal correlators <gs|X|gs> of every
# interaction family) and, for phisc / magX, apply_Cops between sectors on those vectors.
# ---------------------------------------------------------------------------------------------------------
def _golden_model(name):
    g = GOLD[name]
    if name in REPLICA_DIRS:
        om, _ = replica_golden_models(g["input"])
    else:
        inp, par = _from_dir(name)
        pm_par = {k: v for k, v in par.items() if k not in ("ed_hw_bath", "deltasc")}
        om, _ = golden_models(inp["ED_MODE"], inp["BATH_TYPE"], int(inp["NORB"]), int(inp["NBATH"]), pm_par)
    O.to_struct(om)   # fills the init_dmft_bath start bath
    return om, g


# *_SUPERC with a normal / hybrid bath: a second state 1.5e-6 .. 1e-5 above the ground state in the same sector, so any
# eigensolver's vector carries an admixture that shows at first order in observables (see the dens / docc test above)
_OBS_TOL = {"NORMAL_SUPERC": 5e-8, "HYBRID_SUPERC": 5e-8}


@pytest.mark.parametrize("name", DIRS + REPLICA_DIRS)
def test_oracle_reproduces_doubles_energy_imp(name):
    from tests import observables as ob
    om, g = _golden_model(name)
    e0, states = ob.ground_manifold(om)
    doubles, energy, imp = ob.doubles_energy_imp(om, states, e0)
    tol = _OBS_TOL.get(name, 1e-9)
    assert np.max(np.abs(doubles - np.array(g["doubles"]))) < tol
    assert np.max(np.abs(energy - np.array(g["energy"]))) < tol
    assert np.max(np.abs(imp - np.array(g["imp"]))) < tol


@pytest.mark.parametrize("name", ["REPLICA_NORMAL", "GENERAL_NORMAL"])
def test_oracle_reproduces_exciton(name):
    """exciton.check of the replica / general NORMAL directories: exct_S0 and exct_Tz between the two orbitals through
    apply_Cops in normal mode (c_1s + c_2s on the ground state, both spin species)."""
    from tests import observables as ob
    om, g = _golden_model(name)
    e0, states = ob.ground_manifold(om)
    assert np.max(np.abs(ob.exciton_normal(om, states) - np.array(g["exciton"]))) < 1e-9


@pytest.mark.parametrize("name", ["REPLICA_NONSU2", "GENERAL_NONSU2"])
def test_oracle_reproduces_exciton_nonsu2(name):
    """exciton.check of the replica / general NONSU2 directories: S0, Tx, Ty, Tz between the two orbitals, six apply_Cops
    combinations (same spin, opposite spin, with a factor -i) on the ground state."""
    from tests import observables as ob
    from tests.gf_flat import apply_cops
    om, g = _golden_model(name)
    e0, states = ob.ground_manifold(om)
    cache = {}

    def hsector(sec):
        if sec not in cache:
            cache[sec] = O.HFlat(om, sec)
        return cache[sec]

    got = ob.exciton_nonsu2(om, states, lambda h1, h2, v, ops: apply_cops(h1, h2, v, ops, om.ns), hsector)
    assert np.max(np.abs(got - np.array(g["exciton"]))) < 1e-9


@pytest.mark.parametrize("name", [d for d in DIRS + REPLICA_DIRS if "phisc" in GOLD[d] or "magX" in GOLD[d]])
def test_oracle_reproduces_phisc_magx(name):
    """apply_Cops on the ground state into the neighbouring sector (ED_SECTOR.f90:839-960), as the reference's
    observables do for the superconducting order parameter and the in-plane magnetisation."""
    from tests import observables as ob
    from tests.gf_flat import apply_cops
    om, g = _golden_model(name)
    e0, states = ob.ground_manifold(om)
    cache = {}

    def hsector(sec):
        if sec not in cache:
            cache[sec] = O.HFlat(om, sec)
        return cache[sec]

    def cops(h1, h2, v, ops):
        return apply_cops(h1, h2, v, ops, om.ns)

    tol = _OBS_TOL.get(name, 1e-9)
    if "phisc" in g:
        phi = ob.phisc(om, states, cops, hsector)
        assert np.max(np.abs(phi.real - np.array(g["phisc"]))) < tol and np.max(np.abs(phi.imag)) < tol
    if "magX" in g:
        assert np.max(np.abs(ob.magx(om, states, cops, hsector) - np.array(g["magX"]))) < tol
