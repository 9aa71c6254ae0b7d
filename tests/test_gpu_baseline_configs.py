"""BASELINE.json configs at their literal sizes, each with a correctness assert ("configs_untested").

cfg3 (3 orbitals, hybrid, Nbath=8, sector (5,6), Dim 213 444) and cfg4 (superc, 2 orbitals, hybrid, Nbath=8,
Sz=0, Dim 184 756) are small enough for the CPU oracle: H*v and a 30-step tridiagonalisation are compared
with it directly.  cfg5 (nonsu2, Dim 10 400 600) and cfg2 (Dim 11 778 624) are beyond the oracle's reach in
seconds and are checked through size-independent properties: two independent device evaluations of the same
operator (on-the-fly vs stored image), Hermiticity, linearity, and the default one-reduction recurrence
(beta^2 from the sweep's partials) against the literal one (beta = |w - alpha v|) on the continued fraction the
Green's functions are made of.

Tolerances: 1e-12 relative for H*v, 1e-10 for alpha / beta and for the continued fraction (north_star)."""
import numpy as np
import pytest

from tests.common import rel_err

pytestmark = pytest.mark.gpu


def _oracle_model(pm):
    """oracle.Model with exactly the parameters of an edipack_amd ImpurityModel (normal / hybrid baths)."""
    from oracle import oracle as O
    def tab(x):
        return float(x) if np.isscalar(x) else x
    return O.Model(ed_mode=pm.ed_mode, bath_type=pm.bath_type, norb=pm.norb, nbath=pm.nbath, nspin=pm.nspin,
                   hfmode=pm.hfmode, xmu=pm.xmu, uloc=tuple(np.asarray(pm.uloc, float)), ust=tab(pm.ust), jh=tab(pm.jh),
                   jx=tab(pm.jx), jp=tab(pm.jp), hloc=pm.hloc, be=pm.be, bv=pm.bv, bd=pm.bd, bu=pm.bu)


def _cf(alpha, beta, z):
    """continued fraction <v|(z-H)^-1|v> from the tridiagonal (what add_to_lanczos_gf_* consumes)."""
    g = 0.0
    for k in range(len(alpha) - 1, -1, -1):
        b2 = beta[k + 1] ** 2 if k + 1 < len(alpha) else 0.0
        g = 1.0 / (z - alpha[k] - b2 * g)
    return g


def test_cfg3_literal_matches_oracle(gpu):
    from oracle import oracle as O
    from edipack_amd.synthetic import WORKLOADS, build_workload, synthetic_model
    w = WORKLOADS["cfg3"]
    hg = build_workload(w)
    ho = O.HNormal(_oracle_model(synthetic_model(w)), *w.sector)
    assert hg.dim == ho.dim == 213444
    rng = np.random.default_rng(12345)
    v = rng.standard_normal(ho.dim)
    assert rel_err(hg.apply(v), ho.matvec(v)) < 1e-12
    a, b, nd = hg.lanczos_tridiag(v, 30)
    a_ref, b_ref, nd_ref = ho.lanc_tridiag(v, 30)
    assert nd == nd_ref == 30
    assert rel_err(a, a_ref) < 1e-10 and rel_err(b, b_ref) < 1e-10
    hg.destroy()


def test_cfg4_literal_matches_oracle(gpu):
    from oracle import oracle as O
    from edipack_amd.hamiltonian import SectorHamiltonian
    from edipack_amd.synthetic import WORKLOADS, build_workload, synthetic_model
    w = WORKLOADS["cfg4"]
    pm = synthetic_model(w)
    hg = build_workload(w)
    hdir = SectorHamiltonian.direct_from_model(pm, w.sector)
    ho = O.HFlat(_oracle_model(pm), w.sector)
    assert hg.dim == hdir.dim == ho.dim == 184756
    rng = np.random.default_rng(12345)
    v = rng.standard_normal(ho.dim) + 1j * rng.standard_normal(ho.dim)
    ref = ho.matvec(v)
    assert rel_err(hg.apply(v), ref) < 1e-12
    assert rel_err(hdir.apply(v), ref) < 1e-12          # ED_SPARSE_H=F must give the same product
    a, b, nd = hg.lanczos_tridiag(v, 30)
    a_ref, b_ref, nd_ref = ho.lanc_tridiag(v, 30)
    assert nd == nd_ref == 30
    assert rel_err(a, a_ref) < 1e-10 and rel_err(b, b_ref) < 1e-10
    hg.destroy()
    hdir.destroy()


def _dev_apply(h, x):
    import torch
    y = torch.empty_like(x)
    h.apply_dev(x.data_ptr(), y.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return y


def test_cfg5_fullsize_direct_vs_stored(gpu):
    """config 5 (nonsu2, 3 orbitals, Nbath=10 hybrid, N=13: Dim = 10 400 600, complex): the on-the-fly kernel (no
    matrix) against the device-built stored image (2.7 GB), Hermiticity, linearity, and 25 Lanczos steps of both."""
    import torch
    from edipack_amd.hamiltonian import SectorHamiltonian
    from edipack_amd.synthetic import WORKLOADS, synthetic_model
    w = WORKLOADS["cfg5"]
    m = synthetic_model(w)
    hd = SectorHamiltonian.direct_from_model(m, w.sector)
    hs = SectorHamiltonian.flat_from_model(m, w.sector)
    assert hd.dim == hs.dim == 10400600
    g = torch.Generator(device="cuda").manual_seed(5)

    def rnd():
        return torch.complex(torch.randn(hd.dim, dtype=torch.float64, device="cuda", generator=g),
                             torch.randn(hd.dim, dtype=torch.float64, device="cuda", generator=g))
    u, v = rnd(), rnd()
    hv_d, hv_s = _dev_apply(hd, v), _dev_apply(hs, v)
    scale = float(torch.linalg.norm(hv_s))
    assert float(torch.linalg.norm(hv_d - hv_s)) < 1e-13 * scale
    hu = _dev_apply(hd, u)
    assert abs(complex(torch.vdot(u, hv_d) - torch.vdot(hu, v))) < 1e-11 * scale * float(torch.linalg.norm(u))
    lin = _dev_apply(hd, 2.0 * u - 3.0 * v) - (2.0 * hu - 3.0 * hv_d)
    assert float(torch.linalg.norm(lin)) < 1e-13 * scale
    v0 = v.cpu().numpy()
    a1, b1, _ = hd.lanczos_tridiag(v0, 25)
    a2, b2, _ = hs.lanczos_tridiag(v0, 25)
    assert rel_err(a1, a2) < 1e-10 and rel_err(b1, b2) < 1e-10
    assert abs(a1[0] - float((torch.vdot(v, hv_d) / torch.vdot(v, v)).real)) < 1e-10 * abs(a1[0])
    hd.destroy()
    hs.destroy()


@pytest.mark.parametrize("workload", ["cfg2", "cfg3", "cfg4"])
def test_default_recurrence_vs_exact_beta(gpu, monkeypatch, workload):
    """The default fused step takes beta^2 from the sweep's partials (sum (w - sg v)^2 - (alpha - sg)^2); the
    reference recurrence takes beta = |w - alpha v|.  200 steps (lanc_ngfiter) of both on the BASELINE sectors: the
    coefficients agree and so does the continued fraction the Green's functions are built from, at 1e-10."""
    from edipack_amd.synthetic import WORKLOADS, build_workload
    h = build_workload(WORKLOADS[workload])
    rng = np.random.default_rng(77)
    v = rng.standard_normal(h.dim) if not h.is_complex else rng.standard_normal(h.dim) + 1j * rng.standard_normal(h.dim)
    n = 200
    a1, b1, n1 = h.lanczos_tridiag(v, n)
    monkeypatch.setenv("EDIGPU_LANCZOS_EXACTBETA", "1")
    a2, b2, n2 = h.lanczos_tridiag(v, n)
    monkeypatch.delenv("EDIGPU_LANCZOS_EXACTBETA")
    assert n1 == n2 == n
    # the first ~60 coefficients are determined to rounding; later ones amplify rounding differences (Lanczos loses
    # orthogonality) while the continued fraction stays determined
    assert rel_err(a1[:40], a2[:40]) < 1e-10 and rel_err(b1[:40], b2[:40]) < 1e-10
    # z where the Green's functions are evaluated: outside the band (z = E0 + i w_n with E0 under the excited
    # sector's spectrum: ED_GF_NORMAL.f90:410-426).  Inside the band a 200-step fraction has not converged in n
    # (error ~ exp(-n Im z / bandwidth)) and amplifies rounding differences: not a determined quantity
    tm = np.diag(a2) + np.diag(b2[1:], 1) + np.diag(b2[1:], -1)
    th = np.linalg.eigvalsh(tm)
    lo, hi = th[0], th[-1]
    for z in (lo - 0.05, lo - 0.5 + 0.1j, lo - 0.02 + 0.003j, hi + 0.3, hi + 1.0 + 0.2j):
        g1, g2 = _cf(a1, b1, z), _cf(a2, b2, z)
        assert abs(g1 - g2) < 1e-10 * abs(g2), (workload, z, g1, g2)
    h.destroy()


def test_tridiag_breakdown_on_exact_eigenvector(gpu):
    """a seed that is an exact eigenvector (diagonal H) gives beta = 0 on the first step; with any
    threshold (the default, and an explicit zero) the recurrence must stop there with finite coefficients."""
    from edipack_amd.hamiltonian import SectorHamiltonian
    n = 64
    rowptr = np.arange(n + 1, dtype=np.int64)
    col = np.arange(n, dtype=np.int32)
    val = np.linspace(-3.0, 3.0, n)
    h = SectorHamiltonian.csr_from_arrays(rowptr, col, val)
    v = np.zeros(n)
    v[7] = 2.0
    for thr in (1e-12, 0.0):
        a, b, nd = h.lanczos_tridiag(v, 10, threshold=thr)
        assert nd == 1
        assert np.all(np.isfinite(a)) and np.all(np.isfinite(b))
        assert abs(a[0] - val[7]) < 1e-14 and np.all(a[1:] == 0.0) and np.all(b == 0.0)
    h.destroy()
    # the same on a normal-mode sector through the fused step: the atomic limit (no hybridisation) is diagonal
    from tests.common import make_models
    om, pm = make_models("normal", "normal", 1, 3, seed=2)
    pm.bv = np.zeros_like(pm.bv)
    hn = SectorHamiltonian.normal_from_model(pm, 2, 2)
    v = np.zeros(hn.dim)
    v[5] = 1.0
    for thr in (1e-12, 0.0):
        a, b, nd = hn.lanczos_tridiag(v, 8, threshold=thr)
        assert nd == 1 and np.all(np.isfinite(a)) and np.all(np.isfinite(b)) and np.all(b == 0.0)
    hn.destroy()


def test_shifted_spectrum_keeps_the_fused_step_accurate(gpu):
    """a spectrum far from zero (|alpha| >> beta on every step) made beta^2 = <w|w> - alpha^2 cancel
    and sent every step through the single-workgroup exact pass.  The partials are now accumulated about the
    previous alpha; the coefficients must match the oracle's literal recurrence with a large chemical-potential
    shift."""
    from oracle import oracle as O
    from edipack_amd.hamiltonian import SectorHamiltonian
    from tests.common import make_models
    om, pm = make_models("normal", "normal", 2, 3, seed=11)
    # every level (impurity and bath) raised by E0: H -> H + E0 (Nup + Ndw), a constant 8 E0 in this sector
    e0 = 100.0
    hl = np.array(om.hloc, complex)
    for a in range(2):
        hl[0, 0, a, a] += e0
    for m in (om, pm):
        m.hloc, m.be = hl, np.asarray(m.be) + e0
    ho = O.HNormal(om, 4, 4)
    hg = SectorHamiltonian.normal_from_model(pm, 4, 4)
    v = np.random.default_rng(3).standard_normal(ho.dim)
    a, b, nd = hg.lanczos_tridiag(v, 40)
    a_ref, b_ref, _ = ho.lanc_tridiag(v, 40)
    assert abs(a_ref).min() > 100 * abs(b_ref[1:]).max()      # the regime the advice describes
    # (the |v| = 1 shortcut beta^2 = qq - (alpha - sg)^2 drifted from 1e-12 to 1e-2 in twelve steps here: its error is
    # fed back through 2 (alpha - sg) sg / beta^2 ~ 90 per step; the three-sum form is an identity)
    assert rel_err(a, a_ref) < 1e-10 and np.max(np.abs(b - b_ref)) < 1e-10 * np.max(np.abs(a_ref))
    th = np.linalg.eigvalsh(np.diag(a_ref) + np.diag(b_ref[1:], 1) + np.diag(b_ref[1:], -1))
    for z in (th[0] - 0.5, th[0] - 0.05 + 0.01j, th[-1] + 0.3):
        g, g_ref = _cf(a, b, z), _cf(a_ref, b_ref, z)
        assert abs(g - g_ref) < 1e-10 * abs(g_ref), (z, g, g_ref)
    hg.destroy()


# --------------------------------------------------------------------------------------------
# the large-sector forms of the down-term sweep (two columns per lane; LDS-staged row chunks) forced onto small
# sectors the oracle reaches: whole sectors, odd DimUp, down-row shards in the two-phase form, explicit image
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tile,rows", [("0", "72"), ("1", "8"), ("1", "20"), ("1", "72"), ("1", "152")])
@pytest.mark.parametrize("bath,norb,nbath,sec", [
    ("normal", 2, 4, (5, 5)),     # DimUp = DimDw = 252, Hnd terms
    ("hybrid", 3, 5, (4, 4)),     # 3 orbitals: 70 x 70, several Hnd terms per row, one panel narrower than a wave
    ("hybrid", 3, 4, (3, 4)),     # odd DimUp = 35 (8-byte aligned rows, the EDGE path)
    ("normal", 1, 6, (3, 4)),     # no Hnd at all
])
def test_down_sweep_variants_match_oracle(gpu, monkeypatch, tile, rows, bath, norb, nbath, sec):
    import torch
    from oracle import oracle as O
    from edipack_amd.hamiltonian import SectorHamiltonian
    from tests.common import make_models
    monkeypatch.setenv("EDIGPU_PANEL_VEC2_MIN", "1")
    monkeypatch.setenv("EDIGPU_PANEL_TILE", tile)
    monkeypatch.setenv("EDIGPU_TILE_ROWS", rows)
    om, pm = make_models("normal", bath, norb, nbath, seed=21)
    ho = O.HNormal(om, *sec)
    hg = SectorHamiltonian.normal_from_model(pm, *sec)
    rng = np.random.default_rng(4)
    v = rng.standard_normal(ho.dim)
    ref = ho.matvec(v)
    assert rel_err(hg.apply(v), ref) < 1e-12
    a, b, _ = hg.lanczos_tridiag(v, 25)             # the fused step: the three sums in the sweep's epilogue
    a_ref, b_ref, _ = ho.lanc_tridiag(v, 25)
    assert rel_err(a, a_ref) < 1e-10 and rel_err(b, b_ref) < 1e-10
    hg.destroy()
    # explicit (hand-over) image of the same sector
    he = SectorHamiltonian.normal_from_arrays(ho.dimup, ho.dimdw, ho.hd, ho.up, ho.dw, ho.nd if ho.has_nd else None)
    assert rel_err(he.apply(v), ref) < 1e-12
    he.destroy()
    # three down-row shards, two-phase form (the sweep runs on local rows with global partner rows)
    vd = torch.from_numpy(v).cuda()
    cuts = [0, ho.dimdw // 3, ho.dimdw // 3 + 1, ho.dimdw]
    out = []
    for first, last in zip(cuts[:-1], cuts[1:]):
        hs = SectorHamiltonian.normal_from_model(pm, *sec, dw_first=first, dw_count=last - first)
        hv = torch.empty(hs.nloc, dtype=torch.float64, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        hs.apply_local_dev(vd[hs.row_first:].data_ptr(), hv.data_ptr(), st)
        hs.apply_remote_dev(vd.data_ptr(), hv.data_ptr(), st)
        torch.cuda.synchronize()
        out.append(hv.cpu().numpy())
        hs.destroy()
    assert rel_err(np.concatenate(out), ref) < 1e-12


@pytest.mark.parametrize("name,tol", [("NORMAL_SUPERC", 1e-8), ("NORMAL_NONSU2", 1e-10), ("HYBRID_NONSU2", 1e-9),
                                      ("HYBRID_SUPERC", 2e-7)])
@pytest.mark.parametrize("form", ["stored", "direct"])
def test_golden_flat_momenta_through_gpu_tridiag(gpu, name, tol, form):
    """Sigma / Self moments (superc) and Sigma11 / Sigma12 moments (nonsu2: normal bath, and the hybrid bath with every
    G_{ab}^{ss'} channel) of the reference's fixtures with every tridiagonalisation done by edigpu_lanczos_tridiag on
    GPU-built sectors: pins the complex device recurrence (stored SELL image and on-the-fly kernel) on the reference's
    own data."""
    from edipack_amd.hamiltonian import SectorHamiltonian
    from tests.test_oracle_golden import _flat_golden, flat_momenta
    _, _, pm = _flat_golden(name)
    build = SectorHamiltonian.flat_from_model if form == "stored" else SectorHamiltonian.direct_from_model
    cache = {}

    def tridiag(om):
        def run(sec, v, nl):
            if sec not in cache:
                cache[sec] = build(pm, sec)
            a, b, _ = cache[sec].lanczos_tridiag(v, nl)
            return a, b
        return run

    for got, gold in flat_momenta(name, tridiag):
        assert np.max(np.abs(got / gold - 1.0)) < tol
    for h in cache.values():
        h.destroy()


def test_cfg2_handover_image_runs_the_factored_kernels(gpu, monkeypatch):
    """edigpu_normal_create on the explicit arrays of config 2 (what INTEGRATION.md section 2 patches in) recovers the
    factored tables: same H*v as the library-built sector to rounding, the same tridiagonal, and the arrays come back
    from edigpu_normal_export bit for bit."""
    from edipack_amd.hamiltonian import SectorHamiltonian
    from edipack_amd.synthetic import WORKLOADS, synthetic_model
    for k in ("EDIGPU_HANDOVER_FACTOR", "EDIGPU_NORMAL_EXPLICIT", "EDIGPU_ND_NO_MERGE"):
        monkeypatch.delenv(k, raising=False)
    w = WORKLOADS["cfg2"]
    pm = synthetic_model(w)
    hl = SectorHamiltonian.normal_from_model(pm, *w.sector)
    hd, up, dw, nd = hl.export_normal()
    hh = SectorHamiltonian.normal_from_arrays(hl.dim_up, hl.dim_dw, hd, up, dw, nd)
    fac, nterms, ncls, panel = hh.image_info()[:4]
    assert fac == 1 and 1 <= nterms <= hl.image_info()[1] and ncls == 4 and panel == hl.image_info()[3]
    v = np.random.default_rng(8).standard_normal(hl.dim)
    v /= np.linalg.norm(v)
    a, b = hl.apply(v), hh.apply(v)
    assert rel_err(b, a) < 1e-14
    al, bl, _ = hl.lanczos_tridiag(v, 60)
    ah, bh, _ = hh.lanczos_tridiag(v, 60)
    assert rel_err(ah[:30], al[:30]) < 1e-10 and rel_err(bh[:30], bl[:30]) < 1e-10
    for z in (40.0 + 0.1j, -40.0 + 0.1j, 25.0j):
        assert abs(_cf(ah, bh, z) - _cf(al, bl, z)) / abs(_cf(al, bl, z)) < 1e-10
    hd2, _, _, nd2 = hh.export_normal()
    assert np.array_equal(hd2, hd) and all(np.array_equal(x, y) for x, y in zip(nd2, nd))
    hl.destroy(), hh.destroy()


@pytest.mark.parametrize("workload", ["cfg3_ns15", "cfg3_ns16"])
def test_hbm_resident_ladder_panel_major_vs_natural(gpu, monkeypatch, workload):
    """The sectors the roofline fraction is quoted on (41 M and 166 M rows, HBM-resident): the default loop -- the
    impurity-block image on its padded 16-column panels (csrc/kernels_ib.hip) -- against the same recurrence on the
    generic kernels, panel-major (EDIGPU_IB=0) and on the reference's layout (EDIGPU_BLOCKED=0): three different pairs
    of kernels and two layout conversions, same coefficients; and the product's linearity and symmetry at that size
    (host vectors: the impurity-block product; its first Lanczos coefficient against <v|H|v>)."""
    from edipack_amd.hamiltonian import SectorHamiltonian
    from edipack_amd.synthetic import WORKLOADS, synthetic_model
    for k in ("EDIGPU_BLOCKED", "EDIGPU_BLOCKED_W", "EDIGPU_BLOCKED_MIN", "EDIGPU_IB", "EDIGPU_IB_MIN", "EDIGPU_IB_ROWS"):
        monkeypatch.delenv(k, raising=False)
    w = WORKLOADS[workload]
    pm = synthetic_model(w)
    hb = SectorHamiltonian.normal_from_model(pm, *w.sector)
    assert hb.image_info()[4] == 16 and hb.image_info()[5] in (1, 3, 5)
    rng = np.random.default_rng(11)
    v = rng.standard_normal(hb.dim)
    v /= np.linalg.norm(v)
    ab, bb, nb = hb.lanczos_tridiag(v, 40)
    x = rng.standard_normal(hb.dim)
    hv, hx = hb.apply(v), hb.apply(x)
    assert rel_err(hb.apply(2.0 * v - 3.0 * x), 2.0 * hv - 3.0 * hx) < 1e-12
    assert abs(np.dot(x, hv) - np.dot(hx, v)) < 1e-10 * np.linalg.norm(hv) * np.linalg.norm(x)     # symmetric H
    assert abs(np.dot(v, hv) - ab[0]) < 1e-10 * max(1.0, abs(ab[0]))                                # alpha_1 = <v|H|v>
    hb.destroy()
    del hx, x
    monkeypatch.setenv("EDIGPU_IB", "0")
    hp = SectorHamiltonian.normal_from_model(pm, *w.sector)
    assert hp.image_info()[4] == 128 and hp.image_info()[5] == 0
    ap, bp, npm = hp.lanczos_tridiag(v, 40)
    assert rel_err(hp.apply(v), hv) < 1e-12          # the generic kernels' product against the impurity-block one
    hp.destroy()
    del hv
    monkeypatch.setenv("EDIGPU_BLOCKED", "0")
    hn = SectorHamiltonian.normal_from_model(pm, *w.sector)
    assert hn.image_info()[4] == 0
    an, bn, nn = hn.lanczos_tridiag(v, 40)
    hn.destroy()
    assert nb == nn == npm == 40
    assert rel_err(ab[:25], an[:25]) < 1e-10 and rel_err(bb[:25], bn[:25]) < 1e-10
    assert rel_err(ap[:25], an[:25]) < 1e-10 and rel_err(bp[:25], bn[:25]) < 1e-10
    for z in (60.0 + 0.1j, -60.0 + 0.1j, 40.0j):
        assert abs(_cf(ab, bb, z) - _cf(an, bn, z)) / abs(_cf(an, bn, z)) < 1e-10
        assert abs(_cf(ap, bp, z) - _cf(an, bn, z)) / abs(_cf(an, bn, z)) < 1e-10


def test_ns17_rows_staged_in_halves_vs_generic_kernels(gpu, monkeypatch):
    """Ns = 17 (591 M rows, rows of 194 KB: longer than the LDS): the impurity-block kernels with the row staged in two
    halves (image_info[5] == 2) against the generic split-row kernels on the reference's layout -- the product on the same
    vector, its linearity and symmetry at that size, alpha_1 = <v|H|v>, and the first coefficients of the recurrence
    (fused step of the halves form against the unfused natural-layout loop)."""
    from edipack_amd.hamiltonian import SectorHamiltonian
    from edipack_amd.synthetic import WORKLOADS, synthetic_model
    for k in ("EDIGPU_BLOCKED", "EDIGPU_BLOCKED_W", "EDIGPU_BLOCKED_MIN", "EDIGPU_IB", "EDIGPU_IB_MIN", "EDIGPU_IB_ROWS",
              "EDIGPU_IB_SPLIT", "EDIGPU_IB_COLS2"):
        monkeypatch.delenv(k, raising=False)
    w = WORKLOADS["cfg3_ns17"]
    pm = synthetic_model(w)
    hb = SectorHamiltonian.normal_from_model(pm, *w.sector)
    assert hb.image_info()[4] == 16 and hb.image_info()[5] in (2, 4)   # 4: the local-block rows kernel on the halves
    rng = np.random.default_rng(17)
    v = rng.standard_normal(hb.dim)
    v /= np.linalg.norm(v)
    x = rng.standard_normal(hb.dim)
    ab, bb, nb = hb.lanczos_tridiag(v, 12)
    hv, hx = hb.apply(v), hb.apply(x)
    y = 2.0 * v
    y -= 3.0 * x
    hy = hb.apply(y)
    del y
    hy -= 2.0 * hv
    hy += 3.0 * hx
    assert np.linalg.norm(hy) < 1e-12 * np.linalg.norm(hx)                                          # linear
    del hy
    assert abs(np.dot(x, hv) - np.dot(hx, v)) < 1e-10 * np.linalg.norm(hv) * np.linalg.norm(x)     # symmetric H
    assert abs(np.dot(v, hv) - ab[0]) < 1e-10 * max(1.0, abs(ab[0]))                                # alpha_1 = <v|H|v>
    hb.destroy()
    del hx, x
    monkeypatch.setenv("EDIGPU_IB", "0")
    hn = SectorHamiltonian.normal_from_model(pm, *w.sector)
    assert hn.image_info()[5] == 0
    g = hn.apply(v)
    g -= hv
    assert np.linalg.norm(g) < 1e-12 * np.linalg.norm(hv)
    del g, hv
    an, bn, nn = hn.lanczos_tridiag(v, 12)
    hn.destroy()
    assert nb == nn == 12
    assert rel_err(ab, an) < 1e-10 and rel_err(bb, bn) < 1e-10


@pytest.mark.parametrize("name", ["REPLICA_SUPERC", "GENERAL_SUPERC", "REPLICA_NONSU2", "GENERAL_NONSU2"])
def test_golden_replica_flat_momenta_through_gpu_tridiag(gpu, name):
    """The moment files of the replica / general SUPERC and NONSU2 directories (every orbital pair of Self, all sixteen
    Sigma_{ab}^{ss'}) with every tridiagonalisation done by edigpu_lanczos_tridiag on GPU-built sectors."""
    from edipack_amd.hamiltonian import SectorHamiltonian
    from oracle import oracle as O
    from tests.common import replica_golden_models
    from tests.gf_flat import momenta_nonsu2, momenta_superc
    from tests.test_oracle_golden import GOLD
    g = GOLD[name]
    om, pm = replica_golden_models(g["input"])
    O.to_struct(om)
    cache = {}

    def run(sec, v, nl):
        if sec not in cache:
            cache[sec] = SectorHamiltonian.flat_from_model(pm, sec)
        a, b, _ = cache[sec].lanczos_tridiag(v, nl)
        return a, b

    kw = dict(beta=g["input"]["BETA"], ngfiter=int(g["input"]["LANC_NGFITER"]))
    if name.endswith("SUPERC"):
        sig, slf = momenta_superc(om, run, lmats=4096, **kw)
        assert np.max(np.abs(sig / np.array(g["Sigma_momenta"]).reshape(sig.shape) - 1.0)) < 1e-8
        assert np.max(np.abs(slf / np.array(g["Self_momenta"]).reshape(slf.shape) - 1.0)) < 1e-8
    else:
        m = momenta_nonsu2(om, run, lmats=2000, all_components=True, **kw)
        assert np.max(np.abs(m / np.array(g["Sigma_momenta"]).reshape(m.shape) - 1.0)) < 1e-8
    for h in cache.values():
        h.destroy()
