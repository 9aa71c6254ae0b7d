"""Shared input generators for the tests: the SAME physical parameters are handed to the CPU
oracle (oracle.Model) and to the product (edipack_amd.ImpurityModel)."""
from __future__ import annotations

import numpy as np

from edipack_amd.hamiltonian import ImpurityModel
from oracle import oracle as O


def make_models(ed_mode: str, bath_type: str, norb: int, nbath: int, seed: int = 0, jxp: float = 0.25,
                reference_bath: bool = False, **over):
    """Structure-preserving synthetic impurity problem: seeded bath levels
    e~U(-2,2), hybridisations v~U(0.1,0.6), Uloc=2, Ust=1.5, Jh=Jx=Jp=0.25, pair amplitudes
    d~U(-0.1,0.1) (superc), spin-flip u~U(0,0.3) + Hermitian impHloc (nonsu2), xmu=0, hfmode=T."""
    rng = np.random.default_rng(20260630 + seed)
    nspin = 2 if ed_mode == "nonsu2" else 1
    nfoo = 1 if bath_type == "hybrid" else norb
    be = rng.uniform(-2, 2, (nspin, nfoo, nbath))
    bv = rng.uniform(0.1, 0.6, (nspin, norb, nbath))
    bd = rng.uniform(-0.1, 0.1, (nspin, nfoo, nbath)) if ed_mode == "superc" else None
    bu = rng.uniform(0.0, 0.3, (nspin, norb, nbath)) if ed_mode == "nonsu2" else None
    if nspin == 2:
        # keep the two spin species of e and v equal (SU(2)-symmetric bath), break it only via u
        be[1] = be[0]
        bv[1] = bv[0]
    hl = np.zeros((nspin, nspin, norb, norb), complex)
    if ed_mode == "nonsu2":
        a = rng.standard_normal((2 * norb, 2 * norb)) + 1j * rng.standard_normal((2 * norb, 2 * norb))
        a = 0.2 * (a + a.conj().T)
        for s in range(2):
            for t in range(2):
                hl[s, t] = a[s * norb:(s + 1) * norb, t * norb:(t + 1) * norb]
    else:
        a = rng.standard_normal((norb, norb))
        a = 0.3 * (a + a.T)
        hl[0, 0] = a
    if bath_type in ("replica", "general"):
        return _make_replica_models(rng, ed_mode, bath_type, norb, nbath, nspin, hl, jxp, over)
    par = dict(ed_mode=ed_mode, bath_type=bath_type, norb=norb, nbath=nbath, nspin=nspin, hfmode=True, xmu=0.0,
               uloc=tuple([2.0] * norb), ust=1.5 if norb > 1 else 0.0, jh=0.25 if norb > 1 else 0.0,
               jx=jxp if norb > 1 else 0.0, jp=jxp if norb > 1 else 0.0)
    par.update(over)
    if reference_bath:
        om = O.Model(hloc=hl, **par)
        O.init_dmft_bath(om)
        be, bv, bd, bu = om.be, om.bv, om.bd, om.bu
    else:
        om = O.Model(hloc=hl, be=be, bv=bv, bd=bd, bu=bu, **par)
    pm = ImpurityModel(hloc=hl, be=be, bv=bv, bd=bd, bu=bu,
                       **{k: (np.asarray(v) if k == "uloc" else v) for k, v in par.items()})
    return om, pm


def _replica_pair(ed_mode, bath_type, norb, nbath, nspin, hl, hb, v, par):
    """oracle.Model + ImpurityModel of one replica/general problem: hb[is,js,a,b,k], v[nspin,norb,nbath]
    (replica: one value per k)."""
    om = O.Model(hloc=hl, hb=hb, vr=v[0, 0, :].copy(), vg=v.reshape(nspin * norb, nbath).copy(), **par)
    pm = ImpurityModel(hloc=hl, hb=hb, bv=v.copy(),
                       **{k: (np.asarray(x) if k == "uloc" else x) for k, x in par.items()})
    return om, pm


def _make_replica_models(rng, ed_mode, bath_type, norb, nbath, nspin, hl, jxp, over):
    """Seeded replica/general bath: Hermitian per-replica matrices with inter-orbital (and, nonsu2,
    spin-flip; superc, anomalous) blocks -- the structure build_Hreplica produces from a symmetric basis."""
    n1 = 2 if ed_mode in ("superc", "nonsu2") else 1
    hb = np.zeros((n1, n1, norb, norb, nbath), complex)
    for k in range(nbath):
        a = rng.uniform(-0.4, 0.4, (norb, norb))
        a = 0.5 * (a + a.T) + np.diag(rng.uniform(-2, 2, norb))
        if ed_mode == "normal":
            hb[0, 0, :, :, k] = a
        elif ed_mode == "superc":
            d = rng.uniform(-0.2, 0.2, (norb, norb))
            d = 0.5 * (d + d.T)
            hb[0, 0, :, :, k] = a          # Nambu: [[ h, Delta ], [ Delta^+, -h^T ]]
            hb[1, 1, :, :, k] = -a.T
            hb[0, 1, :, :, k] = d
            hb[1, 0, :, :, k] = d.conj().T
        else:
            f = rng.uniform(-0.3, 0.3, (norb, norb)) + 1j * rng.uniform(-0.3, 0.3, (norb, norb))
            b = a + np.diag(rng.uniform(-0.3, 0.3, norb))
            hb[0, 0, :, :, k] = a
            hb[1, 1, :, :, k] = b
            hb[0, 1, :, :, k] = f
            hb[1, 0, :, :, k] = f.conj().T
    if bath_type == "replica":
        v = np.broadcast_to(rng.uniform(0.1, 0.6, nbath), (nspin, norb, nbath)).copy()
    else:
        v = rng.uniform(0.1, 0.6, (nspin, norb, nbath))
        if nspin == 2 and ed_mode != "nonsu2":
            v[1] = v[0]
    par = dict(ed_mode=ed_mode, bath_type=bath_type, norb=norb, nbath=nbath, nspin=nspin, hfmode=True, xmu=0.0,
               uloc=tuple([2.0] * norb), ust=1.5 if norb > 1 else 0.0, jh=0.25 if norb > 1 else 0.0,
               jx=jxp if norb > 1 else 0.0, jp=jxp if norb > 1 else 0.0)
    par.update(over)
    return _replica_pair(ed_mode, bath_type, norb, nbath, nspin, hl, hb, v, par)


def replica_golden_models(inp):
    """The impurity problems of test/src/{REPLICA,GENERAL}_{NORMAL,SUPERC,NONSU2}: Norb=2, Nbath=2; the
    symmetry basis and initial lambdas set in ed_replica_*.f90 / ed_general_*.f90 (:47-95), the init_dmft_bath
    start values (ED_BATH_DMFT.f90:246-290: V=max(0.1,1/sqrt(Nbath)); equal lambdas on a diagonal basis
    matrix get the +-ed_offset_bath spread) and Hloc = Delta*sigma_z (Mh*Gamma5 in nonsu2)."""
    mode, bath, norb, nb = inp["ED_MODE"], inp["BATH_TYPE"], 2, 2
    nspin = 2 if mode == "nonsu2" else 1
    s0 = np.eye(2, dtype=complex)
    sx = np.array([[0, 1], [1, 0]], complex)
    sz = np.diag([1.0, -1.0]).astype(complex)

    def so(m4, n1):  # kron(sigma, tau) -> [is, js, iorb, jorb]
        out = np.zeros((n1, n1, norb, norb), complex)
        for i in range(n1):
            for j in range(n1):
                out[i, j] = m4[i * norb:(i + 1) * norb, j * norb:(j + 1) * norb]
        return out

    lam1 = np.array([-1.0 + 2.0 * i / (nb - 1) for i in range(nb)])
    if mode == "normal":
        basis, lam, n1 = [so(np.kron(s0, s0)[:2, :2], 1), so(np.kron(s0, sx)[:2, :2], 1)], [lam1, np.full(nb, 0.1)], 1
        hl = np.zeros((1, 1, 2, 2), complex)
        hl[0, 0] = inp["DELTA"] * sz
    elif mode == "superc":
        basis = [so(np.kron(sz, s0), 2), so(np.kron(sx, s0), 2), so(np.kron(sx, sx), 2)]
        lam, n1 = [lam1, np.full(nb, 0.1), np.full(nb, 0.2)], 2
        hl = np.zeros((1, 1, 2, 2), complex)
        hl[0, 0] = inp["DELTA"] * sz
    else:
        sb, mh = 0.01, inp["MH"]          # SB_FIELD of the two NONSU2 inputs
        off = np.linspace(-0.1, 0.1, nb)  # ED_OFFSET_BATH
        basis = [so(np.kron(s0, sz), 2), so(np.kron(s0, sx), 2), so(np.kron(sz, sx), 2), so(np.kron(sx, sx), 2)]
        lam, n1 = [mh + off, np.full(nb, sb), np.full(nb, sb), np.full(nb, -sb)], 2
        hl = so(mh * np.kron(s0, sz), 2)
    hb = np.zeros((n1, n1, norb, norb, nb), complex)
    for b, l in zip(basis, lam):
        for k in range(nb):
            hb[..., k] += l[k] * b
    v = np.full((nspin, norb, nb), max(0.1, 1.0 / np.sqrt(nb)))
    par = dict(ed_mode=mode, bath_type=bath, norb=norb, nbath=nb, nspin=nspin, hfmode=True, xmu=0.0,
               uloc=tuple(inp["ULOC"]), ust=inp["UST"], jh=inp["JH"], jx=inp["JX"], jp=inp["JP"])
    return _replica_pair(mode, bath, norb, nb, nspin, hl, hb, v, par)


def rel_err(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    den = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0))


def csr_to_dense(rowptr, col, val, ncol):
    nrow = len(rowptr) - 1
    out = np.zeros((nrow, ncol), dtype=np.asarray(val).dtype)
    for i in range(nrow):
        for k in range(rowptr[i], rowptr[i + 1]):
            out[i, col[k]] += val[k]
    return out


def make_jz_models(nbath: int, seed: int = 0, jx: float = 0.25):
    """A three-orbital nonsu2 problem that conserves Jz = Lz + Sz with Lzdiag = [-1, +1, 0] (ED_VARS_GLOBAL.f90:283),
    replica bath (levels iorb + Norb * ibath, the labelling of build_sector's Jz branch): impHloc and the replica
    matrices are real diagonals plus the two couplings between single-particle states of equal jz -- (orb 1, up) with
    (orb 3, down), jz = -1/2, and (orb 3, up) with (orb 2, down), jz = +1/2 -- as spin-orbit coupling produces them;
    Uloc, Ust, Jh and the spin-exchange Jx conserve Jz, the pair hopping Jp does not and is off."""
    rng = np.random.default_rng(20260704 + seed)
    norb = 3

    def jz_matrix(scale):
        h = np.zeros((2, 2, norb, norb), complex)
        for s_ in range(2):
            h[s_, s_] = np.diag(rng.uniform(-1, 1, norb) * scale)
        for (sa, a, sb, b) in ((0, 0, 1, 2), (0, 2, 1, 1)):       # (spin, orbital) pairs of equal jz
            z = scale * 0.4 * (rng.standard_normal() + 1j * rng.standard_normal())
            h[sa, sb, a, b] = z
            h[sb, sa, b, a] = np.conj(z)
        return h
    hl = jz_matrix(0.5)
    hb = np.zeros((2, 2, norb, norb, nbath), complex)
    for k in range(nbath):
        hb[..., k] = jz_matrix(1.0)
    v = np.tile(rng.uniform(0.1, 0.6, nbath), (2, norb, 1))
    par = dict(ed_mode="nonsu2", bath_type="replica", norb=norb, nbath=nbath, nspin=2, hfmode=True, xmu=0.0,
               uloc=tuple([2.0] * norb), ust=1.5, jh=0.25, jx=jx, jp=0.0)
    return _replica_pair("nonsu2", "replica", norb, nbath, 2, hl, hb, v, par)
