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
