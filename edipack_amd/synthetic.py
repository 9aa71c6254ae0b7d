"""Structure-preserving synthetic impurity problems for the BASELINE.json configs.

The true sector bases and sparsity patterns of each config are kept; the model parameters are
drawn from a seeded RNG (seed = 20260630 + cfg): bath levels e~U(-2,2), hybridisations
v~U(0.1,0.6), Uloc=2, Ust=1.5, Jh=Jx=Jp=0.25, pair amplitudes d~U(-0.1,0.1) (superc),
spin-flip u~U(0,0.3) + random Hermitian impHloc (nonsu2), xmu=0, hfmode=T, Nph=0.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .hamiltonian import ImpurityModel


@dataclass(frozen=True)
class Workload:
    name: str
    cfg: int
    ed_mode: str
    bath_type: str
    norb: int
    nbath: int
    sector: tuple | int
    note: str
    direct: bool = False   # ed_sparse_H=F: on-the-fly H*v


WORKLOADS = {
    # BASELINE.json configs[0..3]; sizes from SURVEY.md 8 (table "Concrete sizes")
    "cfg1": Workload("cfg1", 1, "normal", "normal", 1, 4, (2, 3), "Ns=5, Dim=100 (plumbing)"),
    "cfg2": Workload("cfg2", 2, "normal", "normal", 2, 6, (7, 7), "Ns=14, (7,7), Dim=11 778 624, real"),
    "cfg3": Workload("cfg3", 3, "normal", "hybrid", 3, 8, (5, 6), "Ns=11, Dim=213 444, real (cache resident)"),
    "cfg4": Workload("cfg4", 4, "superc", "hybrid", 2, 8, 0, "Ns=10, Sz=0, Dim=184 756, complex flat CSR"),
    # scale-up ladder of the 3-orbital hybrid structure (Ns = 15/16/17)
    "cfg3_ns15": Workload("cfg3_ns15", 3, "normal", "hybrid", 3, 12, (7, 8), "Ns=15, Dim=41 409 225"),
    "cfg3_ns16": Workload("cfg3_ns16", 3, "normal", "hybrid", 3, 13, (8, 8), "Ns=16, Dim=165 636 900"),
    "cfg3_ns17": Workload("cfg3_ns17", 3, "normal", "hybrid", 3, 14, (8, 9), "Ns=17, Dim=590 976 100"),
    # the same size with a replica bath (3 orbitals x 4 replicas: hops between the bath levels of a replica)
    "cfg3_replica_ns15": Workload("cfg3_replica_ns15", 3, "normal", "replica", 3, 4, (7, 8), "Ns=15, replica bath, Dim=41 409 225"),
    # scale-up of the flat-CSR modes
    "cfg4_ns12": Workload("cfg4_ns12", 4, "superc", "hybrid", 2, 10, 0, "Ns=12, Sz=0, Dim=2 704 156"),
    # BASELINE.json configs[4]: 3 orbitals, Nbath=10 (hybrid), nonsu2, on-the-fly kernel
    "cfg5": Workload("cfg5", 5, "nonsu2", "hybrid", 3, 10, 13, "Ns=13, N=13, Dim=10 400 600, complex, direct", True),
    "cfg5_ns11": Workload("cfg5_ns11", 5, "nonsu2", "hybrid", 3, 8, 11, "Ns=11, N=11, Dim=705 432, direct", True),
    "cfg5_stored": Workload("cfg5_stored", 5, "nonsu2", "hybrid", 3, 10, 13, "Ns=13, N=13, Dim=10 400 600, complex, stored"),
    "cfg5_stored_ns11": Workload("cfg5_stored_ns11", 5, "nonsu2", "hybrid", 3, 8, 11, "Ns=11, N=11, Dim=705 432"),
}


def build_workload(w: Workload, handover: bool = False, **shard):
    """SectorHamiltonian of a workload (single shard unless dw_first/dw_count or row_first/row_count given).
    handover (normal mode): through edigpu_normal_create, i.e. from the explicit arrays (spH0d, spH0ups, spH0dws,
    spH0nd) as the reference's ed_buildh_normal_main leaves them, instead of from the model."""
    from .hamiltonian import SectorHamiltonian
    m = synthetic_model(w)
    if w.ed_mode == "normal":
        h = SectorHamiltonian.normal_from_model(m, *w.sector, **shard)
        if not handover:
            return h
        hd, up, dw, nd = h.export_normal()
        du, dd, first, cnt = h.dim_up, h.dim_dw, h.row_first // h.dim_up, h.nloc // h.dim_up
        h.destroy()
        return SectorHamiltonian.normal_from_arrays(du, dd, hd, up, dw, nd if nd[0][-1] > 0 else None,
                                                    dw_first=first, dw_count=cnt)
    if w.direct:
        return SectorHamiltonian.direct_from_model(m, w.sector, **shard)
    return SectorHamiltonian.flat_from_model(m, w.sector, **shard)


def synthetic_model(w: Workload) -> ImpurityModel:
    rng = np.random.default_rng(20260630 + w.cfg)
    nspin = 2 if w.ed_mode == "nonsu2" else 1
    nfoo = 1 if w.bath_type == "hybrid" else w.norb
    be = rng.uniform(-2.0, 2.0, (nspin, nfoo, w.nbath))
    bv = rng.uniform(0.1, 0.6, (nspin, w.norb, w.nbath))
    bd = rng.uniform(-0.1, 0.1, (nspin, nfoo, w.nbath)) if w.ed_mode == "superc" else None
    bu = rng.uniform(0.0, 0.3, (nspin, w.norb, w.nbath)) if w.ed_mode == "nonsu2" else None
    hl = np.zeros((nspin, nspin, w.norb, w.norb), complex)
    if w.ed_mode == "nonsu2":
        be[1], bv[1] = be[0], bv[0]
        a = rng.standard_normal((2 * w.norb, 2 * w.norb)) + 1j * rng.standard_normal((2 * w.norb, 2 * w.norb))
        a = 0.2 * (a + a.conj().T)
        for s in range(2):
            for t in range(2):
                hl[s, t] = a[s * w.norb:(s + 1) * w.norb, t * w.norb:(t + 1) * w.norb]
    multi = w.norb > 1
    if w.bath_type in ("replica", "general"):
        # per replica a symmetric Norb x Norb matrix (levels e~U(-2,2) on the diagonal, inter-orbital hops ~U(-0.2,0.2)),
        # one hybridisation per replica (item(k)%v) -- the structure build_Hreplica gives with a symmetric basis
        n1 = nspin
        hb = np.zeros((n1, n1, w.norb, w.norb, w.nbath), complex)
        for k in range(w.nbath):
            a = rng.uniform(-0.4, 0.4, (w.norb, w.norb))
            a = 0.5 * (a + a.T) + np.diag(rng.uniform(-2.0, 2.0, w.norb))
            for sp in range(n1):
                hb[sp, sp, :, :, k] = a
        bvr = np.broadcast_to(rng.uniform(0.1, 0.6, w.nbath), (nspin, w.norb, w.nbath)).copy()
        return ImpurityModel(ed_mode=w.ed_mode, bath_type=w.bath_type, norb=w.norb, nbath=w.nbath, nspin=nspin,
                             hfmode=True, xmu=0.0, uloc=np.full(w.norb, 2.0), ust=1.5 if multi else 0.0,
                             jh=0.25 if multi else 0.0, jx=0.25 if multi else 0.0, jp=0.25 if multi else 0.0,
                             hloc=hl, bv=bvr, hb=hb)
    return ImpurityModel(ed_mode=w.ed_mode, bath_type=w.bath_type, norb=w.norb, nbath=w.nbath, nspin=nspin,
                         hfmode=True, xmu=0.0, uloc=np.full(w.norb, 2.0), ust=1.5 if multi else 0.0,
                         jh=0.25 if multi else 0.0, jx=0.25 if multi else 0.0, jp=0.25 if multi else 0.0,
                         hloc=hl, be=be, bv=bv, bd=bd, bu=bu)
