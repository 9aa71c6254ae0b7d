"""Self-energy moments of the reference's normal-mode regression tests, rebuilt from the Lanczos
tridiagonalisation -- the fixture (`Sigma_momenta.check`) that exercises tridiag_Hv_sector_normal +
sp_lanc_tridiag, i.e. rows a12 of SURVEY.md 8.

Restates, for checking only (T=0, Nspin=1, diagonal G):
  lanc_build_gf_normal_diag   ED_NORMAL/ED_GF_NORMAL.f90:131-177   (c^+ / c on every ground state, tridiag)
  add_to_lanczos_gf_normal    ED_NORMAL/ED_GF_NORMAL.f90:363-427   (poles = +-(E_j - E_gs), weights = norm2 Z_1j^2 / zeta)
  tridiag_Hv_sector_normal    ED_NORMAL/ED_HAMILTONIAN_NORMAL.f90:321-369 (normalise seed, Nlanc = min(Dim, lanc_ngfiter))
  Sigma = G0^-1 - G^-1 with G0^-1 = i w + xmu - Hloc_aa - sum_k V_ak^2 / (i w - e_ak)  (normal / hybrid bath)
  compute_momentum            test/src/COMMON.f90:170-192
The tridiagonalisation itself is delegated to `tridiag(sector, seed, nlanc) -> (alanc, blanc)`, so the
same driver checks the oracle (CPU) and the GPU library.
"""
from __future__ import annotations

import numpy as np

from oracle import oracle as O


def _popcount_below(x: np.ndarray, pos: int) -> np.ndarray:
    y = x & ((1 << pos) - 1)
    c = np.zeros_like(y)
    while np.any(y):
        c += y & 1
        y = y >> 1
    return c


def apply_c_up(h_from, h_to, vec, iorb, create):
    """apply_op_CDG / apply_op_C for spin up (ED_SECTOR.f90:465-536): |out> = c^(+)_{iorb,up} |vec>."""
    out = np.zeros(h_to.dim)
    rank_to = {int(s): i for i, s in enumerate(h_to.mapup)}
    bit = 1 << iorb
    occ = (h_from.mapup & bit) != 0
    sel = np.nonzero(~occ if create else occ)[0]
    sgn = 1.0 - 2.0 * (_popcount_below(h_from.mapup[sel], iorb) & 1)
    tgt = np.array([rank_to[int(s) ^ bit] for s in h_from.mapup[sel]], dtype=np.int64)
    v2 = vec.reshape(h_from.dimdw, h_from.dimup)
    o2 = out.reshape(h_to.dimdw, h_to.dimup)
    o2[:, tgt] = v2[:, sel] * sgn[None, :]
    return out


def sigma_momenta_normal(om, tridiag, beta=1000.0, lmats=4096, ngfiter=200, gs_threshold=1e-9, nmom=4):
    """-> array [norb, nmom] as Sigma_momenta.check stores it (orbital-major)."""
    assert om.ed_mode == "normal" and om.bath_type in ("normal", "hybrid", "replica", "general")
    secs = []
    for sec in O.sectors(om):
        h = O.HNormal(om, *sec)
        if h.dim:
            w, v = np.linalg.eigh(h.dense())
            secs.append((sec, h, w, v))
    e0 = min(w[0] for _, _, w, _ in secs)
    states = [(sec, h, w[k], v[:, k]) for sec, h, w, v in secs for k in range(len(w)) if w[k] - e0 <= gs_threshold]
    zeta = float(len(states))
    wm = np.pi / beta * (2.0 * np.arange(1, lmats + 1) - 1.0)
    z = 1j * wm
    ns = om.ns
    hcache = {}

    def sector_h(nup, ndw):
        if (nup, ndw) not in hcache:
            hcache[(nup, ndw)] = O.HNormal(om, nup, ndw)
        return hcache[(nup, ndw)]

    def g_of(ops):
        """sum over ground states and the +-1 particle channels of <O^+ (z -+ (H - E))^-1 O>, O = sum of
        c^(+)_{a,up} over a in ops (lanc_build_gf_normal_diag / _mix)."""
        g = np.zeros(lmats, complex)
        for (nup, ndw), h, ei, vec in states:
            for create, isign in ((True, 1), (False, -1)):
                nup2 = nup + (1 if create else -1)
                if nup2 < 0 or nup2 > ns:
                    continue
                h2 = sector_h(nup2, ndw)
                vv = sum(apply_c_up(h, h2, vec, a, create) for a in ops)
                norm2 = float(vv @ vv)
                if norm2 == 0.0:
                    continue
                nl = min(h2.dim, ngfiter)
                al, bl = tridiag((nup2, ndw), vv / np.sqrt(norm2), nl)
                t = np.diag(al[:nl]) + np.diag(bl[1:nl], 1) + np.diag(bl[1:nl], -1)
                ev, zz = np.linalg.eigh(t)
                poles = isign * (ev - ei)
                wts = norm2 / zeta * zz[0, :] ** 2
                g += np.sum(wts[None, :] / (z[:, None] - poles[None, :]), axis=1)
        return g

    no = om.norb
    gm = np.zeros((lmats, no, no), complex)
    for a in range(no):
        gm[:, a, a] = g_of([a])
    if om.bath_type != "normal":     # off-diagonal G only exists with a shared bath (ed_solve_offdiag_gf)
        for a in range(no):
            for b in range(a + 1, no):
                # (c_a + c_b) channel: G_ab = (G_mix - G_aa - G_bb)/2 (real symmetric), ED_GF_NORMAL.f90:96-110
                gab = 0.5 * (g_of([a, b]) - gm[:, a, a] - gm[:, b, b])
                gm[:, a, b] = gab
                gm[:, b, a] = gab
    g0inv = np.zeros((lmats, no, no), complex)
    if om.bath_type in ("replica", "general"):
        # delta_bath_array, ED_BATH/ED_BATH_FUNCTIONS: Delta(z) = sum_k V_k (z - H_k)^-1 V_k^T with the replica matrices
        # H_k = hb[0, 0, :, :, k]; replica: one amplitude per bath element, general: one per orbital (vg)
        delta = np.zeros((lmats, no, no), complex)
        for k in range(om.nbath):
            hk = np.asarray(om.hb)[0, 0, :, :, k]
            vk = np.full(no, om.vr[k]) if om.bath_type == "replica" else np.asarray(om.vg)[:no, k]
            inv = np.linalg.inv(z[:, None, None] * np.eye(no)[None] - hk[None])
            delta += vk[None, :, None] * inv * vk[None, None, :]
        hl = np.asarray(om.hloc)[0, 0].real
        g0inv = (z[:, None, None] + om.xmu) * np.eye(no)[None] - hl[None] - delta
    for a in range(no if om.bath_type in ("normal", "hybrid") else 0):
        for b in range(no):
            ea = om.be[0, 0 if om.bath_type == "hybrid" else a, :]
            if om.bath_type == "normal" and a != b:
                delta = 0.0
            else:
                delta = np.sum((om.bv[0, a, :] * om.bv[0, b, :])[None, :] / (z[:, None] - ea[None, :]), axis=1)
            g0inv[:, a, b] = (z + om.xmu if a == b else 0.0) - om.hloc[0, 0, a, b].real - delta
    sig = g0inv - np.linalg.inv(gm)
    out = np.zeros((no, nmom))
    for a in range(no):
        sa = np.abs(sig[:, a, a])
        for n in range(1, nmom + 1):
            out[a, n - 1] = np.sum(sa * wm ** n) / np.sum(sa)
    return out
