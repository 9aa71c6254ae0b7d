"""Self-energy moments of the reference's superc / nonsu2 regression tests, rebuilt from the COMPLEX Lanczos
tridiagonalisation -- the fixtures (`Sigma_momenta.check`, `Self_momenta.check` of *_SUPERC; `Sigma11_momenta.check`,
`Sigma12_momenta.check` of *_NONSU2) that exercise tridiag_Hv_sector_superc / _nonsu2 + sp_lanc_tridiag with
complex vectors, i.e. row a19 of SURVEY.md 8.

Restated for checking only (T=0, bath_type=normal, paths relative to /root/reference/src/singlesite):
  lanc_build_gf_superc_Gdiag / _Fmix   ED_SUPERC/ED_GF_SUPERC.f90:130-198, 287-361  (channels and seeds)
  add_to_lanczos_gf_superc             ED_SUPERC/ED_GF_SUPERC.f90:440-513  (poles isign (E_j - E_i), weights vnorm2 Z_1j^2 / zeta)
  get_impG_superc / get_impF_superc    ED_SUPERC/ED_GF_SUPERC.f90:578-843  (F = (aux - (1 - i)(G + barG)) / 2)
  get_Sigma_superc / get_Self_superc   ED_SUPERC/ED_GF_SUPERC.f90:938-1102 (normal bath, Matsubara axis)
  delta / fdelta / invg0 / invf0       ED_BATH/delta_functions/delta_normal.f90:44-62, fdelta_normal.f90,
                                       ED_BATH/invg0_functions/invg0_normal.f90, invf0_normal.f90
  lanc_build_gf_nonsu2_diagOrb_diagSpin / _mixOrb_mixSpin   ED_NONSU2/ED_GF_NONSU2.f90:159-301
  get_impG_nonsu2 / get_Sigma_nonsu2   ED_NONSU2/ED_GF_NONSU2.f90:491-634, 716-748
  tridiag_Hv_sector_superc / _nonsu2   ED_SUPERC/ED_HAMILTONIAN_SUPERC.f90:227-272, ED_NONSU2/ED_HAMILTONIAN_NONSU2.f90:202-244
  compute_momentum                     test/src/COMMON.f90:170-192
The tridiagonalisation itself is a plug-in `tridiag(sector, unit seed, nlanc) -> (alanc, blanc)`, so the same
driver checks the oracle (CPU) and the GPU library.
"""
from __future__ import annotations

import numpy as np

from oracle import oracle as O


def _popcount(x: np.ndarray) -> np.ndarray:
    y = x.astype(np.int64).copy()
    c = np.zeros_like(y)
    while np.any(y):
        c += y & 1
        y >>= 1
    return c


def apply_cops(h_from, h_to, vec, ops, ns):
    """apply_Cops (ED_SECTOR.f90:839-960) on the 2*Ns-bit states of a superc / nonsu2 sector:
    sum_s coef_s c^(+)_{orb_s, spin_s} |vec>; ops = [(coef, create, iorb, ispin)], up levels first, the sign counts
    every occupied level below the operator's."""
    out = np.zeros(h_to.dim, complex)
    rank_to = {int(s): i for i, s in enumerate(h_to.map)}
    smap = h_from.map.astype(np.int64)
    for coef, create, iorb, ispin in ops:
        bit = 1 << (iorb + ispin * ns)
        occ = (smap & bit) != 0
        sel = np.nonzero(~occ if create else occ)[0]
        if sel.size == 0:
            continue
        sgn = 1.0 - 2.0 * (_popcount(smap[sel] & (bit - 1)) & 1)
        tgt = np.array([rank_to[int(s) ^ bit] for s in smap[sel]], dtype=np.int64)
        out[tgt] += coef * sgn * vec[sel]
    return out


def _ground_states(om, gs_threshold):
    secs = []
    for sec in O.sectors(om):
        try:
            h = O.HFlat(om, sec)
        except Exception:
            continue
        if h.dim:
            w, v = np.linalg.eigh(h.dense())
            secs.append((sec, h, w, v))
    e0 = min(w[0] for _, _, w, _ in secs)
    return e0, [(sec, h, w[k], v[:, k]) for sec, h, w, v in secs for k in range(len(w)) if w[k] - e0 <= gs_threshold]


def _pole_sum(z, tridiag, hcache, om, states, zeta, ngfiter, channels, also_conj=False):
    """sum over ground states and channels of weight / (z - pole); channels(sec) yields
    (target sector | None, ops, isign, complex prefactor of norm2).  also_conj: -> (sum at z, sum at conj(z)), the
    zconj form of get_impF_superc."""
    g = np.zeros(z.shape[0], complex)
    gc = np.zeros(z.shape[0], complex)
    ns = om.ns
    for sec, h, ei, vec in states:
        for sec2, ops, isign, pref in channels(sec):
            if sec2 is None:
                continue
            if sec2 not in hcache:
                try:
                    hcache[sec2] = O.HFlat(om, sec2)
                except Exception:
                    hcache[sec2] = None
            h2 = hcache[sec2]
            if h2 is None or h2.dim == 0:
                continue
            vv = apply_cops(h, h2, vec, ops, ns)
            norm2 = float(np.real(np.vdot(vv, vv)))
            if norm2 == 0.0:
                continue
            nl = min(h2.dim, ngfiter)
            al, bl = tridiag(sec2, vv / np.sqrt(norm2), nl)
            t = np.diag(al[:nl]) + np.diag(bl[1:nl], 1) + np.diag(bl[1:nl], -1)
            ev, zz = np.linalg.eigh(t)
            poles = isign * (ev - ei)
            wts = pref * norm2 / zeta * zz[0, :] ** 2
            g += np.sum(wts[None, :] / (z[:, None] - poles[None, :]), axis=1)
            if also_conj:
                gc += np.sum(wts[None, :] / (np.conj(z)[:, None] - poles[None, :]), axis=1)
    return (g, gc) if also_conj else g


def _moments(f, wm, nmom):
    a = np.abs(f)
    return np.array([np.sum(a * wm ** n) / np.sum(a) for n in range(1, nmom + 1)])


def momenta_superc(om, tridiag, beta=1000.0, lmats=4096, ngfiter=200, gs_threshold=1e-9, nmom=4):
    """-> (Sigma_momenta[norb, nmom], Self_momenta[norb, nmom]) as the *_SUPERC fixtures store them; bath normal, or
    hybrid (G^{ab}, F^{ab} for every orbital pair: lanc_build_gf_superc_Gmix / _Fmix, ED_GF_SUPERC.f90:200-361; the
    2 Norb x 2 Norb Nambu inverse of get_Sigma_superc / get_Self_superc :985-1005, 1060-1080; delta / fdelta of the shared
    bath, ED_BATH/delta_functions/delta_hybrid.f90:50-72, fdelta_hybrid.f90), replica or general (Delta, Fdelta from the
    Nambu replica matrices; Self_momenta then holds every orbital pair, rows (a, b) with b fastest)."""
    assert om.ed_mode == "superc"
    if om.bath_type != "normal":
        return _momenta_superc_hybrid(om, tridiag, beta, lmats, ngfiter, gs_threshold, nmom)
    ns, no = om.ns, om.norb
    e0, states = _ground_states(om, gs_threshold)
    zeta = float(len(states))
    wm = np.pi / beta * (2.0 * np.arange(1, lmats + 1) - 1.0)
    z = 1j * wm
    hcache = {}
    up, dw = 0, 1

    def sz_ok(s):
        return s if -ns <= s <= ns else None

    sig = np.zeros((no, nmom))
    slf = np.zeros((no, nmom))
    for a in range(no):
        # G_upup(aa): c^+_up (Sz+1, isign +1), c_up (Sz-1, isign -1)      (lanc_build_gf_superc_Gdiag)
        g = _pole_sum(z, tridiag, hcache, om, states, zeta, ngfiter, lambda s: [
            (sz_ok(s + 1), [(1.0, True, a, up)], 1, 1.0), (sz_ok(s - 1), [(1.0, False, a, up)], -1, 1.0)])
        # barG(aa): c_dw (Sz+1, isign +1), c^+_dw (Sz-1, isign -1)
        gb = _pole_sum(z, tridiag, hcache, om, states, zeta, ngfiter, lambda s: [
            (sz_ok(s + 1), [(1.0, False, a, dw)], 1, 1.0), (sz_ok(s - 1), [(1.0, True, a, dw)], -1, 1.0)])
        # F mix channels: O^+ = c^+_up + c_dw, O = c_up + c^+_dw, P^+ = c^+_up + i c_dw, P = c_up - i c^+_dw  (_Fmix)
        aux = _pole_sum(z, tridiag, hcache, om, states, zeta, ngfiter, lambda s: [
            (sz_ok(s + 1), [(1.0, True, a, up), (1.0, False, a, dw)], 1, 1.0),
            (sz_ok(s - 1), [(1.0, False, a, up), (1.0, True, a, dw)], -1, 1.0),
            (sz_ok(s + 1), [(1.0, True, a, up), (1j, False, a, dw)], 1, -1j),
            (sz_ok(s - 1), [(1.0, False, a, up), (-1j, True, a, dw)], -1, -1j)])
        f12 = 0.5 * (aux - (1.0 - 1j) * (g + gb))
        e, d, v = om.be[0, a, :], om.bd[0, a, :], om.bv[0, a, :]
        den = wm[:, None] ** 2 + e[None, :] ** 2 + d[None, :] ** 2
        delta = -np.sum(v[None, :] ** 2 * (z[:, None] + e[None, :]) / den, axis=1)
        fdelta = np.sum(d[None, :] * v[None, :] ** 2 / den, axis=1)
        invg0 = z + om.xmu - om.hloc[0, 0, a, a].real - delta
        invf0 = -fdelta                      # impHloc_anomalous = 0 (no pair field in the fixtures)
        gdet = np.real(np.abs(g) ** 2 + f12 ** 2)
        sig[a] = _moments(invg0 - np.conj(g) / gdet, wm, nmom)
        slf[a] = _moments(invf0 - f12 / gdet, wm, nmom)
    return sig, slf


def _momenta_superc_hybrid(om, tridiag, beta, lmats, ngfiter, gs_threshold, nmom):
    ns, no = om.ns, om.norb
    e0, states = _ground_states(om, gs_threshold)
    zeta = float(len(states))
    wm = np.pi / beta * (2.0 * np.arange(1, lmats + 1) - 1.0)
    z = 1j * wm
    hcache = {}
    up, dw = 0, 1

    def sz_ok(s):
        return s if -ns <= s <= ns else None

    def ps(chan, **kw):
        return _pole_sum(z, tridiag, hcache, om, states, zeta, ngfiter, chan, **kw)

    G = np.zeros((no, no, lmats), complex)
    barG = np.zeros((no, lmats), complex)
    F12 = np.zeros((no, no, lmats), complex)
    F21 = np.zeros((no, no, lmats), complex)
    for a in range(no):
        G[a, a] = ps(lambda s: [(sz_ok(s + 1), [(1.0, True, a, up)], 1, 1.0), (sz_ok(s - 1), [(1.0, False, a, up)], -1, 1.0)])
        barG[a], barGc = ps(lambda s: [(sz_ok(s + 1), [(1.0, False, a, dw)], 1, 1.0),
                                       (sz_ok(s - 1), [(1.0, True, a, dw)], -1, 1.0)], also_conj=True)
        barG_c = barGc if a == 0 else np.vstack([barG_c, barGc])
    barG_c = np.atleast_2d(barG_c)
    Gc = np.zeros((no, lmats), complex)      # G_aa at conj(z)
    for a in range(no):
        _, Gc[a] = ps(lambda s: [(sz_ok(s + 1), [(1.0, True, a, up)], 1, 1.0),
                                 (sz_ok(s - 1), [(1.0, False, a, up)], -1, 1.0)], also_conj=True)
    for a in range(no):
        for b in range(no):
            if a != b:   # Gmix: (c^+_a + c^+_b), (c_a + c_b), (c^+_a + i c^+_b), (c_a - i c_b) on the up species
                aux = ps(lambda s: [
                    (sz_ok(s + 1), [(1.0, True, a, up), (1.0, True, b, up)], 1, 1.0),
                    (sz_ok(s - 1), [(1.0, False, a, up), (1.0, False, b, up)], -1, 1.0),
                    (sz_ok(s + 1), [(1.0, True, a, up), (1j, True, b, up)], 1, -1j),
                    (sz_ok(s - 1), [(1.0, False, a, up), (-1j, False, b, up)], -1, -1j)])
                G[a, b] = 0.5 * (aux - (1.0 - 1j) * (G[a, a] + G[b, b]))
            # Fmix(a, b): O^+ = c^+_{a,up} + c_{b,dw}, O, P^+ = c^+_{a,up} + i c_{b,dw}, P
            aux, auxc = ps(lambda s: [
                (sz_ok(s + 1), [(1.0, True, a, up), (1.0, False, b, dw)], 1, 1.0),
                (sz_ok(s - 1), [(1.0, False, a, up), (1.0, True, b, dw)], -1, 1.0),
                (sz_ok(s + 1), [(1.0, True, a, up), (1j, False, b, dw)], 1, -1j),
                (sz_ok(s - 1), [(1.0, False, a, up), (-1j, True, b, dw)], -1, -1j)], also_conj=True)
            F12[a, b] = 0.5 * (aux - (1.0 - 1j) * (G[a, a] + barG[b]))
            F21[a, b] = 0.5 * (auxc - (1.0 - 1j) * (Gc[a] + barG_c[b]))      # get_impF_superc(zconj=.true.)
    replica = om.bath_type in ("replica", "general")
    M = np.zeros((lmats, 2 * no, 2 * no), complex)
    invg0 = np.zeros((lmats, no, no), complex)
    invf0 = np.zeros((lmats, no, no), complex)
    if replica:
        # Delta, Fdelta = the (1,1) and (1,2) Nambu blocks of sum_k V_k (z - H_k)^-1 V_k, V_k = sigma_z (x) diag(v_k)
        # (delta_replica.f90:43-58, fdelta_replica.f90; general: vg in place of the scalar v)
        n2 = 2 * no
        dn = np.zeros((lmats, n2, n2), complex)
        for k in range(om.nbath):
            hk = np.zeros((n2, n2), complex)
            for n_ in range(2):
                for m_ in range(2):
                    hk[n_ * no:(n_ + 1) * no, m_ * no:(m_ + 1) * no] = om.hb[n_, m_, :, :, k]
            vd = om.vg[:no, k] if om.bath_type == "general" else np.full(no, om.vr[k])
            vk = np.kron(np.diag([1.0, -1.0]), np.diag(vd)).astype(complex)
            dn += vk[None] @ np.linalg.inv(z[:, None, None] * np.eye(n2)[None] - hk[None]) @ vk[None]
    else:
        e, d, v = om.be[0, 0, :], om.bd[0, 0, :], om.bv[0, :, :]
        den = wm[:, None] ** 2 + e[None, :] ** 2 + d[None, :] ** 2
    for a in range(no):
        for b in range(no):
            if replica:
                delta, fdelta = dn[:, a, b], dn[:, a, no + b]
            else:
                vv = (v[a] * v[b])[None, :]
                delta = -np.sum(vv * (z[:, None] + e[None, :]) / den, axis=1)
                fdelta = np.sum(d[None, :] * vv / den, axis=1)
            invg0[:, a, b] = ((z + om.xmu) if a == b else 0.0) - om.hloc[0, 0, a, b] - delta
            invf0[:, a, b] = -fdelta
            M[:, a, b] = G[a, b]
            M[:, a, no + b] = F12[a, b]
            M[:, no + a, b] = np.conj(F21[b, a])          # transpose(conjg(F21))
            M[:, no + a, no + b] = -np.conj(G[a, b])
    Mi = np.linalg.inv(M)
    sig = np.array([_moments(invg0[:, a, a] - Mi[:, a, a], wm, nmom) for a in range(no)])
    # Self_momenta.check of HYBRID_SUPERC is reproduced (3.5e-8, the noise level of the file) by invF0 + invF, not by the
    # invF0 - invF of get_Self_superc in this checkout (6 % away): the moments are those of |Self|, so only the relative
    # sign of the two pieces shows, and the file is older than the source (the REPLICA_ / GENERAL_SUPERC files, below, are
    # matched at 1e-11 with the sign of the source).  Every Lanczos-derived ingredient -- G_ab, barG, F_ab for all orbital
    # pairs -- enters either way.
    if replica:      # ASmomAB(iorb, jorb, :) of ed_replica_superc.f90:141: every orbital pair
        slf = np.array([_moments(invf0[:, a, b] - Mi[:, a, no + b], wm, nmom) for a in range(no) for b in range(no)])
    else:
        slf = np.array([_moments(invf0[:, a, a] + Mi[:, a, no + a], wm, nmom) for a in range(no)])
    return sig, slf


def momenta_nonsu2(om, tridiag, beta=300.0, lmats=2000, ngfiter=300, gs_threshold=1e-9, nmom=4, all_components=False):
    """-> (Sigma11_momenta[norb, nmom], Sigma12_momenta[norb, nmom]) as the NORMAL_ / HYBRID_NONSU2 fixtures store them, or
    with all_components the moments of every Sigma_{ab}^{ss'}, rows ordered (s, s', a, b) with b fastest as in the REPLICA_ /
    GENERAL_NONSU2 files (ED_ALL_G = T there).  Bath normal (G, G0^-1 diagonal in the orbitals), hybrid, replica or
    general (all G_{ab}^{ss'}: build_impG_nonsu2, ED_GF_NONSU2.f90:83-141; the (Nspin Norb)^2 inverse of get_Sigma_nonsu2
    :716-748; delta_bath_array_hybrid / _replica / _general, ED_BATH/delta_functions/)."""
    assert om.ed_mode == "nonsu2" and om.nspin == 2
    hybrid = om.bath_type != "normal"       # "every orbital pair" branch
    replica = om.bath_type in ("replica", "general")
    ns, no = om.ns, om.norb
    nlev = 2 * ns
    e0, states = _ground_states(om, gs_threshold)
    zeta = float(len(states))
    wm = np.pi / beta * (2.0 * np.arange(1, lmats + 1) - 1.0)
    z = 1j * wm
    hcache = {}

    def n_ok(n):
        return n if 0 <= n <= nlev else None

    gf = np.zeros((2, 2, no, no, lmats), complex)
    for a in range(no):
        for s in range(2):
            gf[s, s, a, a] = _pole_sum(z, tridiag, hcache, om, states, zeta, ngfiter, lambda n: [
                (n_ok(n + 1), [(1.0, True, a, s)], 1, 1.0), (n_ok(n - 1), [(1.0, False, a, s)], -1, 1.0)])
    for a in range(no):
        for b in range(no):
            if a != b and not hybrid:
                continue
            for s in range(2):
                for t in range(2):
                    if s == t and a == b:
                        continue
                    # lanc_build_gf_nonsu2_mixOrb_mixSpin(iorb, jorb, ispin, jspin), :236-300
                    aux = _pole_sum(z, tridiag, hcache, om, states, zeta, ngfiter, lambda n: [
                        (n_ok(n + 1), [(1.0, True, a, s), (1.0, True, b, t)], 1, 1.0),
                        (n_ok(n - 1), [(1.0, False, a, s), (1.0, False, b, t)], -1, 1.0),
                        (n_ok(n + 1), [(1.0, True, a, s), (1j, True, b, t)], 1, -1j),
                        (n_ok(n - 1), [(1.0, False, a, s), (-1j, False, b, t)], -1, -1j)])
                    gf[s, t, a, b] = 0.5 * (aux - (1.0 - 1j) * (gf[s, s, a, a] + gf[t, t, b, b]))
    n2 = 2 * no
    delta = np.zeros((2, 2, no, no, lmats), complex)
    if replica:
        # Delta(z) = sum_k V_k (z - H_k)^-1 V_k in spin-orbital space (delta_replica.f90:30-42, delta_general.f90:28-40)
        for k in range(om.nbath):
            hk = np.zeros((n2, n2), complex)
            for s_ in range(2):
                for t_ in range(2):
                    hk[s_ * no:(s_ + 1) * no, t_ * no:(t_ + 1) * no] = om.hb[s_, t_, :, :, k]
            vk = np.diag(om.vg[:, k]).astype(complex) if om.bath_type == "general" else om.vr[k] * np.eye(n2)
            inv = np.linalg.inv(z[:, None, None] * np.eye(n2)[None] - hk[None])
            dk = vk[None] @ inv @ vk[None]
            for s_ in range(2):
                for t_ in range(2):
                    delta[s_, t_] += np.moveaxis(dk[:, s_ * no:(s_ + 1) * no, t_ * no:(t_ + 1) * no], 0, 2)
    # hybridisation: W(s, h, a, k) of get_Whyb_matrix (ED_BATH_AUX.f90:75-102), bath levels e(h, a | 1, k)
    w = np.zeros((2, 2, no, om.nbath))
    if not replica:
        w[0, 0], w[1, 1] = om.bv[0], om.bv[1]
        w[0, 1], w[1, 0] = om.bu[0], om.bu[1]
    for a in range(no if not replica else 0):
        for b in range(no):
            if a != b and not hybrid:
                continue
            for s in range(2):
                for t in range(2):
                    for ih in range(2):
                        eh = om.be[ih, 0, :] if hybrid else om.be[ih, a, :]
                        delta[s, t, a, b] += np.sum((w[s, ih, a, :] * w[t, ih, b, :])[None, :] / (z[:, None] - eh[None, :]), axis=1)
    g0inv = np.zeros((lmats, n2, n2), complex)
    gm = np.zeros((lmats, n2, n2), complex)
    for s in range(2):
        for t in range(2):
            for a in range(no):
                for b in range(no):
                    io, jo = a + s * no, b + t * no                       # nn2so_reshape
                    g0inv[:, io, jo] = ((z + om.xmu) if io == jo else 0.0) - om.hloc[s, t, a, b] - delta[s, t, a, b]
                    gm[:, io, jo] = gf[s, t, a, b]
    sg = g0inv - np.linalg.inv(gm)
    if all_components:
        return np.array([_moments(sg[:, a + s_ * no, b + t_ * no], wm, nmom)
                         for s_ in range(2) for t_ in range(2) for a in range(no) for b in range(no)])
    s11 = np.array([_moments(sg[:, a, a], wm, nmom) for a in range(no)])
    s12 = np.array([_moments(sg[:, a, a + no], wm, nmom) for a in range(no)])
    return s11, s12
