"""Ground-state observables of the reference's regression fixtures that need more than densities: `doubles.check`,
`energy.check`, `imp.check` (all twelve <BATH>_<MODE> directories), `phisc.check` (*_SUPERC) and `magX.check`
(NORMAL_/HYBRID_NONSU2).  Restated for checking only (T = 0; paths relative to /root/reference/src/singlesite):

  local_energy_normal / _superc / _nonsu2   ED_NORMAL/ED_OBSERVABLES_NORMAL.f90:491-940 and the two sister files --
        every accumulated term is <gs| X |gs> with X one term family of the Hamiltonian (the loops there are the
        builder's loops with v(i) v(j) in place of the insertion), so X is taken from the oracle's own builder on a model
        that keeps that family only:  Dust, Dund, Dse, Dph = the Ust / (Ust - Jh) / Jx / Jp families with unit couplings
        (ED_IO/get_doubles.f90), Eint = the interaction with the model's couplings, the Hartree shift = the HFMODE
        terms, <Hloc> = impHloc.  energy.check holds [Eint + Hartree, Eint, <Hloc>, Hartree] in that order.
  imp.check = [s2tot, Egs]: <(sum_a (n_a,up - n_a,dw) / 2)^2> and the ground-state energy (…OBSERVABLES_*.f90, ed_imp_info)
  phisc     ED_SUPERC/ED_OBSERVABLES_SUPERC.f90:202-249: Phi_ab from ||(a_dw + b^+_up) gs||^2 and ||(a_dw + i b^+_up) gs||^2
            through apply_Cops into the sector Sz + 1
  magX      ED_NONSU2/ED_OBSERVABLES_NONSU2.f90:244-296: ||(c_up + c_dw) gs||^2 - n_up - n_dw through apply_Cops into N - 1

The eigenvectors come from a plug-in `eigvec(sector, H object) -> (energies, vectors)` and the operator application from a
plug-in `cops(h_from, h_to, vec, ops)`, so the same driver checks the oracle (dense LAPACK, numpy) and the GPU library
(edigpu_lanczos_eigh_multi, edigpu_apply_op_flat)."""
from __future__ import annotations

import dataclasses

import numpy as np

from oracle import oracle as O


def dense_eigvec(sec, h):
    return np.linalg.eigh(h.dense())


def ground_manifold(om, eigvec=dense_eigvec, gs_threshold=1e-9):
    """-> (E0, [(sector, H object, vector)]) over the degenerate ground states of all sectors."""
    found = []
    for sec in O.sectors(om):
        h = O.hbuild(om, sec)
        if h.dim == 0:
            continue
        w, v = eigvec(sec, h)
        found.append((np.atleast_1d(w), v, sec, h))
    e0 = min(w[0] for w, _, _, _ in found)
    states = [(sec, h, v[:, k]) for w, v, sec, h in found for k in range(len(w)) if w[k] - e0 <= gs_threshold]
    return e0, states


def _family(om, **keep):
    """The model with every coupling switched off except `keep` (same sectors, same bases)."""
    def z(a):
        return None if a is None else np.zeros_like(a)
    base = dict(uloc=tuple(0.0 for _ in om.uloc), ust=0.0, jh=0.0, jx=0.0, jp=0.0, xmu=0.0, hfmode=False,
                hloc=z(om.hloc), be=z(om.be), bv=z(om.bv), bd=z(om.bd), bu=z(om.bu), hb=z(om.hb), vr=z(om.vr),
                vg=z(om.vg), pair_field=None)
    base.update(keep)
    return dataclasses.replace(om, **base)


def _expect(fam, states):
    s = 0.0
    cache = {}
    for sec, _, v in states:
        if sec not in cache:
            cache[sec] = O.hbuild(fam, sec).dense()
        s += float(np.real(np.vdot(v, cache[sec] @ v)))
    return s / len(states)


def _occupations(om, sec, h):
    ns = om.ns
    if om.ed_mode == "normal":
        iup, idw = np.arange(h.dim) % h.dimup, np.arange(h.dim) // h.dimup
        mu, md = h.mapup[iup], h.mapdw[idw]
    else:
        mu, md = h.map & ((1 << ns) - 1), h.map >> ns
    nu = np.array([(mu >> a) & 1 for a in range(om.norb)], dtype=float)
    nd = np.array([(md >> a) & 1 for a in range(om.norb)], dtype=float)
    return nu, nd


def doubles_energy_imp(om, states, e0):
    inter = dict(uloc=om.uloc, ust=om.ust, jh=om.jh, jx=om.jx, jp=om.jp)
    doubles = np.array([_expect(_family(om, ust=1.0, jh=1.0), states), _expect(_family(om, jh=-1.0), states),
                        _expect(_family(om, jx=1.0), states), _expect(_family(om, jp=1.0), states)])
    eint = _expect(_family(om, **inter), states)
    ehartree = _expect(_family(om, hfmode=om.hfmode, **inter), states) - eint
    eloc = _expect(_family(om, hloc=om.hloc), states)
    s2 = 0.0
    for sec, h, v in states:
        nu, nd = _occupations(om, sec, h)
        s2 += float(np.sum(np.abs(v) ** 2 * (0.5 * np.sum(nu - nd, axis=0)) ** 2))
    return doubles, np.array([eint + ehartree, eint, eloc, ehartree]), np.array([s2 / len(states), e0])


def _densities(om, states):
    up, dw = np.zeros(om.norb), np.zeros(om.norb)
    for sec, h, v in states:
        nu, nd = _occupations(om, sec, h)
        p = np.abs(v) ** 2
        up += nu @ p
        dw += nd @ p
    return up / len(states), dw / len(states)


def dens_docc(om, states):
    """dens.check, docc.check: <n_a>, <n_a,up n_a,dw> over the ground-state manifold."""
    dens, docc = np.zeros(om.norb), np.zeros(om.norb)
    for sec, h, v in states:
        nu, nd = _occupations(om, sec, h)
        p = np.abs(v) ** 2
        dens += (nu + nd) @ p
        docc += (nu * nd) @ p
    return dens / len(states), docc / len(states)


def phisc(om, states, cops, hsector):
    """Phi_ab (complex) in Fortran element order (a fastest).  cops(h_from, h_to, vec, ops) with ops = [(coef, create,
    iorb, ispin)]; hsector(sector) -> H object of a sector (cached by the caller).  The phisc.check files hold the signed
    real part (negative entries occur; this checkout's ED_OBSERVABLES_SUPERC.f90:459 stores the modulus, the files are
    older); Im Phi vanishes for the real models of the fixtures."""
    no, ns = om.norb, om.ns
    re, im = np.zeros((no, no)), np.zeros((no, no))
    for sec, h, v in states:
        if sec >= ns:
            continue
        h2 = hsector(sec + 1)
        for a in range(no):
            for b in range(no):
                veta = cops(h, h2, v, [(1.0, False, a, 1), (1.0, True, b, 0)])
                vkap = cops(h, h2, v, [(1.0, False, a, 1), (1.0j, True, b, 0)])
                re[a, b] += float(np.real(np.vdot(veta, veta))) / len(states)
                im[a, b] += float(np.real(np.vdot(vkap, vkap))) / len(states)
    dup, ddw = _densities(om, states)
    phi = np.zeros((no, no), complex)
    for a in range(no):
        for b in range(no):
            phi[a, b] = 0.5 * (re[a, b] - ddw[a] - (1.0 - dup[b])) + 0.5j * (im[a, b] - ddw[a] - (1.0 - dup[b]))
    return phi.flatten(order="F")


def magx(om, states, cops, hsector):
    no = om.norb
    m = np.zeros(no)
    for sec, h, v in states:
        if sec < 1:
            continue
        h2 = hsector(sec - 1)
        for a in range(no):
            vv = cops(h, h2, v, [(1.0, False, a, 0), (1.0, False, a, 1)])
            m[a] += float(np.real(np.vdot(vv, vv))) / len(states)
    dup, ddw = _densities(om, states)
    return m - dup - ddw


def _apply_c_normal(h_from, h_to, vec, iorb, ispin):
    """c_{iorb, ispin} on a normal-mode sector vector (apply_op_C, ED_SECTOR.f90:465-536) up to the sector-wide sign
    (-1)^Nup of a down operator, which drops out of the norms taken below."""
    out = np.zeros(h_to.dim, dtype=vec.dtype)
    mp_from, mp_to = (h_from.mapup, h_to.mapup) if ispin == 0 else (h_from.mapdw, h_to.mapdw)
    rank_to = {int(s): i for i, s in enumerate(mp_to)}
    bit = 1 << iorb
    sel = np.nonzero((mp_from & bit) != 0)[0]
    if sel.size == 0:
        return out
    below = mp_from[sel] & (bit - 1)
    par = np.zeros(sel.size, dtype=np.int64)
    for k in range(iorb):
        par += (below >> k) & 1
    sgn = 1.0 - 2.0 * (par & 1)
    tgt = np.array([rank_to[int(s) ^ bit] for s in mp_from[sel]], dtype=np.int64)
    v2, o2 = vec.reshape(h_from.dimdw, h_from.dimup), out.reshape(h_to.dimdw, h_to.dimup)
    if ispin == 0:
        o2[:, tgt] = v2[:, sel] * sgn[None, :]
    else:
        o2[tgt, :] = v2[sel, :] * sgn[:, None]
    return out


def numpy_cops_normal(h_from, h_to, vec, ops):
    """sum_s coef_s c_{orb_s, spin_s} |vec> (annihilators of one spin species), ops = [(coef, iorb, ispin)]."""
    out = np.zeros(h_to.dim, dtype=vec.dtype)
    for coef, iorb, ispin in ops:
        out = out + coef * _apply_c_normal(h_from, h_to, vec, iorb, ispin)
    return out


def exciton_normal(om, states, cops=numpy_cops_normal, hsector=None):
    """[exct_S0(1,2), exct_Tz(1,2)] of ED_OBSERVABLES_NORMAL.f90:228-296 (exciton.check of REPLICA_ / GENERAL_NORMAL):
    theta_ss = ||(c_{1s} + c_{2s}) gs||^2 through apply_Cops into the sector with one electron of spin s less;
    S0 = (theta_up + theta_dw - n_1 - n_2) / 2, Tz = (theta_up - theta_dw - m_1 - m_2) / 2."""
    cache = {}

    def default_hsector(sec):
        if sec not in cache:
            cache[sec] = O.HNormal(om, *sec)
        return cache[sec]

    hsector = hsector or default_hsector
    th = np.zeros(2)
    for sec, h, v in states:
        for ispin in range(2):
            sec2 = (sec[0] - 1, sec[1]) if ispin == 0 else (sec[0], sec[1] - 1)
            if min(sec2) < 0:
                continue
            vv = cops(h, hsector(sec2), v, [(1.0, 0, ispin), (1.0, 1, ispin)])
            th[ispin] += float(np.real(np.vdot(vv, vv))) / len(states)
    dup, ddw = _densities(om, states)
    dens, magz = dup + ddw, dup - ddw
    return np.array([0.5 * (th[0] + th[1] - dens[0] - dens[1]), 0.5 * (th[0] - th[1] - magz[0] - magz[1])])


def exciton_nonsu2(om, states, cops, hsector):
    """[exct_S0, exct_Tx, exct_Ty, exct_Tz](1,2) of ED_OBSERVABLES_NONSU2.f90:325-425 (exciton.check of REPLICA_ /
    GENERAL_NONSU2; no factor 1/2 in this mode): norms of (c_{1s} + c_{2s'}) gs and (c_{1s} - i c_{2s'}) gs in the sector
    N - 1.  cops(h_from, h_to, vec, ops) with ops = [(coef, create, iorb, ispin)]."""
    th = {}
    keys = {"upup": (0, 0, 1.0), "dwdw": (1, 1, 1.0), "updw": (0, 1, 1.0), "dwup": (1, 0, 1.0),
            "o_updw": (0, 1, -1.0j), "o_dwup": (1, 0, -1.0j)}
    for k in keys:
        th[k] = 0.0
    for sec, h, v in states:
        if sec < 1:
            continue
        h2 = hsector(sec - 1)
        for k, (s, t, c2) in keys.items():
            vv = cops(h, h2, v, [(1.0, False, 0, s), (c2, False, 1, t)])
            th[k] += float(np.real(np.vdot(vv, vv))) / len(states)
    dup, ddw = _densities(om, states)
    dens, magz = dup + ddw, dup - ddw
    return np.array([th["upup"] + th["dwdw"] - dens[0] - dens[1],
                     th["updw"] + th["dwup"] - dens[0] - dens[1],
                     th["o_updw"] - th["o_dwup"] - magz[0] + magz[1],
                     th["upup"] - th["dwdw"] - magz[0] - magz[1]])
