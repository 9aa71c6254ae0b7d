#!/usr/bin/env python3
"""Zero-temperature impurity Green's function of a two-orbital Anderson model, device-resident.

What EDIpack does between `ed_solve` and `ed_get_gimp` for ed_mode=normal (ED_NORMAL/ED_DIAG_NORMAL.f90,
ED_NORMAL/ED_GF_NORMAL.f90:131-177, 363-427), with the hot path on the GPU:

  1. every (Nup,Ndw) sector is built on the device (edigpu_normal_build: nothing O(Dim) on the host) and its two
     lowest states are found by thick-restart Lanczos (edigpu_lanczos_eigh_multi); the eigenvector stays in HBM
  2. c^+ / c is applied device-to-device (edigpu_apply_op_normal)
  3. the neighbouring sector is tridiagonalised from that seed (edigpu_lanczos_tridiag_dev); only the
     alpha/beta coefficients and the seed norm reach the host
  4. poles and weights from the small tridiagonal matrix give G(i w)

Run on a machine with an MI355X:   python examples/impurity_gf_on_device.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from edipack_amd import capi
    from edipack_amd.hamiltonian import ImpurityModel, SectorHamiltonian

    capi.init(0)
    norb, nbath = 2, 3
    rng = np.random.default_rng(1)
    hloc = np.zeros((1, 1, norb, norb), complex)
    hloc[0, 0] = np.diag([0.25, -0.25])
    model = ImpurityModel(ed_mode="normal", bath_type="normal", norb=norb, nbath=nbath, nspin=1, hfmode=True, xmu=0.0,
                          uloc=np.array([2.0, 2.0]), ust=1.5, jh=0.25, jx=0.25, jp=0.25, hloc=hloc,
                          be=rng.uniform(-2, 2, (1, norb, nbath)), bv=rng.uniform(0.2, 0.6, (1, norb, nbath)))
    ns = model.ns

    # 1. ground-state manifold
    found = []
    for nup in range(ns + 1):
        for ndw in range(ns + 1):
            h = SectorHamiltonian.normal_from_model(model, nup, ndw)
            ev, vec, _, _ = h.lanczos_eigh_multi(min(2, h.dim), tol=1e-12)
            found += [(ev[k], (nup, ndw), vec[k].copy()) for k in range(len(ev))]
            h.destroy()
    e0 = min(f[0] for f in found)
    states = [f for f in found if f[0] - e0 < 1e-9]
    print(f"ground-state energy {e0:.10f}, degeneracy {len(states)}, sector(s) {[s[1] for s in states]}")

    # 2.-4. G_aa(i w_n) on the first Matsubara frequencies
    beta, nw = 50.0, 8
    z = 1j * np.pi / beta * (2 * np.arange(nw) + 1)
    for a in range(norb):
        g = np.zeros(nw, complex)
        for ei, (nup, ndw), vec in states:
            src = SectorHamiltonian.normal_from_model(model, nup, ndw)
            vd = torch.from_numpy(vec).cuda()
            for create, sign in ((True, 1), (False, -1)):
                n2 = nup + (1 if create else -1)
                if not 0 <= n2 <= ns:
                    continue
                dst = SectorHamiltonian.normal_from_model(model, n2, ndw)
                seed = torch.empty(dst.dim, dtype=torch.float64, device="cuda")
                src.apply_op_to(dst, vd.data_ptr(), seed.data_ptr(), a, 0, create)
                nl = min(dst.dim, 200)
                al, bl, _, norm2 = dst.lanczos_tridiag_dev(seed.data_ptr(), nl)
                dst.destroy()
                if norm2 == 0.0:
                    continue
                t = np.diag(al[:nl]) + np.diag(bl[1:nl], 1) + np.diag(bl[1:nl], -1)
                ev, y = np.linalg.eigh(t)
                g += np.sum((norm2 / len(states) * y[0] ** 2)[None, :] / (z[:, None] - sign * (ev - ei)[None, :]), axis=1)
            src.destroy()
        print(f"orbital {a}: G(i w_0..2) =", np.array2string(g[:3], precision=6))


if __name__ == "__main__":
    main()
