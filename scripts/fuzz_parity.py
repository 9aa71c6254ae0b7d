#!/usr/bin/env python3
"""Long randomized parity sweep (not part of the test suite: `python scripts/fuzz_parity.py --trials 400`): random
models / sectors of every mode and bath type, with and without phonons (random coupling matrices), complex normal
mode, the two products of the transposed exchange with the all-to-all emulated, against the CPU oracle.  Run it with
EDIGPU_PANEL_VEC2_MIN=1 to force the two-column panel kernel on the small sectors as well."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=300)
    ap.add_argument("--seed", type=int, default=777)
    args = ap.parse_args()
    import numpy as np
    import torch
    from edipack_amd import capi
    from edipack_amd.hamiltonian import SectorHamiltonian
    from edipack_amd.sharding import ShardPlan
    from oracle import oracle as O
    from tests.common import make_models

    capi.init(0)
    L = capi.lib()
    rng = np.random.default_rng(args.seed)

    def rel(a, b):
        return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))

    worst, ncase, t0 = 0.0, 0, time.time()
    for trial in range(args.trials):
        mode = ["normal", "superc", "nonsu2"][int(rng.integers(0, 3))]
        bath = ["normal", "hybrid", "replica", "general"][int(rng.integers(0, 4))]
        norb, nbath = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        if mode != "normal" and (norb + (nbath if bath == "hybrid" else nbath * norb)) > 7:
            nbath = 1
        if bath in ("replica", "general") and norb == 1 and mode == "nonsu2":
            bath = "normal"
        om, pm = make_models(mode, bath, norb, nbath, seed=5000 + trial, jxp=float(rng.choice([0.0, 0.25])))
        ns = om.ns
        nph = int(rng.choice([0, 0, 1, 2]))
        cmplx = mode == "normal" and nph == 0 and rng.random() < 0.25
        if nph:
            g = rng.uniform(-0.3, 0.3, (norb, norb))
            g = 0.5 * (g + g.T) if rng.random() < 0.5 else np.diag(np.diag(g))
            aph = float(rng.choice([0.0, 0.2]))
            for m in (om, pm):
                m.nph, m.w0_ph, m.a_ph, m.g_ph = nph, 0.6, aph, g
        if cmplx:
            t = rng.uniform(-0.4, 0.4, (norb, norb))
            t = t - t.T
            for m in (om, pm):
                hl = np.asarray(m.hloc, complex).copy()
                hl[0, 0] = hl[0, 0] + 1j * t
                m.hloc = hl
        # the terms only some inputs switch on: random coulomb_sundry lines with their Hermitian conjugates (normal and
        # superc: spin-conserving lines; nonsu2: any), spin_field z and exc_field (1), (4) in normal mode (real algebra),
        # all components in nonsu2, none in superc
        extra = (mode != "normal" or (nph == 0 and not cmplx)) and rng.random() < 0.4
        if extra:
            lines = []
            for _ in range(int(rng.integers(1, 4))):
                while True:
                    ops = [(int(rng.integers(0, norb)), int(rng.integers(0, 2))) for _ in range(4)]
                    bal = [0, 0]
                    for k_, (_, sp) in enumerate(ops):
                        bal[sp] += 1 if k_ < 2 else -1
                    if bal == [0, 0] or mode == "nonsu2":
                        break
                u = float(rng.uniform(-0.5, 0.5))
                lines.append((u, ops[0], ops[1], ops[2], ops[3]))
                # Hermitian conjugate of cd_i cd_j c_k c_l applied as c_l, cd_j, c_k, cd_i (per-spin-word signs): the
                # reversed string c^+_l c^+_k ... is again of the form (cd_l' cd_k' ...) with (i, j, k, l) -> (l, k, j, i)
                lines.append((u, ops[3], ops[2], ops[1], ops[0]))
            sf = np.zeros((norb, 3))
            sf[:, 2] = rng.uniform(-0.3, 0.3, norb)
            ef = np.array([rng.uniform(-0.2, 0.2), 0.0, 0.0, rng.uniform(-0.2, 0.2)])
            if mode == "nonsu2":
                sf[:, :2] = rng.uniform(-0.3, 0.3, (norb, 2))
                ef[1:3] = rng.uniform(-0.2, 0.2, 2)
            elif mode == "superc":
                sf[:], ef[:] = 0.0, 0.0
            for m in (om, pm):
                m.sundry, m.spin_field, m.exc_field = lines, sf, ef
        tag = (trial, mode, bath, norb, nbath, nph, cmplx, extra)
        if mode == "normal":
            sec = (int(rng.integers(0, ns + 1)), int(rng.integers(0, ns + 1)))
            if cmplx:
                ho, hs = O.HNormalCmplx(om, *sec), [SectorHamiltonian.normal_cmplx_from_model(pm, *sec)]
                v = rng.standard_normal(ho.dim) + 1j * rng.standard_normal(ho.dim)
            else:
                ho, hs = O.HNormal(om, *sec), [SectorHamiltonian.normal_from_model(pm, *sec)]
                v = rng.standard_normal(ho.dim)
                if nph == 0 and 0 < ho.dim <= 60000:
                    # the hand-over boundary on the oracle's (= the reference's) arrays: factored when they allow it
                    hs.append(SectorHamiltonian.normal_from_arrays(ho.dimup, ho.dimdw, ho.hd, ho.up, ho.dw,
                                                                   ho.nd if ho.has_nd else None))
        else:
            sec = int(rng.integers(-ns, ns + 1)) if mode == "superc" else int(rng.integers(0, 2 * ns + 1))
            ho = O.HFlat(om, sec)
            if ho.dim == 0 or ho.dim > 40000:
                continue
            hs = [SectorHamiltonian.flat_from_model(pm, sec), SectorHamiltonian.direct_from_model(pm, sec)]
            v = rng.standard_normal(ho.dim) + 1j * rng.standard_normal(ho.dim)
        if ho.dim > 60000:
            for h in hs:
                h.destroy()
            continue
        ref = ho.matvec(v)
        # error relative to max(|H v|, |v|): a 1 x 1 sector whose diagonal element is a cancellation of O(10) terms (seen:
        # -1.8e-4) would otherwise turn one ulp of those terms into 2e-12 "relative"
        den = max(float(np.max(np.abs(ref))) if ref.size else 0.0, float(np.max(np.abs(v))) if v.size else 0.0, 1e-300)
        for h in hs:
            e = float(np.max(np.abs(h.apply(v) - ref)) / den) if ref.size else 0.0
            worst = max(worst, e)
            if not e < 1e-12:
                print("MISMATCH", tag, sec, e, "kind", h.kind, "\n got", h.apply(v)[:6], "\n ref", ref[:6],
                      "\n g_ph", getattr(pm, "g_ph", None), "a_ph", pm.a_ph, om.a_ph, "w0", pm.w0_ph, "nph", pm.nph, om.nph,
                      "\n v", v[:4], "\n dense", ho.dense() if ho.dim <= 4 else None, flush=True)
            assert e < 1e-12, (tag, sec, e)
            if ho.dim >= 8:
                nl = min(ho.dim, 8)
                ao, bo, _ = ho.lanc_tridiag(v, nl)
                ag, bg, _ = h.lanczos_tridiag(v, nl)
                # compare up to the first (near) breakdown of the recurrence: tiny sectors with symmetric baths have
                # Krylov spaces of a few dimensions, and past beta ~ 0 both sides amplify rounding
                nc = 4
                for kk in range(1, 4):
                    if abs(bo[kk]) < 1e-6 * max(1.0, float(np.max(np.abs(ao[:4])))):
                        nc = kk
                        break
                okc = rel(ag[:nc], ao[:nc]) < 1e-8 and (nc < 2 or rel(bg[:nc], bo[:nc]) < 1e-8)
                if not okc:
                    print("TRIDIAG MISMATCH", tag, sec, "\n gpu a", ag[:5], "b", bg[:5], "\n ref a", ao[:5], "b", bo[:5],
                          flush=True)
                assert okc, (tag, sec)
        # transposed exchange, emulated, on the plain real normal sectors
        if mode == "normal" and not cmplx and nph == 0 and ho.dim > 0:
            h = hs[0]
            world = int(rng.integers(1, 6))
            du, dd = h.dim_up, h.dim_dw
            try:
                halo = h.transpose_halo()
            except capi.EdigpuError:       # more than 16 Hnd terms: explicit image, served by the all-gather form
                for hh in hs:
                    hh.destroy()
                ncase += 1
                continue
            plans = [ShardPlan(units=dd, unit_len=du, world=world, rank=r) for r in range(world)]
            q, pcol = plans[0].q, -(-du // world)
            pw, st = pcol + 2 * halo, torch.cuda.current_stream().cuda_stream
            n = world * q * pw
            tmp, send = [], []
            for pl in plans:
                x = torch.zeros(max(pl.chunk, 1), dtype=torch.float64, device="cuda")
                x[:pl.nloc] = torch.from_numpy(v[pl.row_first:pl.row_first + pl.nloc]).cuda()
                sb = torch.zeros(max(n, 1), dtype=torch.float64, device="cuda")
                capi.check(L.edigpu_transpose_pack(du, pl.count, q, world, pcol, halo, x.data_ptr(), sb.data_ptr(), st))
                t_ = torch.zeros(max(pl.chunk, 1), dtype=torch.float64, device="cuda")
                h.apply_rows_dev(pl.first, pl.count, x.data_ptr(), t_.data_ptr(), st)
                tmp.append(t_), send.append(sb[:n].view(world, q * pw))
            hvc = []
            for c in range(world):
                recv = torch.cat([send[r][c] for r in range(world)]).contiguous()
                out = torch.zeros(max(n, 1), dtype=torch.float64, device="cuda")
                cf = min(c * pcol, du)
                h.apply_cols_dev(cf, max(0, min(pcol, du - cf)), pw, halo, recv.data_ptr(), out.data_ptr(), st)
                hvc.append(out[:n].view(world, q * pw))
            res = []
            for r, pl in enumerate(plans):
                back = torch.cat([hvc[c][r] for c in range(world)]).contiguous()
                capi.check(L.edigpu_transpose_unpack_add(du, pl.count, q, world, pcol, halo, back.data_ptr(),
                                                         tmp[r].data_ptr(), st))
                res.append(tmp[r][:pl.nloc].cpu().numpy())
            e = float(np.max(np.abs(np.concatenate(res) - ref)) / den)
            worst = max(worst, e)
            assert e < 1e-12, (tag, sec, "transposed", world, e)
        for h in hs:
            h.destroy()
        ncase += 1
        if trial % 50 == 49:
            print(f"trial {trial + 1}: {ncase} cases, worst rel err {worst:.2e}, {time.time() - t0:.0f} s", flush=True)
    print(f"fuzz ok: {ncase} cases, worst rel err {worst:.2e}")


if __name__ == "__main__":
    main()
