#!/usr/bin/env python3
"""Debug probe: edigpu_lanczos_eigh_multi on every sector of a golden directory against dense LAPACK."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "HYBRID_NORMAL"
    neigen = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    import torch  # noqa: F401
    from edipack_amd import capi
    from edipack_amd.hamiltonian import SectorHamiltonian
    from oracle import oracle as O
    from tests.common import replica_golden_models
    from tests.test_oracle_golden import GOLD, REPLICA_DIRS, _from_dir, golden_models
    capi.init(0)
    g = GOLD[name]
    if name in REPLICA_DIRS:
        om, pm = replica_golden_models(g["input"])
    else:
        inp, par = _from_dir(name)
        pm_par = {k: v for k, v in par.items() if k not in ("ed_hw_bath", "deltasc")}
        om, pm = golden_models(inp["ED_MODE"], inp["BATH_TYPE"], int(inp["NORB"]), int(inp["NBATH"]), pm_par)
    O.to_struct(om)
    only = eval(sys.argv[3]) if len(sys.argv) > 3 else None
    for sec in O.sectors(om):
        if only is not None and sec != only:
            continue
        h = O.hbuild(om, sec)
        if h.dim <= 8:
            continue
        wd = np.linalg.eigvalsh(h.dense())
        hg = SectorHamiltonian.normal_from_model(pm, *sec) if om.ed_mode == "normal" else SectorHamiltonian.flat_from_model(pm, sec)
        w, v, nconv, nmv = hg.lanczos_eigh_multi(min(neigen, hg.dim), tol=1e-13)
        bad = abs(w[0] - wd[0]) > 1e-9
        print(sec, "dim", h.dim, "nconv", nconv, "nmv", nmv, "gpu", w, "dense", wd[:neigen], "BAD" if bad else "", flush=True)
        hg.destroy()


if __name__ == "__main__":
    main()
