#!/usr/bin/env python3
"""Debug probe: H*v of every sector of a golden directory against the oracle, plus the symmetry defect of the GPU image."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "HYBRID_NORMAL"
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
    rng = np.random.default_rng(1)
    for sec in O.sectors(om):
        h = O.hbuild(om, sec)
        if h.dim == 0:
            continue
        hg = SectorHamiltonian.normal_from_model(pm, *sec) if om.ed_mode == "normal" else SectorHamiltonian.flat_from_model(pm, sec)
        v = rng.standard_normal(h.dim).astype(hg.dtype)
        e = np.max(np.abs(hg.apply(v) - h.matvec(v)))
        print(sec, "dim", h.dim, "err", e, "BAD" if e > 1e-12 else "", flush=True)
        hg.destroy()


if __name__ == "__main__":
    main()
