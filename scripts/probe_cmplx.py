#!/usr/bin/env python3
"""_CMPLX_NORMAL product on the config-2 structure: the doubled real sector against the four-product composite and
against one real product of the same sector.  python scripts/probe_cmplx.py [fourproducts]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "fourproducts":
    os.environ["EDIGPU_CMPLX_FOURPRODUCTS"] = "1"
import numpy as np
import torch  # noqa
from edipack_amd import capi
from edipack_amd.hamiltonian import SectorHamiltonian
from edipack_amd.synthetic import WORKLOADS, synthetic_model

capi.init(0)
w = WORKLOADS[os.environ.get("WL", "cfg2")]
pm = synthetic_model(w)
hr = SectorHamiltonian.normal_from_model(pm, *w.sector)
t_real = hr.time_apply(3, 20, lanczos=0)
hr.destroy()
rng = np.random.default_rng(1)
t = rng.uniform(-0.3, 0.3, (w.norb, w.norb))
hl = np.asarray(pm.hloc, complex).copy()
hl[0, 0] = hl[0, 0] + 1j * (t - t.T)
pm.hloc = hl
hz = SectorHamiltonian.normal_cmplx_from_model(pm, *w.sector)
t_z = hz.time_apply(3, 20, lanczos=0)
t_l = hz.time_apply(3, 20, lanczos=1)
print(f"{w.name}: real product {t_real:.4f} ms, complex product {t_z:.4f} ms = {t_z / t_real:.2f}x, complex Lanczos step {t_l:.4f} ms"
      f" ({'four products' if os.environ.get('EDIGPU_CMPLX_FOURPRODUCTS') else 'doubled real sector'})")
hz.destroy()
