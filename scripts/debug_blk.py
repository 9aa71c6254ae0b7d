"""Step-by-step run of the panel-major Lanczos loop on one small sector (debug aid): prints after every stage."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["EDIGPU_BLOCKED"] = "1"
os.environ["EDIGPU_BLOCKED_MIN"] = "0"
os.environ["EDIGPU_BLOCKED_W"] = sys.argv[1] if len(sys.argv) > 1 else "16"
import torch  # noqa
from edipack_amd import capi
from edipack_amd.hamiltonian import SectorHamiltonian
from tests.common import make_models
def say(*a):
    print(*a, flush=True)
capi.init(0)
bath, norb, nbath, sec, jxp = "normal", 2, 4, (5, 5), 0.0
if len(sys.argv) > 2:
    bath, norb, nbath, sec, jxp = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), (int(sys.argv[5]), int(sys.argv[6])), float(sys.argv[7])
om, pm = make_models("normal", bath, norb, nbath, seed=71, jxp=jxp)
hb = SectorHamiltonian.normal_from_model(pm, *sec)
say("built", hb.dim_up, hb.dim_dw, hb.image_info())
v = np.random.default_rng(3).standard_normal(hb.dim)
a, b, n = hb.lanczos_tridiag(v, 5)
say("tridiag 5", a[:3], b[:3])
a, b, n = hb.lanczos_tridiag(v, 40)
say("tridiag 40", n)
e, x, nd = hb.lanczos_eigh(nitermax=20, tol=1e-13, v0=v)
say("eigh 20", e, nd)
e, x, nd = hb.lanczos_eigh(nitermax=min(300, hb.dim), tol=1e-13, v0=v)
say("eigh 300", e, nd)
say("resid", np.linalg.norm(hb.apply(x) - e * x))
say("bench", hb.lanczos_bench(2, 3))
