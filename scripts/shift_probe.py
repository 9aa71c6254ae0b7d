import numpy as np, sys, os
sys.path.insert(0,'.')
from oracle import oracle as O
from edipack_amd.hamiltonian import SectorHamiltonian
from edipack_amd import capi
from tests.common import make_models
from tests.test_gpu_baseline_configs import _cf
capi.init(0)
om, pm = make_models("normal", "normal", 2, 3, seed=11)
e0 = 100.0
hl = np.array(om.hloc, complex)
for a in range(2): hl[0, 0, a, a] += e0
for m in (om, pm): m.hloc, m.be = hl, np.asarray(m.be) + e0
ho = O.HNormal(om, 4, 4)
v = np.random.default_rng(3).standard_normal(ho.dim)
a_ref, b_ref, _ = ho.lanc_tridiag(v, 40)
res={}
for name, env in (("fused",{}),("exact",{"EDIGPU_LANCZOS_EXACTBETA":"1"}),("unfused",{"EDIGPU_LANCZOS_UNFUSED":"1"})):
    for k,val in env.items(): os.environ[k]=val
    hg = SectorHamiltonian.normal_from_model(pm, 4, 4)
    a,b,_ = hg.lanczos_tridiag(v, 40)
    hg.destroy()
    for k in env: del os.environ[k]
    res[name]=(a,b)
th = np.linalg.eigvalsh(np.diag(a_ref) + np.diag(b_ref[1:], 1) + np.diag(b_ref[1:], -1))
z=th[0]-0.5
print("ref cf", _cf(a_ref,b_ref,z))
for name,(a,b) in res.items():
    print(name, "cf", _cf(a,b,z), "da", np.abs(a-a_ref)[:12], "db", np.abs(b-b_ref)[:12])
# exact tridiagonalisation in higher precision? compare with dense resolvent
d = ho.dense(); vn=v/np.linalg.norm(v)
print("dense resolvent", vn @ np.linalg.solve(z*np.eye(ho.dim)-d, vn))
