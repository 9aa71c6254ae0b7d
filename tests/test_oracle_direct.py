"""The oracle's on-the-fly nonsu2 product (oracle/edipack_oracle_flat.inc orc_directmatvec_nonsu2_main, restating
directMatVec_nonsu2_main, ED_NONSU2/ED_HAMILTONIAN_NONSU2_DIRECT_HxV.f90:22-126 with direct/HxV{imp,int,bath,imp_bath}.f90)
against the oracle's stored product of the same sector (spMatVec_nonsu2_main on the arrays of
ED_NONSU2/stored/*.f90, pinned by the golden energies of tests/test_oracle_golden.py).  The on-the-fly form is what
bench.py times as the CPU baseline of the on-the-fly workload (config 5)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.common import make_models
from tests.test_host_builders import SUNDRY2, SUNDRY3

CASES = [
    ("hybrid", 1, 3, 3, {}),
    ("hybrid", 2, 2, 4, {}),
    ("hybrid", 3, 1, 5, {}),                                     # config 5's structure one step down
    ("hybrid", 3, 1, 0, {}),                                     # the empty sector: one state
    ("normal", 2, 2, 5, {}),
    ("replica", 2, 2, 4, {}),
    ("general", 2, 1, 3, {}),
    ("hybrid", 2, 2, 4, dict(sundry=SUNDRY2)),
    ("hybrid", 3, 1, 4, dict(sundry=SUNDRY3)),
    ("hybrid", 2, 2, 3, dict(exc_field=np.array([0.12, 0.5, 0.3, 0.07]))),
    ("normal", 2, 1, 4, dict(spin_field=np.array([[0.3, 0.1, 0.2], [0.0, 0.2, -0.15]]))),
]


@pytest.mark.parametrize("bath,norb,nbath,ntot,extra", CASES)
def test_on_the_fly_product_equals_stored_product(bath, norb, nbath, ntot, extra):
    om, _ = make_models("nonsu2", bath, norb, nbath, seed=11, **extra)
    h = O.HFlat(om, ntot)
    rng = np.random.default_rng(5)
    v = rng.standard_normal(h.dim) + 1j * rng.standard_normal(h.dim)
    want = h.matvec(v)
    got = O.direct_matvec_nonsu2(om, ntot, v)
    assert np.abs(got - want).max() <= 1e-13 * max(1.0, np.abs(want).max())
    # the threaded row ranges of the timed form give the same vector
    d = O.DirectNonsu2(om, ntot)
    assert d.dim == h.dim and np.array_equal(d.map[:d.dim], h.map)
    hv = np.full(h.dim, np.nan + 0j)
    d.matvec(v, hv, threads=3)
    assert np.array_equal(hv, got)
    h.close()


def test_on_the_fly_product_is_hermitian():
    om, _ = make_models("nonsu2", "hybrid", 3, 1, seed=12)
    d = O.DirectNonsu2(om, 4)
    rng = np.random.default_rng(6)
    x = rng.standard_normal(d.dim) + 1j * rng.standard_normal(d.dim)
    y = rng.standard_normal(d.dim) + 1j * rng.standard_normal(d.dim)
    hx, hy = np.empty_like(x), np.empty_like(y)
    d.matvec(x, hx)
    d.matvec(y, hy)
    assert abs(np.vdot(y, hx) - np.vdot(hy, x)) < 1e-11
