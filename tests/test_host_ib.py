"""The impurity-block image (csrc/host_ib.cpp) and the per-block routines the gfx950 kernels run (csrc/ib_core.hpp),
evaluated on the CPU by tests/host_ib.cpp: H*v through the image == H*v through the explicit arrays of the same
sector (spMatVec_normal_main's terms, ED_NORMAL/ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:517-650).  The GPU tests check
the kernels; this one pins the tables and the index / sign logic they share, without a GPU."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from tests.common import make_models
from tests.test_host_builders import SUNDRY2, SUNDRY3

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    so = str(tmp_path_factory.mktemp("host_ib") / "host_ib.so")
    csrc = os.path.join(ROOT, "edipack_amd", "csrc")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"),
                           "-I", csrc, "-o", so, os.path.join(ROOT, "tests", "host_ib.cpp"),
                           os.path.join(csrc, "host_ib.cpp"), os.path.join(csrc, "host_build.cpp")])
    lib = C.CDLL(so)
    lib.host_ib_error.restype = C.c_char_p
    return lib


def _check(lib, pm, nup, ndw, max_rows):
    info = (C.c_int32 * 7)()
    diff = C.c_double(-1.0)
    m = pm.to_c()
    rc = lib.host_ib_check(C.byref(m), nup, ndw, max_rows, info, C.byref(diff))
    return rc, list(info), diff.value, lib.host_ib_error().decode()


CASES = [
    # bath, norb, nbath, sector, chunk rows, extra
    ("normal", 1, 5, (3, 3), 8, {}),
    ("normal", 1, 7, (4, 3), 480, {}),
    ("normal", 2, 3, (4, 4), 16, {}),                      # Jx = Jp != 0: Hnd terms, padded panels
    ("normal", 2, 3, (3, 5), 480, dict(jxp=0.0)),          # no Hnd: unpadded columns
    ("normal", 2, 4, (5, 4), 40, {}),
    ("hybrid", 2, 5, (3, 4), 12, {}),
    ("hybrid", 3, 4, (3, 4), 10, {}),
    ("hybrid", 3, 5, (4, 4), 24, {}),
    ("hybrid", 3, 5, (1, 7), 480, {}),                     # classes missing on both sides
    ("hybrid", 3, 6, (5, 4), 30, dict(jxp=0.0)),
    ("hybrid", 3, 3, (3, 3), 480, dict(sundry=SUNDRY3)),
    ("normal", 2, 2, (3, 2), 480, dict(sundry=SUNDRY2)),
    ("hybrid", 3, 3, (2, 3), 6, dict(exc_field=np.array([0.12, 0.5, 0.5, 0.07]))),
    ("normal", 2, 3, (4, 3), 480, dict(spin_field=np.array([[0.3, 0.1, 0.2], [0.0, 0.0, -0.15]]))),
]


@pytest.mark.parametrize("bath,norb,nbath,sec,rows,extra", CASES)
def test_image_product_matches_explicit_arrays(shim, bath, norb, nbath, sec, rows, extra):
    _, pm = make_models("normal", bath, norb, nbath, seed=31, **extra)
    rc, info, diff, msg = _check(shim, pm, sec[0], sec[1], rows)
    assert rc == 0, msg
    assert info[0] == 1 and diff < 1e-13, (info, diff)
    if (norb == 1 or extra.get("jxp", 0.25) == 0.0) and "sundry" not in extra:
        assert info[5] == 0 and info[6] < 16          # no Hnd: only the last panel is padded
    else:
        assert info[5] > 0


@pytest.mark.parametrize("bath,norb,nbath,sec,rows,extra", CASES)
def test_split_row_image_matches_explicit_arrays(shim, bath, norb, nbath, sec, rows, extra):
    """Rows longer than the LDS: the image of half a row at a time (one value of the top bath bit), the hop over the top
    level from the vector.  Forced here on small sectors (budget < 0)."""
    _, pm = make_models("normal", bath, norb, nbath, seed=37, **extra)
    info = (C.c_int32 * 8)()
    diff = C.c_double(-1.0)
    m = pm.to_c()
    rc = shim.host_ib_check2(C.byref(m), sec[0], sec[1], rows, 10 ** 8, info, C.byref(diff))
    assert rc == 0 and info[7] == 1                           # the budget is not exceeded: one image
    rc = shim.host_ib_check2(C.byref(m), sec[0], sec[1], rows, -1, info, C.byref(diff))
    msg = shim.host_ib_error().decode()
    if rc == 1:
        assert "half of the split row is empty" in msg, msg    # every block has the same top bit: refused, generic kernels
        assert sec in ((1, 7),)
        return
    assert rc == 0, msg
    assert info[0] == 1 and info[7] == 2 and diff.value < 1e-13, (list(info), diff.value)


PAIR_CASES = [
    # replica / general baths: hops between the levels of one replica (stored/H_up.f90:26-50)
    ("replica", 2, 2, (3, 3), 480),
    ("replica", 2, 3, (4, 3), 12),          # chunks of 12 rows: bath-bath hops that leave the chunk
    ("replica", 3, 2, (4, 5), 24),
    ("replica", 3, 3, (6, 5), 40),
    ("general", 2, 3, (3, 5), 16),
    ("general", 3, 2, (5, 4), 480),
    ("replica", 1, 4, (2, 3), 8),           # one orbital: a replica is one level, no bath-bath hop
]


@pytest.mark.parametrize("bath,norb,nbath,sec,rows", PAIR_CASES)
def test_bath_bath_hops_of_replica_baths(shim, bath, norb, nbath, sec, rows):
    _, pm = make_models("normal", bath, norb, nbath, seed=33)
    rc, info, diff, msg = _check(shim, pm, sec[0], sec[1], rows)
    assert rc == 0, msg
    assert info[0] == 1 and diff < 1e-13, (info, diff)
    # rows staged in halves: refused with bath-bath hops (the generic kernels take the sector)
    info2 = (C.c_int32 * 8)()
    d2 = C.c_double(-1.0)
    m = pm.to_c()
    rc = shim.host_ib_check2(C.byref(m), sec[0], sec[1], rows, -1, info2, C.byref(d2))
    if norb > 1:
        assert rc == 1 and "bath-bath hops in a row staged in halves" in shim.host_ib_error().decode()


def test_chunks_follow_the_row_budget(shim):
    _, pm = make_models("normal", "hybrid", 3, 6, seed=5)
    few = _check(shim, pm, 4, 4, 480)[1]
    many = _check(shim, pm, 4, 4, 12)[1]
    assert few[2] < many[2] and many[3] <= 12 and few[1] > many[1]


def test_refusals(shim):
    # more than three orbitals
    _, pm = make_models("normal", "normal", 4, 1, seed=3)
    rc, _, _, msg = _check(shim, pm, 4, 4, 480)
    assert rc == 1 and "1..3" in msg
