"""The local-block tables (csrc/host_sb.cpp) and the per-block routines the gfx950 kernels of round 4 run
(csrc/sb_core.hpp: blocks that hold the nb0 lowest bath levels besides the impurity levels), evaluated on the CPU by
tests/host_sb.cpp: H*v through the tables == H*v through the explicit arrays of the same sector (spMatVec_normal_main's
terms, ED_NORMAL/ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:517-650).  The GPU tests check the kernels; this one pins the
tables, the wave-slot plans and the index / sign logic they share, without a GPU."""
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
    so = str(tmp_path_factory.mktemp("host_sb") / "host_sb.so")
    csrc = os.path.join(ROOT, "edipack_amd", "csrc")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"),
                           "-I", csrc, "-o", so, os.path.join(ROOT, "tests", "host_sb.cpp"),
                           os.path.join(csrc, "host_sb.cpp"), os.path.join(csrc, "host_ib.cpp"),
                           os.path.join(csrc, "host_build.cpp")])
    lib = C.CDLL(so)
    lib.host_sb_error.restype = C.c_char_p
    return lib


def _check(lib, pm, nup, ndw, nb0, max_rows, nt=128, nbt=8, nw=2, gs=8):
    info = (C.c_int32 * 8)()
    diff = C.c_double(-1.0)
    m = pm.to_c()
    rc = lib.host_sb_check(C.byref(m), nup, ndw, nb0, max_rows, nt, nbt, nw, gs, info, C.byref(diff))
    return rc, list(info), diff.value, lib.host_sb_error().decode()


CASES = [
    # bath, norb, nbath, sector, nb0, chunk rows, extra
    ("normal", 1, 5, (3, 3), 2, 8, {}),
    ("normal", 1, 7, (4, 3), 4, 480, {}),
    ("normal", 1, 7, (3, 5), 3, 24, {}),
    ("normal", 2, 3, (4, 4), 2, 16, {}),                      # Jx = Jp != 0: Hnd terms, padded panels; one orbital per level
    ("normal", 2, 3, (3, 5), 3, 480, dict(jxp=0.0)),          # the three low levels all belong to orbital 1
    ("normal", 2, 4, (5, 4), 3, 40, {}),
    ("normal", 2, 4, (4, 4), 1, 60, {}),
    ("hybrid", 2, 5, (3, 4), 2, 12, {}),
    ("hybrid", 2, 6, (4, 4), 3, 64, {}),
    ("hybrid", 3, 4, (3, 4), 2, 20, {}),
    ("hybrid", 3, 5, (4, 4), 2, 24, {}),
    ("hybrid", 3, 5, (1, 7), 2, 480, {}),                     # classes missing on both sides
    ("hybrid", 3, 6, (5, 4), 3, 60, dict(jxp=0.0)),
    ("hybrid", 3, 6, (4, 5), 1, 30, {}),
    ("hybrid", 3, 4, (3, 3), 2, 480, dict(sundry=SUNDRY3)),
    ("normal", 2, 2, (3, 2), 1, 480, dict(sundry=SUNDRY2)),
    ("hybrid", 3, 4, (2, 3), 2, 12, dict(exc_field=np.array([0.12, 0.5, 0.5, 0.07]))),
    ("normal", 2, 3, (4, 3), 2, 480, dict(spin_field=np.array([[0.3, 0.1, 0.2], [0.0, 0.0, -0.15]]))),
]


@pytest.mark.parametrize("bath,norb,nbath,sec,nb0,rows,extra", CASES)
def test_local_block_product_matches_explicit_arrays(shim, monkeypatch, bath, norb, nbath, sec, nb0, rows, extra):
    monkeypatch.setenv("EDIGPU_SB_AMODE", "1")      # the per-orbital walk where the bath allows it (the default is the other)
    _, pm = make_models("normal", bath, norb, nbath, seed=41, **extra)
    rc, info, diff, msg = _check(shim, pm, sec[0], sec[1], nb0, rows, gs=4 if nbath % 2 else 8)
    assert rc == 0, msg
    assert info[0] == 1 and diff < 1e-13, (info, diff)
    assert info[7] == norb + nb0
    assert info[4] == (1 if bath == "normal" and norb > 1 else 0)
    if (norb == 1 or extra.get("jxp", 0.25) == 0.0) and "sundry" not in extra:
        assert info[5] == 0
    else:
        assert info[5] > 0


@pytest.mark.parametrize("bath,norb,nbath,sec,nb0,rows,extra", CASES)
def test_split_rows_match_explicit_arrays(shim, bath, norb, nbath, sec, nb0, rows, extra):
    """Rows staged in halves (one value of the top walked bit at a time, the hop over the top level from the vector
    itself): the form the rows kernel takes when a row image does not fit the LDS (Ns = 17).  Forced here."""
    _, pm = make_models("normal", bath, norb, nbath, seed=43, **extra)
    info = (C.c_int32 * 8)()
    diff = C.c_double(-1.0)
    m = pm.to_c()
    rc = shim.host_sb_check_split(C.byref(m), sec[0], sec[1], nb0, rows, 128, 8, 2, 4 if nbath % 2 else 8, info, C.byref(diff))
    msg = shim.host_sb_error().decode()
    if rc == 1:
        # every block has the same top bit, or a single walked level: refused (the whole-row kernels or the round-3 ones)
        assert "half of the split row is empty" in msg or "nothing to split" in msg or "every low word" in msg, msg
        return
    assert rc == 0, msg
    assert info[0] == 1 and info[6] >= 200 and diff.value < 1e-13, (list(info), diff.value)


def test_all_orbital_walk_on_a_one_orbital_per_level_bath(shim):
    """bath_type normal with the default (all-orbital) walk: the amplitudes of the other orbitals are zeros in the tables."""
    _, pm = make_models("normal", "normal", 2, 4, seed=47)
    rc, info, diff, msg = _check(shim, pm, 5, 4, 3, 40)
    assert rc == 0 and info[4] == 0 and diff < 1e-13, (msg, info, diff)


def test_wave_slot_plan_is_a_permutation(shim):
    """Any geometry of the rows kernel (threads, blocks per thread) and of the columns kernel (waves) gives the same product."""
    _, pm = make_models("normal", "hybrid", 3, 6, seed=43)
    for nt, nbt, nw, gs in ((64, 12, 1, 8), (128, 6, 3, 4), (256, 3, 4, 8), (512, 2, 8, 4)):
        rc, info, diff, msg = _check(shim, pm, 4, 5, 2, 48, nt, nbt, nw, gs)
        assert rc == 0 and diff < 1e-13, (nt, nbt, nw, gs, msg, diff)


def test_refusals(shim):
    _, pm = make_models("normal", "hybrid", 3, 6, seed=43)
    rc, _, _, msg = _check(shim, pm, 4, 5, 2, 48, nt=64, nbt=1)          # too few wave-slots
    assert rc == 1 and "wave-slots" in msg
    rc, _, _, msg = _check(shim, pm, 4, 5, 4, 48)                         # 3 + 4 local levels
    assert rc == 1 and "local levels" in msg
    _, pm = make_models("normal", "replica", 2, 2, seed=3)
    rc, _, _, msg = _check(shim, pm, 3, 3, 1, 480)
    assert rc == 1 and "bath-bath" in msg
