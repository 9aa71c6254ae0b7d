"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports every
symbol include/edigpu.h declares, and fails loudly (no CPU fallback) when no GPU is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from edipack_amd import capi


def _declared_symbols():
    txt = open(capi.HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(edigpu_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(built):
    L = capi.lib()
    decl = _declared_symbols()
    assert len(decl) >= 20
    for name in decl:
        assert hasattr(L, name), f"{name} declared in include/edigpu.h but not exported"
    # and the Python binding table covers the header exactly
    assert sorted(capi.SIGNATURES) == decl


def test_model_struct_layout_matches_header(built):
    """sizeof(edigpu_model) computed from the header's array extents == ctypes layout."""
    mo, mb, ms = capi.MAXORB, capi.MAXBATH, capi.MAXSUNDRY
    n_int, n_dbl = 6, 1 + mo + 4 * mo * mo + 2 * 2 * mo * mo * 2 + mo + 4 * 2 * mo * mb + 2 * 2 * mo * mo * mb * 2 + 1 + 2 + mo * mo
    n_int, n_dbl = n_int + 2 + 8 * ms, n_dbl + 3 * mo + 4 + ms      # spin_field, exc_field, coulomb_sundry
    assert C.sizeof(capi.EdigpuModel) == n_int * 4 + n_dbl * 8
    assert C.sizeof(capi.EdigpuModel) == capi.lib().edigpu_model_sizeof()


def test_error_reporting_without_compute(built):
    L = capi.lib()
    assert L.edigpu_version() >= 100
    # NULL handle / bad arguments are rejected with a message, not a crash
    assert L.edigpu_info(None, None) != 0 and "NULL" in capi.last_error()
    assert L.edigpu_destroy(None) == 0


def test_fails_loudly_without_gpu(built):
    """On a box without a HIP device (this container) construction must raise, never fall back."""
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    from edipack_amd.hamiltonian import SectorHamiltonian
    with pytest.raises(capi.EdigpuError, match="no usable HIP device|no CPU fallback"):
        SectorHamiltonian.csr_from_arrays(np.array([0, 1], np.int64), np.array([0], np.int32), np.array([1.0]))
    with pytest.raises(capi.EdigpuError):
        capi.init(0)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under edipack_amd/ may reference it."""
    root = os.path.dirname(capi.HERE)
    for dp, _, files in os.walk(os.path.join(root, "edipack_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "edipack_oracle" not in txt, f


def test_sector_maps_are_bit_exact(built):
    """Index data bit-exact: the library's sector maps against the oracle's restatement of
    build_sector (ED_SECTOR.f90:165-373) for every sector of small normal / superc / nonsu2 models.  Host-only."""
    import numpy as np
    from edipack_amd.hamiltonian import sector_map
    from oracle import oracle as O
    from tests.common import make_models
    for mode, bath, norb, nbath in (("normal", "normal", 2, 2), ("normal", "hybrid", 3, 4), ("superc", "hybrid", 2, 3),
                                    ("nonsu2", "normal", 1, 4)):
        om, pm = make_models(mode, bath, norb, nbath, seed=1)
        for sec in O.sectors(om):
            if mode == "normal":
                ho = O.HNormal(om, *sec)
                assert np.array_equal(sector_map(pm, sec[0], sec[1], 0), ho.mapup.astype(np.int32))
                assert np.array_equal(sector_map(pm, sec[0], sec[1], 1), ho.mapdw.astype(np.int32))
            else:
                ho = O.HFlat(om, sec)
                assert np.array_equal(sector_map(pm, sec), ho.map.astype(np.int32)), (mode, sec)
