"""edipack_amd -- MI355X (gfx950) Lanczos H*v engine for EDIpack's Hamiltonian hot path.

Only what the path needs: the C-ABI library (csrc/ -> lib/libedigpu.so), its ctypes binding
(:mod:`edipack_amd.capi`), the host-side mirror of the reference's ED_HAMILTONIAN interface
(:mod:`edipack_amd.hamiltonian`) and the multi-GPU sharding plan (:mod:`edipack_amd.sharding`).
"""
from . import capi  # noqa: F401
from .capi import EdigpuError  # noqa: F401
from .hamiltonian import (ImpurityModel, SectorHamiltonian, build_Hv_sector_nonsu2,  # noqa: F401
                          build_Hv_sector_normal, build_Hv_sector_superc, delete_Hv_sector,
                          spHtimesV_cc, spHtimesV_p, tridiag_Hv_sector, vecDim_Hv_sector)

__version__ = "0.1.0"
