"""ctypes binding of the C ABI in include/edigpu.h (libedigpu.so, hipcc-built for gfx950).

This is the Python-side twin of the Fortran ISO_C_BINDING shim (fortran/edigpu_shim.f90):
plain pointers and sizes in, status code out.  There is no CPU fallback -- if the shared
library is missing or no HIP device is usable every call raises :class:`EdigpuError`.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EDIGPU_LIB", os.path.join(HERE, "lib", "libedigpu.so"))  # override: A/B builds
HEADER_PATH = os.path.join(os.path.dirname(HERE), "include", "edigpu.h")

MAXORB = 5
MAXSUNDRY = 64
MAXBATH = 16


class EdigpuError(RuntimeError):
    """Raised when a C-ABI call returns non-zero (the Fortran shim does ``stop msg``)."""


class EdigpuModel(C.Structure):
    """struct edigpu_model (include/edigpu.h)."""
    _fields_ = [
        ("ed_mode", C.c_int32), ("bath_type", C.c_int32),
        ("norb", C.c_int32), ("nbath", C.c_int32), ("nspin", C.c_int32), ("hfmode", C.c_int32),
        ("xmu", C.c_double),
        ("uloc", C.c_double * MAXORB),
        ("ust", C.c_double * (MAXORB * MAXORB)),
        ("jh", C.c_double * (MAXORB * MAXORB)),
        ("jx", C.c_double * (MAXORB * MAXORB)),
        ("jp", C.c_double * (MAXORB * MAXORB)),
        ("hloc", C.c_double * (2 * 2 * MAXORB * MAXORB * 2)),
        ("pair_field", C.c_double * MAXORB),
        ("be", C.c_double * (2 * MAXORB * MAXBATH)),
        ("bv", C.c_double * (2 * MAXORB * MAXBATH)),
        ("bd", C.c_double * (2 * MAXORB * MAXBATH)),
        ("bu", C.c_double * (2 * MAXORB * MAXBATH)),
        ("hb", C.c_double * (2 * 2 * MAXORB * MAXORB * MAXBATH * 2)),
        ("nph", C.c_int32), ("pad_", C.c_int32),
        ("w0_ph", C.c_double), ("a_ph", C.c_double),
        ("g_ph", C.c_double * (MAXORB * MAXORB)),
        ("spin_field", C.c_double * (MAXORB * 3)),
        ("exc_field", C.c_double * 4),
        ("nsundry", C.c_int32),
        ("pad2_", C.c_int32),
        ("sundry_op", C.c_int32 * (MAXSUNDRY * 8)),
        ("sundry_u", C.c_double * MAXSUNDRY),
    ]


_lib = None

_vp = C.c_void_p
_i64 = C.c_int64
_pi64 = C.POINTER(C.c_int64)
_pi32 = C.POINTER(C.c_int32)
_pd = C.POINTER(C.c_double)
_pint = C.POINTER(C.c_int)

# name -> (restype, argtypes); kept in one table so tests can check it against the header
SIGNATURES = {
    "edigpu_last_error": (C.c_char_p, []),
    "edigpu_version": (C.c_int, []),
    "edigpu_image_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "edigpu_model_sizeof": (C.c_int64, []),
    "edigpu_device_count": (C.c_int, [_pint]),
    "edigpu_init": (C.c_int, [C.c_int]),
    "edigpu_normal_create": (C.c_int, [C.POINTER(_vp), _i64, _i64, _i64, _i64, _pd, _pi64, _pi32, _pd,
                                       _pi64, _pi32, _pd, _pi64, _pi32, _pd]),
    "edigpu_csr_create_d": (C.c_int, [C.POINTER(_vp), _i64, _i64, _i64, _pi64, _pi32, _pd]),
    "edigpu_csr_create_z": (C.c_int, [C.POINTER(_vp), _i64, _i64, _i64, _pi64, _pi32, _pd]),
    "edigpu_normal_build": (C.c_int, [C.POINTER(_vp), C.POINTER(EdigpuModel), C.c_int, C.c_int, _i64, _i64]),
    "edigpu_normal_build_z": (C.c_int, [C.POINTER(_vp), C.POINTER(EdigpuModel), C.c_int, C.c_int]),
    "edigpu_flat_build": (C.c_int, [C.POINTER(_vp), C.POINTER(EdigpuModel), C.c_int, _i64, _i64]),
    "edigpu_flat_build_jz": (C.c_int, [C.POINTER(_vp), C.POINTER(EdigpuModel), C.c_int, C.c_int, _i64, _i64]),
    "edigpu_direct_build_jz": (C.c_int, [C.POINTER(_vp), C.POINTER(EdigpuModel), C.c_int, C.c_int, _i64, _i64]),
    "edigpu_sector_map_jz": (C.c_int, [_vp, C.c_int, C.c_int, _pi32, _pi64]),
    "edigpu_direct_build": (C.c_int, [C.POINTER(_vp), C.POINTER(EdigpuModel), C.c_int, _i64, _i64]),
    "edigpu_sector_dim": (C.c_int, [C.POINTER(EdigpuModel), C.c_int, C.c_int, _pi64]),
    "edigpu_info": (C.c_int, [_vp, _pi64]),
    "edigpu_algorithmic_bytes": (C.c_int, [_vp, _pd, _pd]),
    "edigpu_normal_export": (C.c_int, [_vp, _pd, _pi64, _pi32, _pd, _pi64, _pi32, _pd, _pi64, _pi32, _pd]),
    "edigpu_csr_export": (C.c_int, [_vp, _pi64, _pi32, _pd]),
    "edigpu_apply_d": (C.c_int, [_vp, _i64, _pd, _pd]),
    "edigpu_apply_z": (C.c_int, [_vp, _i64, _pd, _pd]),
    "edigpu_apply_dev": (C.c_int, [_vp, _vp, _vp, _vp]),
    "edigpu_apply_local_dev": (C.c_int, [_vp, _vp, _vp, _vp]),
    "edigpu_apply_remote_dev": (C.c_int, [_vp, _vp, _vp, _vp]),
    "edigpu_normal_transpose_info": (C.c_int, [_vp, C.POINTER(C.c_int32)]),
    "edigpu_normal_apply_rows_dev": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp]),
    "edigpu_normal_apply_cols_dev": (C.c_int, [_vp, _i64, _i64, _i64, C.c_int32, _vp, _vp, _vp]),
    "edigpu_transpose_pack": (C.c_int, [_i64, _i64, _i64, C.c_int32, _i64, C.c_int32, _vp, _vp, _vp]),
    "edigpu_transpose_unpack_add": (C.c_int, [_i64, _i64, _i64, C.c_int32, _i64, C.c_int32, _vp, _vp, _vp]),
    "edigpu_transpose_rotate_pack": (C.c_int, [C.c_int32, _i64, _i64, _i64, C.c_int32, _i64, C.c_int32, _vp, _vp, _vp,
                                               _vp, _vp]),
    "edigpu_transpose_unpack_add_dot2": (C.c_int, [_i64, _i64, _i64, C.c_int32, _i64, C.c_int32, _vp, _vp, _vp, _vp,
                                                   _vp, _vp, _vp]),
    "edigpu_lanczos_tridiag": (C.c_int, [_vp, _pd, C.c_int, _pd, _pd, C.c_double, _pint]),
    "edigpu_lanczos_eigh": (C.c_int, [_vp, C.c_int, C.c_double, C.c_int, _pd, _pd, _pd, _pint]),
    "edigpu_orbs_build": (C.c_int, [C.POINTER(_vp), C.POINTER(EdigpuModel), _pi32, _pi32]),
    "edigpu_orbs_build_rows": (C.c_int, [C.POINTER(_vp), C.POINTER(EdigpuModel), _pi32, _pi32, _i64, _i64]),
    "edigpu_orbs_create": (C.c_int, [C.POINTER(_vp), C.c_int, _pi64, _pd, _pi64, _pi32, _pd]),
    "edigpu_lanczos_eigh_multi": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, C.c_int, _vp, _pd, _vp,
                                            C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "edigpu_apply_op_normal": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "edigpu_apply_op_flat": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "edigpu_apply_cops_normal": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _pd, _pi32, _pi32, _pi32, _vp]),
    "edigpu_lanczos_tridiag_dev": (C.c_int, [_vp, _vp, C.c_int, _pd, _pd, C.c_double, C.POINTER(C.c_int), _pd]),
    "edigpu_vec_work_doubles": (C.c_int, []),
    "edigpu_vec_rotate": (C.c_int, [_i64, _vp, _vp, _vp, _vp]),
    "edigpu_vec_add_dot": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "edigpu_vec_axpy_nrm2": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "edigpu_vec_scale": (C.c_int, [_i64, _vp, _vp, _vp]),
    "edigpu_vec_rotate_lazy": (C.c_int, [_i64, _vp, _vp, _vp, _vp]),
    "edigpu_vec_add_dot2": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "edigpu_time_apply": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _pd]),
    "edigpu_lanczos_bench": (C.c_int, [_vp, C.c_int, C.c_int, _pd, _pd]),
    "edigpu_membw": (C.c_int, [_i64, _pd]),
    "edigpu_sector_map": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _pi32, _pi64]),
    "edigpu_cache_create": (C.c_int, [C.POINTER(_vp), _i64]),
    "edigpu_cache_get": (C.c_int, [_vp, C.POINTER(EdigpuModel), C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "edigpu_cache_stats": (C.c_int, [_vp, _pi64]),
    "edigpu_cache_clear": (C.c_int, [_vp]),
    "edigpu_cache_destroy": (C.c_int, [_vp]),
    "edigpu_dev_alloc": (C.c_int, [_i64, C.POINTER(_vp)]),
    "edigpu_dev_free": (C.c_int, [_vp]),
    "edigpu_dev_upload": (C.c_int, [_vp, _vp, _i64]),
    "edigpu_dev_download": (C.c_int, [_vp, _vp, _i64]),
    "edigpu_shard_plan": (C.c_int, [_i64, C.c_int32, C.c_int32, _pi64, _pi64, _pi64]),
    "edigpu_shard_info": (C.c_int, [_vp, _vp, C.POINTER(C.c_int32)]),
    "edigpu_exchange_send_map": (C.c_int, [_i64, _i64, _i64, C.c_int32, _i64, C.c_int32, _pi64]),
    "edigpu_exchange_back_map": (C.c_int, [_i64, _i64, _i64, C.c_int32, _i64, C.c_int32, _pi64]),
    "edigpu_comm_unique_id": (C.c_int, [_vp]),
    "edigpu_comm_create": (C.c_int, [C.POINTER(_vp), C.c_int32, C.c_int32, _vp]),
    "edigpu_comm_create_shm": (C.c_int, [C.POINTER(_vp), C.c_int32, C.c_int32, C.c_char_p, _i64]),
    "edigpu_comm_info": (C.c_int, [_vp, _pi32, _pi32, _pi32]),
    "edigpu_comm_destroy": (C.c_int, [_vp]),
    "edigpu_apply_sharded_d": (C.c_int, [_vp, _vp, _i64, _pd, _pd]),
    "edigpu_apply_sharded_z": (C.c_int, [_vp, _vp, _i64, _pd, _pd]),
    "edigpu_lanczos_tridiag_sharded": (C.c_int, [_vp, _vp, _vp, C.c_int, _pd, _pd, C.c_double, _pint, _pd]),
    "edigpu_lanczos_eigh_multi_sharded": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_double, C.c_int, _vp, _pd, _vp, _pint,
                                                    _pint]),
    "edigpu_lanczos_eigh_sharded": (C.c_int, [_vp, _vp, C.c_int, C.c_double, _vp, _pd, _vp, _pint]),
    "edigpu_exchange_bench": (C.c_int, [_vp, _vp, C.c_int, C.POINTER(C.c_int32), _pd]),
    "edigpu_apply_cops_sharded": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, _pd, _vp, _vp, _vp]),
    "edigpu_apply_cops_flat": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _pd, _vp, _vp, _vp, _vp]),
    "edigpu_lanczos_bench_sharded": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _pd, _pi64]),
    "edigpu_destroy": (C.c_int, [_vp]),
}


def lib() -> C.CDLL:
    """Load libedigpu.so (built by ``__graft_entry__.build()``); fail loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EdigpuError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so (SONAME
    # libamdhip64.so.7) and looks it up by file name, so if libedigpu.so pulled in /opt/rocm's copy
    # first, torch would load a second runtime that cannot see the GPU.  Importing torch first makes
    # the dynamic loader resolve our NEEDED libamdhip64.so.7 to the copy torch already mapped.
    # Without torch (e.g. the Fortran host) the system runtime is used.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional plumbing
        pass
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def last_error() -> str:
    msg = lib().edigpu_last_error()
    return msg.decode() if msg else ""


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise EdigpuError(f"{what}: {last_error()}" if what else last_error())


def device_count() -> int:
    n = C.c_int(0)
    rc = lib().edigpu_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def init(device: int = 0) -> None:
    check(lib().edigpu_init(int(device)), "edigpu_init")


def membw(nbytes: int = 1 << 30):
    """(read, copy, triad) GB/s streaming ceilings of the selected device (edigpu_membw)."""
    out = (C.c_double * 3)()
    check(lib().edigpu_membw(int(nbytes), out), "edigpu_membw")
    return tuple(float(x) for x in out)


def kernel_source_hash() -> str:
    """sha256 over the kernel / host sources of libedigpu.so: profiles/pmc_traffic.json is stamped with it, so a
    counter-derived traffic figure is only reported for the kernels it was measured on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(HERE, "csrc", "*.h*")) + glob.glob(os.path.join(HERE, "csrc", "*.cpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pd(a: np.ndarray):
    return a.ctypes.data_as(_pd)


def pi64(a: np.ndarray):
    return a.ctypes.data_as(_pi64)


def pi32(a: np.ndarray):
    return a.ctypes.data_as(_pi32)
