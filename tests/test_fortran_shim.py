"""The Fortran side of the drop-in boundary: fortran/edigpu_shim.f90 (ISO_C_BINDING module with
dd_sparse_HxV / cc_sparse_HxV compatible procedures) compiled with flang and linked against
libedigpu.so; fortran/test_shim.f90 drives it through procedure pointers like the reference's
spHtimesV_p / spHtimesV_cc (ED_VARS_GLOBAL.f90:196-197)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "fortran", "build", "test_shim")
FLANG = "/opt/rocm/lib/llvm/bin/flang"


@pytest.mark.skipif(not os.path.exists(FLANG), reason="flang not present")
def test_fortran_shim_builds_and_fails_loudly_without_gpu(built):
    assert os.path.exists(EXE), "build() did not produce fortran/build/test_shim"
    from edipack_amd import capi
    if capi.device_count() > 0:
        pytest.skip("a GPU is present (covered by the gpu test)")
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0
    assert "no usable HIP device" in (r.stdout + r.stderr)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(FLANG), reason="flang not present")
def test_fortran_host_through_shim(gpu):
    assert os.path.exists(EXE)
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "FORTRAN SHIM OK" in r.stdout


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(FLANG), reason="flang not present")
@pytest.mark.parametrize("world", [2, 3])
def test_fortran_mpi_ranks_through_shim(gpu, world):
    """The -D_MPI side of the boundary: `world` Fortran processes (one per MPI rank) share the GPU, create the library's
    communicator (shared-memory transport) through the shim and call spMatVec_mpi_gpu_d / gpu_lanc_tridiag_mpi_d on
    their shards -- the (Nloc, v, Hv) contract of spMatVec_mpi_normal_main and sp_lanc_tridiag(MpiComm, ...)."""
    name = f"edigpu_f90_{os.getpid()}_{world}"
    procs = [subprocess.Popen([EXE, "mpi", str(r), str(world), name], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              text=True) for r in range(world)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        print(o)
        assert p.returncode == 0 and "FORTRAN SHIM MPI OK" in o, o
