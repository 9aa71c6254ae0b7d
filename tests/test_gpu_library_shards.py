"""The N > 1 path inside libedigpu.so (csrc/edigpu_shard.hip), called through the C ABI: communicator, the
(Nloc, v, Hv) product on shards (spMatVec_mpi_*) and the sharded tridiagonalisation (sp_lanc_tridiag with MpiComm),
against the CPU oracle.

World size 1: the RCCL communicator (collectives short-circuited) and the shared-memory one.  World sizes 2 and 3:
ranks share the one GPU of the test box and exchange through the library's host-staged shared-memory transport --
the same C loop, pack / unpack kernels, halo columns and padded tails as under RCCL; only the transport differs.
"""
import multiprocessing as mp
import os

import numpy as np
import pytest

from tests.common import make_models, rel_err

pytestmark = pytest.mark.gpu

CASES = [
    # mode, bath, norb, nbath, sector, direct, exchange
    ("normal", "normal", 2, 3, (4, 4), False, "auto"),       # transposed exchange, Hnd through halo columns
    ("normal", "hybrid", 3, 3, (3, 2), False, "auto"),       # 3 orbitals: halo 2, DimDw = 15 not divisible by 2 ranks... ragged tails
    ("normal", "normal", 2, 3, (4, 4), False, "allgather"),  # down-row shards, all-gather form
    ("superc", "hybrid", 2, 3, 0, False, "auto"),            # stored flat CSR, loc / non-loc blocks
    ("nonsu2", "hybrid", 2, 3, 5, True, "auto"),             # on-the-fly kernel: gather first
    ("normal", "hybrid", 3, 2, (3, 2), False, "cmplx"),      # _CMPLX_NORMAL through its doubled real sector
    ("normal", "normal", 2, 2, (3, 2), False, "phonon"),     # (Nph + 1) blocks, one exchange per block, local phonon pass
    ("superc", "hybrid", 2, 2, 0, False, "phonon"),          # stored rows: one all-gather per phonon block
    ("nonsu2", "normal", 2, 2, 4, True, "phonon"),           # on the fly
    # transposed exchange on the padded panels of the local-block kernels (round 4): no packing, no halo
    ("normal", "normal", 2, 3, (4, 4), False, "block"),      # Hnd terms inside the panels, 5 panels over 1-3 ranks
    ("normal", "hybrid", 3, 5, (4, 3), False, "block"),      # three orbitals; DimDw = 56 rows: ragged tails, blocks cut by ranks
    ("normal", "hybrid", 3, 6, (4, 5), False, "block"),      # 126 x 126
    # the same exchange with the recurrence's vectors in the reference's row layout (EDIGPU_SHARD_PANEL_LOOP=0: a conversion
    # on either side of every product) -- the default keeps them in the panel layout from the seed on
    ("normal", "hybrid", 3, 5, (4, 3), False, "block-rowloop"),
]
NPH = 3


def _reference(mode, bath, norb, nbath, sector, seed=31, cmplx=False, phonon=False):
    from oracle import oracle as O
    om, pm = make_models(mode, bath, norb, nbath, seed=seed)
    if phonon:
        for m in (om, pm):
            m.nph, m.w0_ph, m.a_ph, m.g_ph = NPH, 0.8, 0.15, np.diag([0.4, -0.3][:norb])
    if cmplx:
        from tests.test_gpu_parity import _complexify
        _complexify(om, pm, seed + 1)
        ho = O.HNormalCmplx(om, *sector)
    else:
        ho = O.HNormal(om, *sector) if mode == "normal" else O.HFlat(om, sector)
    rng = np.random.default_rng(17)
    v = rng.standard_normal(ho.dim)
    if mode != "normal" or cmplx:
        v = v + 1j * rng.standard_normal(ho.dim)
    return ho, pm, v


def _shard_index(ho, mode, first, count, phonon):
    """Global indices of a rank's shard in the order the reference holds them (phonon sectors: block after block of the
    rank's down rows, spMatVec_mpi_normal_main's i = iup + (idw-1) DimUp + (iph-1) DimUp MpiQdw)."""
    ul = ho.dimup if mode == "normal" else 1
    base = np.arange(first * ul, (first + count) * ul)
    if not phonon:
        return base
    return np.concatenate([b * ho.dim_el + base for b in range(NPH + 1)])


def _rank_main(rank, world, name, case, q):
    try:
        import torch  # noqa: F401  (one HIP runtime per process)
        from edipack_amd import capi
        from edipack_amd.sharding import LibraryComm, library_sharded_sector
        capi.init(0)
        mode, bath, norb, nbath, sector, direct, exchange = case
        ho, pm, v = _reference(mode, bath, norb, nbath, sector, cmplx=exchange == "cmplx", phonon=exchange == "phonon")
        comm = LibraryComm(rank, world, shm_name=name, slot_bytes=1 << 22)
        block = exchange.startswith("block")
        if block:      # small sectors get the impurity-block image + local-block tables on request only
            os.environ.update(EDIGPU_IB="1", EDIGPU_IB_MIN="0", EDIGPU_IB_ROWS="24")
            if exchange == "block-rowloop":
                os.environ["EDIGPU_SHARD_PANEL_LOOP"] = "0"
        h, first, count = library_sharded_sector(pm, sector, comm, direct=direct, exchange="auto" if block else exchange,
                                                 cmplx=exchange == "cmplx")
        if block:
            assert comm.shard_info(h)[0] == 2, comm.shard_info(h)
        elif exchange == "auto" and mode == "normal":
            assert comm.shard_info(h)[0] == 1, comm.shard_info(h)
        ix = _shard_index(ho, mode, first, count, exchange == "phonon")
        hv = comm.apply(h, v[ix])
        a, b, nd, n2 = comm.tridiag(h, v[ix], 20)
        h.destroy()
        comm.destroy()
        q.put((rank, ix, None, hv, a, b, nd, n2, None))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, 0, 0, None, None, None, 0, 0.0, traceback.format_exc() + str(e)))


@pytest.mark.parametrize("case", CASES, ids=[f"{c[0]}-{c[1]}-{c[6]}{'-direct' if c[5] else ''}" for c in CASES])
@pytest.mark.parametrize("world", [1, 2, 3])
def test_library_shards_share_one_gpu(gpu, world, case):
    mode, bath, norb, nbath, sector, direct, exchange = case
    ho, _, v = _reference(mode, bath, norb, nbath, sector, cmplx=exchange == "cmplx", phonon=exchange == "phonon")
    ref = ho.matvec(v)
    a_ref, b_ref, _ = ho.lanc_tridiag(v, 20)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = f"edigpu_test_{os.getpid()}_{world}_{abs(hash(case)) % 100000}"
    procs = [ctx.Process(target=_rank_main, args=(r, world, name, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[8] is None, r[8]
    got = np.zeros_like(ref)
    for rank, ix, _, hv, a, b, nd, n2, _ in res:
        got[ix] = hv
        assert nd == 20 and abs(n2 - np.real(np.vdot(v, v))) < 1e-10 * abs(n2)
        assert rel_err(a, a_ref) < 1e-10 and rel_err(b, b_ref) < 1e-10      # every rank holds the same coefficients
    assert rel_err(got, ref) < 1e-12


# ---------------------------------------------------------------------------------------------------------
# eigenpairs with the vector sharded (sp_eigh / sp_lanc_eigh with MpiComm,
# ED_NORMAL/ED_DIAG_NORMAL.f90:179-214)
# ---------------------------------------------------------------------------------------------------------
EIG_CASES = [CASES[0], CASES[1], CASES[2], CASES[3], CASES[4], CASES[5], CASES[6], CASES[10]]


def _eig_rank_main(rank, world, name, case, q):
    try:
        import torch  # noqa: F401
        from edipack_amd import capi
        from edipack_amd.sharding import LibraryComm, library_sharded_sector
        capi.init(0)
        mode, bath, norb, nbath, sector, direct, exchange = case
        ho, pm, v = _reference(mode, bath, norb, nbath, sector, cmplx=exchange == "cmplx", phonon=exchange == "phonon")
        comm = LibraryComm(rank, world, shm_name=name, slot_bytes=1 << 22)
        if exchange == "block":
            os.environ.update(EDIGPU_IB="1", EDIGPU_IB_MIN="0", EDIGPU_IB_ROWS="24")
        h, first, count = library_sharded_sector(pm, sector, comm, direct=direct, exchange="auto" if exchange == "block" else exchange,
                                                 cmplx=exchange == "cmplx")
        assert exchange != "block" or comm.shard_info(h)[0] == 2
        ix = _shard_index(ho, mode, first, count, exchange == "phonon")
        ev, x, nconv, nmv = comm.eigh_multi(h, 3, len(ix), v0_shard=v[ix], tol=1e-11)
        e1, x1, nm1 = comm.eigh(h, len(ix), v0_shard=v[ix], tol=1e-11)
        h.destroy()
        comm.destroy()
        q.put((rank, ix, ev, x, nconv, nmv, e1, x1, None))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, 0, None, None, 0, 0, 0.0, None, traceback.format_exc() + str(e)))


@pytest.mark.parametrize("case", EIG_CASES, ids=[f"{c[0]}-{c[1]}-{c[6]}{'-direct' if c[5] else ''}" for c in EIG_CASES])
@pytest.mark.parametrize("world", [1, 2, 3])
def test_library_eigenpairs_on_shards(gpu, world, case):
    """Lowest eigenpairs by the thick-restart solver with every vector a shard: eigenvalues against dense
    diagonalisation of the oracle's matrix (1e-10 relative), vectors by their residual on the assembled vector and
    their mutual orthogonality; every rank must report the same numbers."""
    mode, bath, norb, nbath, sector, direct, exchange = case
    ho, _, v = _reference(mode, bath, norb, nbath, sector, cmplx=exchange == "cmplx", phonon=exchange == "phonon")
    dense = ho.dense()
    w_ref = np.linalg.eigvalsh(dense)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = f"edigpu_eig_{os.getpid()}_{world}_{abs(hash(case)) % 100000}"
    procs = [ctx.Process(target=_eig_rank_main, args=(r, world, name, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[8] is None, r[8]
    X = np.zeros((3, ho.dim), dtype=dense.dtype if np.iscomplexobj(v) else float)
    x1 = np.zeros(ho.dim, dtype=X.dtype)
    for rank, ix, ev, x, nconv, nmv, e1, xs, _ in res:
        assert nconv == 3 and nmv > 0
        assert np.allclose(ev, res[0][2], rtol=0, atol=0)                      # the same on every rank
        assert abs(e1 - res[0][6]) == 0.0
        X[:, ix] = x
        x1[ix] = xs
    ev, scale = res[0][2], max(1.0, np.abs(w_ref).max())
    assert np.max(np.abs(ev - w_ref[:3])) < 1e-10 * scale
    assert abs(res[0][6] - w_ref[0]) < 1e-10 * scale
    for i in range(3):
        assert np.linalg.norm(dense @ X[i] - ev[i] * X[i]) < 1e-8 * scale
    assert np.linalg.norm(X.conj() @ X.T - np.eye(3)) < 1e-9
    assert np.linalg.norm(dense @ x1 - res[0][6] * x1) < 1e-7 * scale and abs(np.linalg.norm(x1) - 1.0) < 1e-10


# ---------------------------------------------------------------------------------------------------------
# apply_Cops on shards: the Green's-function seeds (ED_NORMAL/ED_GF_NORMAL.f90:141-175) without gathering on a master
# ---------------------------------------------------------------------------------------------------------
COPS_CASES = [
    # mode, bath, norb, nbath, source sector, destination sector, ops [(coef, create, iorb, ispin)]
    ("normal", "normal", 2, 3, (4, 4), (3, 4), [(1.0, False, 0, 0), (0.7, False, 1, 0)]),     # c_up combination
    ("normal", "normal", 2, 3, (4, 4), (4, 3), [(1.0, False, 1, 1)]),                         # c_dw: rows move
    ("normal", "hybrid", 3, 3, (3, 2), (3, 3), [(1.0, True, 2, 1), (-0.4, True, 0, 1)]),      # c+_dw, ragged shards
    ("normal", "hybrid", 3, 3, (3, 2), (4, 2), [(1.0, True, 1, 0)]),                          # c+_up
    ("nonsu2", "hybrid", 2, 3, 5, 4, [(1.0, False, 0, 0), (-1.0j, False, 1, 1)]),             # c_1up - i c_2dw
    ("nonsu2", "hybrid", 2, 3, 5, 6, [(0.5, True, 1, 1)]),
    ("superc", "hybrid", 2, 3, 0, -1, [(1.0, False, 0, 0), (1.0, True, 1, 1)]),               # both lead to Sz - 1
]


def _cops_rank_main(rank, world, name, case, q):
    try:
        import torch  # noqa: F401
        from edipack_amd import capi
        from edipack_amd.sharding import LibraryComm, library_sharded_sector
        capi.init(0)
        mode, bath, norb, nbath, s1, s2, ops = case
        ho, pm, v = _reference(mode, bath, norb, nbath, s1)
        comm = LibraryComm(rank, world, shm_name=name, slot_bytes=1 << 22)
        h1, f1, c1 = library_sharded_sector(pm, s1, comm)
        h2, f2, c2 = library_sharded_sector(pm, s2, comm)
        ul1 = ho.dimup if mode == "normal" else 1
        ul2 = h2.dim_up if mode == "normal" else 1
        out = comm.apply_cops(h1, h2, v[f1 * ul1:(f1 + c1) * ul1], c2 * ul2, ops)
        h1.destroy(), h2.destroy()
        comm.destroy()
        q.put((rank, f2 * ul2, out, None))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, 0, None, traceback.format_exc() + str(e)))


@pytest.mark.parametrize("case", COPS_CASES, ids=[f"{c[0]}-{c[4]}-{c[5]}" for c in COPS_CASES])
@pytest.mark.parametrize("world", [1, 2, 3])
def test_library_apply_cops_on_shards(gpu, world, case):
    """Every rank's shard of sum_s coef_s O_s v against the same combination applied to the whole vector on one GPU
    (edigpu_apply_cops_normal / edigpu_apply_cops_flat, which the golden observables pin on the reference's files)."""
    import torch
    from edipack_amd.hamiltonian import SectorHamiltonian
    mode, bath, norb, nbath, s1, s2, ops = case
    ho, pm, v = _reference(mode, bath, norb, nbath, s1)
    if mode == "normal":
        g1, g2 = SectorHamiltonian.normal_from_model(pm, *s1), SectorHamiltonian.normal_from_model(pm, *s2)
        src = torch.from_numpy(np.ascontiguousarray(v)).cuda()
        dst = torch.zeros(g2.dim, dtype=torch.float64, device="cuda")
        g1.apply_cops_to(g2, src.data_ptr(), dst.data_ptr(), [o[0] for o in ops], [o[1] for o in ops], [o[2] for o in ops],
                         [o[3] for o in ops], torch.cuda.current_stream().cuda_stream)
    else:
        g1, g2 = SectorHamiltonian.flat_from_model(pm, s1), SectorHamiltonian.flat_from_model(pm, s2)
        src = torch.from_numpy(np.ascontiguousarray(v, dtype=np.complex128)).cuda()
        dst = torch.zeros(g2.dim, dtype=torch.complex128, device="cuda")
        g1.apply_cops_flat_to(g2, src.data_ptr(), dst.data_ptr(), [o[0] for o in ops], [o[1] for o in ops],
                              [o[2] for o in ops], [o[3] for o in ops], torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ref = dst.cpu().numpy()
    assert np.linalg.norm(ref) > 0.0
    g1.destroy(), g2.destroy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = f"edigpu_cops_{os.getpid()}_{world}_{abs(hash(str(case))) % 100000}"
    procs = [ctx.Process(target=_cops_rank_main, args=(r, world, name, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    got = np.zeros_like(ref)
    for rank, off, out, err in res:
        assert err is None, err
        got[off:off + len(out)] = out
    assert rel_err(got, ref) < 1e-14


@pytest.mark.parametrize("force", [True, False])
def test_library_comm_rccl_world_of_one(gpu, monkeypatch, force):
    """The RCCL communicator itself (ncclCommInitRank with a unique id, world of one) and the sharded calls on it:
    with the collectives forced (the N > 1 code path, RCCL calls included) and as a user gets it (a world of one
    rank is handed to the fused single-GPU recurrence)."""
    import torch  # noqa: F401
    if force:
        monkeypatch.setenv("EDIGPU_FORCE_COLLECTIVES", "1")
    from edipack_amd.sharding import LibraryComm, library_sharded_sector
    uid = LibraryComm.unique_id()
    assert len(uid) == 128
    comm = LibraryComm(0, 1, unique_id=uid)
    for case in CASES[:1] + CASES[3:4] + CASES[-3:]:
        mode, bath, norb, nbath, sector, direct, exchange = case
        ho, pm, v = _reference(mode, bath, norb, nbath, sector, phonon=exchange == "phonon")
        h, first, count = library_sharded_sector(pm, sector, comm, direct=direct, exchange=exchange)
        assert (first, count) == (0, ho.dimdw if mode == "normal" else ho.dim_el)
        assert rel_err(comm.apply(h, v), ho.matvec(v)) < 1e-12
        a, b, nd, n2 = comm.tridiag(h, v, 25)
        a_ref, b_ref, _ = ho.lanc_tridiag(v, 25)
        assert nd == 25 and rel_err(a, a_ref) < 1e-10 and rel_err(b, b_ref) < 1e-10
        # breakdown: a seed inside a one-dimensional invariant subspace cannot be made here; a zero seed returns norm2 = 0
        a0, b0, nd0, n20 = comm.tridiag(h, np.zeros_like(v), 5)
        assert n20 == 0.0 and nd0 == 0 and not a0.any()
        h.destroy()
    comm.destroy()


def test_one_communicator_serves_sectors_of_changing_geometry(gpu, monkeypatch):
    """A communicator is shared across sectors: all-gather shards (superc / nonsu2, explicit arrays) and transposed
    whole sectors of different sizes in turn.  Its workspace buffers grow one by one (the gathered-vector buffer used to
    keep the size of the first all-gather call when a larger transposed call came in between: a device heap overflow
    on the third call).  RCCL world of one with the collectives forced, so that every buffer is really written."""
    import torch  # noqa: F401
    monkeypatch.setenv("EDIGPU_FORCE_COLLECTIVES", "1")
    from edipack_amd.sharding import LibraryComm, library_sharded_sector
    comm = LibraryComm(0, 1, unique_id=LibraryComm.unique_id())
    seq = [("superc", "hybrid", 2, 2, 0, False, "auto"),        # all-gather, small
           ("normal", "normal", 2, 3, (4, 4), False, "auto"),   # transposed, larger chunk
           ("nonsu2", "hybrid", 2, 3, 5, False, "auto"),        # all-gather again, larger than the first
           ("normal", "hybrid", 3, 3, (3, 2), False, "allgather"),
           ("superc", "hybrid", 2, 3, 0, False, "auto"),
           ("normal", "normal", 2, 3, (4, 4), False, "allgather"),
           # padded panels, the recurrence kept in that layout: a larger sector, then a smaller one whose padding lies
           # where the larger one left numbers (the loop clears its buffers when it starts)
           ("normal", "hybrid", 3, 5, (4, 3), False, "block"),
           ("normal", "normal", 2, 3, (4, 4), False, "block"),
           ("normal", "hybrid", 3, 5, (4, 3), False, "auto")]
    for mode, bath, norb, nbath, sector, direct, exchange in seq:
        ho, pm, v = _reference(mode, bath, norb, nbath, sector)
        for k, val in (("EDIGPU_IB", "1"), ("EDIGPU_IB_MIN", "0"), ("EDIGPU_IB_ROWS", "24")):
            if exchange == "block":
                monkeypatch.setenv(k, val)
            else:
                monkeypatch.delenv(k, raising=False)
        h, first, count = library_sharded_sector(pm, sector, comm, direct=direct,
                                                 exchange="auto" if exchange == "block" else exchange)
        assert (comm.shard_info(h)[0] == 2) == (exchange == "block")
        assert rel_err(comm.apply(h, v), ho.matvec(v)) < 1e-12
        a, b, nd, _ = comm.tridiag(h, v, 12)
        a_ref, b_ref, _ = ho.lanc_tridiag(v, 12)
        assert nd == 12 and rel_err(a, a_ref) < 1e-10 and rel_err(b, b_ref) < 1e-10
        assert comm.exchange_bench(h, 3)[0] in (-1, 1)          # RCCL (world of one: no communicator object, reported as 1)
        h.destroy()
    comm.destroy()


def _full_rank_main(rank, world, name, wl, nlanc, q):
    try:
        import torch  # noqa: F401
        from edipack_amd import capi
        from edipack_amd.sharding import LibraryComm, library_sharded_sector
        from edipack_amd.synthetic import WORKLOADS, synthetic_model
        capi.init(0)
        os.environ["EDIGPU_IB_MINROW"] = "0"     # what bench.py --gpus N sets: shards of short rows on the padded panels
        w = WORKLOADS[wl]
        comm = LibraryComm(rank, world, shm_name=name, slot_bytes=1 << 30)
        h, first, count = library_sharded_sector(synthetic_model(w), w.sector, comm)
        kind = comm.shard_info(h)[0]
        v = np.random.default_rng(5).standard_normal(h.dim)
        a, b, nd, _ = comm.tridiag(h, v[first * h.dim_up:(first + count) * h.dim_up], nlanc)
        ref = h.lanczos_tridiag(v, nlanc)[:2] if rank == 0 else None   # the single-GPU loop of the same handle
        h.destroy()
        comm.destroy()
        q.put((rank, kind, a, b, nd, ref, None))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, -1, None, None, 0, None, traceback.format_exc() + str(e)))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_recurrence_at_full_size(gpu, world):
    """BASELINE config 2 (Dim = 11 778 624) tridiagonalised by 2 and 3 ranks that share the GPU (shared-memory transport):
    transposed exchange on padded panels with the recurrence kept in that layout, ragged last rank (3432 rows and 215
    panels over 3 ranks).  Every rank's coefficients equal those of the single-GPU loop of the same sector, which the
    parity suite pins on the oracle at the sizes the oracle reaches (size-independent property: the sharding is not
    observable)."""
    nlanc = 12
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = f"edigpu_full_{os.getpid()}_{world}"
    procs = [ctx.Process(target=_full_rank_main, args=(r, world, name, "cfg2", nlanc, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[6] is None, r[6]
    ra, rb = [r[5] for r in res if r[5] is not None][0]
    for _, kind, a, b, nd, _, _ in res:
        assert kind == 2 and nd == nlanc
        assert rel_err(a, ra) < 1e-11 and rel_err(b, rb) < 1e-11


def test_library_shard_error_paths(gpu):
    import torch  # noqa: F401
    from edipack_amd import capi
    from edipack_amd.hamiltonian import SectorHamiltonian
    from edipack_amd.sharding import LibraryComm
    comm = LibraryComm(0, 1)
    _, pm, v = _reference("superc", "hybrid", 2, 3, 0)
    h = SectorHamiltonian.flat_from_model(pm, 0, row_first=3, row_count=10)      # not the shard of rank 0 of 1
    with pytest.raises(capi.EdigpuError):
        comm.apply(h, v[3:13])
    h.destroy()
    h = SectorHamiltonian.flat_from_model(pm, 0)
    with pytest.raises(capi.EdigpuError):
        comm.apply(h, v[:-1])                                                   # Nloc mismatch
    h.destroy()
    with pytest.raises(capi.EdigpuError):
        LibraryComm(2, 2, shm_name="edigpu_bad")                                # rank outside the world
    # a general g_ph(a,b) couples electronic states across rows: such phonon sectors stay on one GPU
    _, pm, _ = _reference("normal", "normal", 2, 2, (3, 2), phonon=True)
    pm.g_ph = np.array([[0.4, 0.2], [0.2, -0.3]])
    h = SectorHamiltonian.normal_from_model(pm, 3, 2)
    with pytest.raises(capi.EdigpuError, match="density couplings"):
        comm.apply(h, np.zeros(h.dim))
    h.destroy()
    comm.destroy()
