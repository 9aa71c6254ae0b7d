#!/usr/bin/env python3
"""bench.py -- Lanczos H*v throughput of the MI355X engine on BASELINE.json's workload.

    python bench.py --gpus N --steps K --warmup W [--workload cfg2]

One "step" = one Lanczos iteration (H*v + three-term recurrence) on the largest sector of the
workload, vectors resident in HBM.  N=1: single-shard device-resident loop inside libedigpu.so.
N>1 (launched by torch.distributed.run, one rank per GPU): the vector is row-sharded; normal mode uses the
transposed exchange (two RCCL all-to-alls per product, the first overlapped with the row half of H*v;
EDIGPU_EXCHANGE=allgather selects the all-gather form), the flat modes an all-gather overlapped with the
shard-local block; total work is fixed (strong scaling).  Rank 0 prints ONE JSON line.

roofline  : algorithmic bytes of one H*v in the reference's storage format (SURVEY.md 8d,
            edigpu_algorithmic_bytes) / average H*v launch duration from HIP events recorded
            around every launch of the timed steps.
cpu_baseline (N=1 only): the CPU oracle (a port of the reference algorithm, NOT the reference
            binary) timed on a bounded sample of the same matrices: the reference's MPI row
            decomposition on all host cores (OpenMP threads; the reported value) and the serial loop.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def _traffic_from_profiles(workload: str):
    """HBM bytes per H*v launch from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json, written by
    scripts/collect_profiles.sh with the gfx950 corrections of profiles/r02_fetch_calibration.txt).  The file is
    stamped with a hash of the kernel sources it was measured on: for any other build the figure is stale -> None."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(p):
        return None
    try:
        from edipack_amd import capi
        rec = json.load(open(p)).get(workload, {})
        if rec.get("source_hash") != capi.kernel_source_hash():
            return None
        return rec.get("hbm_bytes_per_launch")
    except Exception:
        return None


def _hbm_resident(steps: int):
    """The sector the 0.60 target is phrased on with a working set beyond the 256 MiB Infinity Cache: the 3-orbital
    hybrid structure at Ns=16 (Dim 165 636 900, 1.33 GB per vector).  Plain H*v, HIP events."""
    from edipack_amd.synthetic import WORKLOADS, build_workload
    w = WORKLOADS["cfg3_ns16"]
    h = build_workload(w)
    try:
        bytes_hv, _ = h.algorithmic_bytes()
        n = max(5, min(steps, 20))
        ms = h.time_apply(2, n, lanczos=2)            # the product as the Lanczos loops compute it (as roofline.*)
        ms_boundary = h.time_apply(2, n, lanczos=0)   # edigpu_apply_dev: vectors in the reference's layout
        tr = _traffic_from_profiles(w.name)
        return {"workload": f"{w.name}: 3 orbitals, hybrid bath, Ns=16, sector {w.sector}, Dim={h.dim}",
                "ms_hv": ms, "achieved": bytes_hv / (ms * 1e-3) / 1e9, "unit": "GB/s",
                "frac": bytes_hv / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": bytes_hv,
                "ms_hv_reference_layout": ms_boundary, "traffic": tr,
                "traffic_frac": (tr / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if tr else None}
    finally:
        h.destroy()


def _host_cores() -> int:
    """CPU cores this process may actually use: the cgroup quota when there is one (a GPU box hands out a
    share of the host, e.g. 16 of 256 hardware threads), else the affinity mask."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return n


def cpu_baseline(h, workload, budget_s: float = 15.0):
    """Time the oracle's spMatVec restatement on the SAME matrices (exported from the handle): single
    thread in the reference's serial loop order, and the reference's MPI row decomposition with one OpenMP
    thread per host core (the reported value; `cores` = threads used)."""
    import numpy as np
    from oracle import oracle as O
    rng = np.random.default_rng(12345)
    cores = _host_cores()
    if h.kind == 0:
        hd, up, dw, nd = h.export_normal()
        ndarg = nd if nd[0][-1] > 0 else None
        v = rng.standard_normal(h.dim)
        v /= np.linalg.norm(v)
        hv = np.empty_like(v)

        def one():
            O.normal_matvec_arrays(h.dim_up, h.dim_dw, hd, up, dw, ndarg, v, hv)

        def one_mt():
            O.normal_matvec_arrays_mt(h.dim_up, h.dim_dw, hd, up, dw, ndarg, v, hv, cores)
    else:
        rp, col, val = h.export_csr()
        rp = np.ascontiguousarray(rp, dtype=np.int64)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.complex128)
        v = (rng.standard_normal(h.dim) + 1j * rng.standard_normal(h.dim)).astype(np.complex128)
        v /= np.linalg.norm(v)
        y = np.empty_like(v)

        def one():
            O.csr_matvec(rp, col, val, v)

        def one_mt():
            O.csr_matvec_z_mt(rp, col, val, v, y, cores)

    def rate(fn, budget):
        t0 = time.perf_counter()
        fn()
        t1 = time.perf_counter() - t0
        n = int(max(2, min(200, budget / max(t1, 1e-6))))
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        return n, (time.perf_counter() - t0) / n

    n1, dt1 = rate(one, 0.4 * budget_s)
    nm, dtm = rate(one_mt, 0.6 * budget_s)
    return {"value": 1.0 / dtm, "unit": "H*v/s", "cores": cores, "kind": "port",
            "single_thread_value": 1.0 / dt1,
            "sample": f"{nm} H*v products of workload {workload} (same matrices as the GPU run) with the oracle's C "
                      f"restatement of the reference's MPI row decomposition on {cores} OpenMP threads "
                      f"({dtm * 1e3:.1f} ms each); serial loop order, 1 thread: {n1} products, {dt1 * 1e3:.1f} ms each"}


def cpu_baseline_direct(w, budget_s: float = 15.0):
    """On-the-fly workload (nonsu2): the oracle's restatement of directMatVec_nonsu2_main
    (ED_NONSU2/ED_HAMILTONIAN_NONSU2_DIRECT_HxV.f90:22-126) -- elements regenerated row by row, one binary search of the
    sector map per element -- on the SAME sector and model as the GPU run.  A whole product of config 5 is ~10 M rows x
    ~60 searched elements; the bounded sample is a leading range of rows, timed on one thread and on all host cores
    (row ranges, as the reference's MPI ranks take them), scaled to the sector's rows."""
    import numpy as np
    from oracle import oracle as O
    from edipack_amd.synthetic import synthetic_model
    pm = synthetic_model(w)
    om = O.Model(ed_mode=pm.ed_mode, bath_type=pm.bath_type, norb=pm.norb, nbath=pm.nbath, nspin=pm.nspin, hfmode=pm.hfmode,
                 xmu=pm.xmu, uloc=tuple(float(u) for u in pm.uloc), ust=pm.ust, jh=pm.jh, jx=pm.jx, jp=pm.jp,
                 hloc=pm.hloc, be=pm.be, bv=pm.bv, bd=pm.bd, bu=pm.bu)
    d = O.DirectNonsu2(om, w.sector)
    rng = np.random.default_rng(12345)
    v = (rng.standard_normal(d.dim) + 1j * rng.standard_normal(d.dim)).astype(np.complex128)
    v /= np.linalg.norm(v)
    hv = np.zeros_like(v)
    cores = _host_cores()

    def timed(rows, threads):
        t0 = time.perf_counter()
        d.matvec(v, hv, threads=threads, rows=rows)
        return time.perf_counter() - t0

    probe = min(d.dim, 20000)
    t1 = timed(probe, 1)
    rows1 = int(min(d.dim, max(probe, probe * 0.4 * budget_s / max(t1, 1e-6))))
    dt1 = timed(rows1, 1) * d.dim / rows1
    tm = timed(min(d.dim, probe * cores), cores)
    rowsm = int(min(d.dim, max(probe * cores, probe * cores * 0.6 * budget_s / max(tm, 1e-6))))
    dtm = timed(rowsm, cores) * d.dim / rowsm
    return {"value": 1.0 / dtm, "unit": "H*v/s", "cores": cores, "kind": "port", "single_thread_value": 1.0 / dt1,
            "sample": f"rows 1..{rowsm} of the {d.dim} of workload {w.name} (same sector and model as the GPU run) through the "
                      f"oracle's C restatement of directMatVec_nonsu2_main (elements regenerated on the fly, a binary search "
                      f"per element) on {cores} threads over row ranges, scaled by rows: {dtm * 1e3:.0f} ms per product; "
                      f"1 thread, rows 1..{rows1}: {dt1 * 1e3:.0f} ms per product"}


def run_single(args):
    import torch  # noqa: F401  (first: one HIP runtime per process, see edipack_amd/capi.py)
    from edipack_amd import capi
    from edipack_amd.synthetic import WORKLOADS, build_workload

    w = WORKLOADS[args.workload]
    capi.init(0)
    t0 = time.perf_counter()
    h = build_workload(w, handover=args.image == "handover")
    t_build = time.perf_counter() - t0
    bytes_hv, bytes_step = h.algorithmic_bytes()
    # untimed: one short run first, so that a 20-step timed region does not start on idle clocks and first-touch
    # allocations (the driver's 20 x 0.18 ms region read 10 % below the 200-step figure in round 2); the timed region
    # below is still exactly `warmup` untimed + `steps` timed steps
    h.lanczos_bench(5, 40)
    ms_step, ms_hv = h.lanczos_bench(args.warmup, args.steps)
    ms_hv_only = h.time_apply(max(2, args.warmup), args.steps, lanczos=False)   # boundary product, reference layout
    achieved = bytes_hv / (ms_hv * 1e-3) / 1e9
    traffic = _traffic_from_profiles(w.name)
    rd, cp, tr = capi.membw(1 << 30)
    out = {
        "metric": "Lanczos H*v iterations/sec, largest (Nup,Ndw) sector",
        "value": 1e3 / ms_step, "unit": "it/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "c128" if h.is_complex else "f64", "data": "synthetic",
        "config": {"workload": f"{w.name}: {w.ed_mode} mode, bath={w.bath_type}, Norb={w.norb}, Nbath={w.nbath}, "
                               f"sector={w.sector}, Dim={h.dim} ({w.note})",
                   "storage": {0: "Kronecker (Hd,Hup,Hdw,Hnd)", 1: "flat CSR", 2: "direct (on-the-fly)"}[h.kind],
                   "image": args.image + (" (factored=%d, Hnd terms=%d, classes=%d, panel=%d, panel-major W=%d)" % h.image_info()[:5]
                                          if h.kind == 0 else ""),
                   "parallelism": "1 GPU, device-resident Lanczos", "build_s": round(t_build, 3),
                   "hv_only_ms": ms_hv_only, "lanczos_step_GBs": bytes_step / (ms_step * 1e-3) / 1e9},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     # counter-based rate: what actually crossed the fabric per launch / launch duration (for the flat
                     # modes the algorithmic figure exceeds it: their device format moves ~3x fewer bytes)
                     "traffic_GBs": (traffic / (ms_hv * 1e-3) / 1e9) if traffic else None,
                     "traffic_frac": (traffic / (ms_hv * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                     "measured_ceiling_GBs": {"read": rd, "copy": cp, "triad": tr,
                                              "note": "streaming kernels on 1 GiB buffers, this device, this run"},
                     "kernel": {0: ("sb_rows_kernel + sb_cols_kernel (local blocks of 5 levels on padded 16-column panels; the "
                                    "Lanczos step of `value` runs the same two kernels, fused, or with the new vector as its own pass in "
                                    "the 768-thread geometry)"
                                    if h.kind == 0 and h.image_info()[5] == 3 else
                                    "normal_rows_kernel in position order + sb_cols_kernel (short rows, EDIGPU_POSROWS=1: padded 16-column panels)"
                                    if h.kind == 0 and h.image_info()[5] == 5 else
                                    "sb_rows_kernel x 2 (rows staged in halves, EDIGPU_SB_SPLIT=1) + ib_cols_kernel"
                                    if h.kind == 0 and h.image_info()[5] == 4 else
                                    "ib_rows_kernel + ib_cols_kernel (impurity-block image, padded 16-column panels)"
                                    if h.kind == 0 and h.image_info()[5] == 1 else
                                    "ib_rows_kernel x 2 (rows staged in halves) + ib_cols_kernel (impurity-block image)"
                                    if h.kind == 0 and h.image_info()[5] == 2 else
                                    "normal_rows_kernel + normal_dw_tile_kernel (sectors of >= 2M rows; normal_dw_panel_kernel below)"),
                                1: "sell_rows_packed_kernel (SELL-64 + value dictionary; csr_rows_kernel fallback)",
                                2: "direct_rows_kernel"}[h.kind],
                     "algorithmic_bytes_per_launch": bytes_hv, "ms_per_launch": ms_hv},
    }
    if h.kind == 1:
        # superc / nonsu2 STORED: the device image (SELL-64 + value dictionary) moves a fraction of the bytes the
        # reference's CSR format holds, so the algorithmic figure overstates the use of the memory system (it can exceed
        # the peak).  roofline.frac is therefore the COUNTER-based fraction where a counter pass of this build exists
        # (null otherwise); the algorithmic one stays as frac_reference_format.  On the fly (kind 2) the kernel fetches
        # MORE than the algorithmic bytes (scattered gathers): there frac stays the algorithmic fraction -- a fraction
        # that rises when a kernel wastes traffic is not a roofline fraction -- and traffic_frac stands beside it.
        r = out["roofline"]
        r["frac_reference_format"] = r["frac"]
        r["frac"] = r["traffic_frac"]
    if not args.no_cpu and h.kind != 2:
        out["cpu_baseline"] = cpu_baseline(h, w.name, args.cpu_seconds)
    dim_gpu = h.dim
    h.destroy()
    if not args.no_cpu and out.get("cpu_baseline") is None:
        out["cpu_baseline"] = cpu_baseline_direct(w, args.cpu_seconds)
    if args.workload == "cfg2" and not args.no_resident:
        out["config"]["hbm_resident"] = _hbm_resident(args.steps)
    print(json.dumps(out), flush=True)


def run_multi(args):
    """N > 1: one rank per GPU.  torch.distributed is only the rendezvous (it hands the 128-byte RCCL id of rank 0 to
    the other ranks and reduces the timings at the end); the communicator, the exchange and the whole sharded
    recurrence live in libedigpu.so (csrc/edigpu_shard.hip: edigpu_comm_create + edigpu_lanczos_bench_sharded) --
    the same entry points the Fortran + MPI host binds (fortran/edigpu_shim.f90)."""
    import torch
    import torch.distributed as dist
    from edipack_amd import capi
    from edipack_amd.sharding import LibraryComm, library_sharded_sector
    from edipack_amd.synthetic import WORKLOADS, synthetic_model

    # RCCL prints a version banner on stdout when the communicator comes up: keep the real stdout for the one
    # JSON line and send everything else (from any library, any rank) to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", str(rank))
    os.environ.setdefault("WORLD_SIZE", str(world))
    local = int(os.environ.get("LOCAL_RANK", rank))
    # EDIGPU_DIST_BACKEND=gloo: several ranks share one GPU through the library's shared-memory transport (a
    # rehearsal of the N > 1 data flow on a one-GPU box; the timed configuration is RCCL over xGMI)
    backend = os.environ.get("EDIGPU_DIST_BACKEND", "nccl")
    dev = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo")          # rendezvous and scalar reductions only: no GPU traffic goes through it
    capi.init(dev)
    w = WORKLOADS[args.workload]
    model = synthetic_model(w)
    comm = None
    if backend == "nccl":
        # every rank must end up on the same transport: agree on the outcome over the rendezvous group
        err = ""
        try:
            ids = [LibraryComm.unique_id() if rank == 0 else None]
        except Exception as e:        # RCCL missing / not loadable on rank 0
            ids, err = [None], str(e)
        dist.broadcast_object_list(ids, src=0)
        if ids[0] is not None:
            try:
                comm = LibraryComm(rank, world, unique_id=ids[0])
            except Exception as e:
                err = str(e)
        ok = torch.tensor([1.0 if comm is not None else 0.0])
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok[0]) == 0.0:
            # LOUD fallback, visible in the JSON line ("transport"): the host-staged shared-memory transport of the
            # library still runs the N > 1 loop on N GPUs of one node, far below xGMI rates
            print(f"bench.py rank {rank}: RCCL communicator not available ({err or 'another rank failed'}); "
                  f"falling back to the shared-memory transport", file=sys.stderr, flush=True)
            if comm is not None:
                comm.destroy()
            comm, backend = None, "shm-fallback"
    if comm is None:
        comm = LibraryComm(rank, world, shm_name=f"edigpu_bench_{os.environ['MASTER_PORT']}", slot_bytes=1 << 30)
    exchange = os.environ.get("EDIGPU_EXCHANGE", "auto")
    # Short rows (config 2: 27 KB): a single GPU keeps such sectors on the generic kernels (the block ROWS kernel loses to
    # the generic one there), but on shards the block image wins -- its exchange moves the padded panels as they are and
    # the recurrence stays in that layout (0.465 against 0.58 ms per step in the one-rank rehearsal;
    # INTEGRATION.md recommends the same switch for -D_MPI hosts).  Set before the sector is built; a user's value wins.
    os.environ.setdefault("EDIGPU_IB_MINROW", "0")
    h, first, count = library_sharded_sector(model, w.sector, comm, direct=w.direct, exchange=exchange)
    transposed = False
    if h.kind == 0 and h.nloc == h.dim and exchange != "allgather":
        try:                      # the library takes the transposed exchange exactly when this query succeeds
            h.transpose_halo()
            transposed = True
        except capi.EdigpuError:
            pass
    bytes_hv, _ = h.algorithmic_bytes()   # a shard handle: this shard's share; a whole-sector handle: all of it
    if transposed:
        bytes_hv /= world
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ms_step, sent = comm.bench(h, args.warmup, args.steps)     # barrier + stream sync on both sides of the K steps
    torch.cuda.synchronize()
    dist.barrier()
    wall = time.perf_counter() - t0
    try:   # after the timed region, collective on every rank
        rccl_ranks, ms_exchange = comm.exchange_bench(h, max(5, min(args.steps, 50)))
    except Exception as e:   # never lose the measured line over the diagnostics
        print(f"bench.py rank {rank}: exchange_bench failed: {e}", file=sys.stderr, flush=True)
        rccl_ranks, ms_exchange = -2, None
    t = torch.tensor([ms_step, wall], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    tb = torch.tensor([bytes_hv], dtype=torch.float64)
    dist.all_reduce(tb, op=dist.ReduceOp.SUM)
    if rank == 0:
        ms = float(t[0])
        achieved = float(tb[0]) / (ms * 1e-3) / 1e9
        out = {
            "metric": "Lanczos H*v iterations/sec, largest (Nup,Ndw) sector",
            "value": 1e3 / ms, "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "c128" if h.is_complex else "f64", "data": "synthetic",
            "config": {"workload": f"{w.name}: {w.ed_mode} mode, bath={w.bath_type}, Norb={w.norb}, "
                                   f"Nbath={w.nbath}, sector={w.sector}, Dim={h.dim} ({w.note})",
                       "parallelism": (f"row-sharded over {world} GPUs, in-library loop, transposed exchange: two RCCL "
                                       f"all-to-alls per H*v, the first beside the row half (Hd + Hup)"
                                       if transposed else
                                       f"row-sharded over {world} GPUs, in-library loop, RCCL all-gather of v beside the "
                                       f"shard-local part of H*v"),
                       "exchange": {0: "all-gather", 1: "transposed, column blocks with halo columns",
                                    2: "transposed, padded panels (no packing; recurrence kept in the panel layout"
                                       + (")" if os.environ.get("EDIGPU_SHARD_PANEL_LOOP", "1") != "0" else
                                          " SWITCHED OFF: row loop)")}.get(comm.shard_info(h)[0], "?"),
                       "transport": {"nccl": "rccl", "shm-fallback": "shared memory, host-staged (RCCL NOT AVAILABLE: "
                                     "not an xGMI measurement)"}.get(backend, "shared memory (one-GPU rehearsal)"),
                       "exchange_bytes_per_rank_per_hv": int(sent),
                       # what RCCL itself reports (ncclCommCount; 0 = not an RCCL communicator) and what one step's
                       # collectives cost with nothing else running (rank 0's figure)
                       "rccl_ranks": int(rccl_ranks), "exchange_ms_per_step": ms_exchange},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                         "frac": achieved / (HBM_PEAK_GBS * world), "traffic": None,
                         "note": "whole-job: algorithmic bytes of one H*v / slowest rank's time per Lanczos step "
                                 "(exchange and vector updates included)", "ms_per_launch": ms},
        }
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    h.destroy()
    comm.destroy()
    dist.destroy_process_group()


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh processes, one rank per GPU, with the environment
    torch.distributed.run would give them, forward rank 0's JSON line and return the worst exit code.  This process has
    not touched the GPU (nothing above imports torch or loads libedigpu.so), and the children are new interpreters: no
    exec of a process that holds a HIP context."""
    import socket
    import subprocess
    n = args.gpus
    with socket.socket() as sk:            # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if rank == 0 else sys.stderr.fileno()))
    out0, _ = procs[0].communicate()
    worst = procs[0].returncode
    for p in procs[1:]:
        try:
            p.wait(timeout=600 if worst == 0 else 20)
        except subprocess.TimeoutExpired:
            p.kill()          # (the exact process started above)
            p.wait()
        worst = worst or p.returncode
    lines = [ln for ln in out0.decode(errors="replace").splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    elif worst == 0:
        worst = 1
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--image", choices=["library", "handover"], default="library",
                    help="normal mode: sector built by the library from the model (default) or created from the "
                         "reference's explicit arrays through edigpu_normal_create (the INTEGRATION.md section 2 patch)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-resident", action="store_true", help="skip the HBM-resident (Ns=16 ladder) H*v measurement")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()
    # EDIGPU_FORCE_MULTI=1: take the N > 1 code path with a single rank (RCCL world of one; with
    # EDIGPU_FORCE_COLLECTIVES=1 the collectives are issued too) -- a one-GPU rehearsal of the nccl calls
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        # a launcher that started a different number of ranks must not record its numbers as N-GPU ones
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU (or none: bench.py then "
                 f"starts its own)")
    if args.gpus > 1 or world > 1 or os.environ.get("EDIGPU_FORCE_MULTI"):
        run_multi(args)
    else:
        run_single(args)


if __name__ == "__main__":
    main()
