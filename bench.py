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
    """HBM bytes per H*v launch from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json)."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(p):
        return None
    try:
        return json.load(open(p)).get(workload, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


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


def run_single(args):
    import torch  # noqa: F401  (first: one HIP runtime per process, see edipack_amd/capi.py)
    from edipack_amd import capi
    from edipack_amd.synthetic import WORKLOADS, build_workload

    w = WORKLOADS[args.workload]
    capi.init(0)
    t0 = time.perf_counter()
    h = build_workload(w)
    t_build = time.perf_counter() - t0
    bytes_hv, bytes_step = h.algorithmic_bytes()
    ms_step, ms_hv = h.lanczos_bench(args.warmup, args.steps)
    ms_hv_only = h.time_apply(max(2, args.warmup), args.steps, lanczos=False)
    achieved = bytes_hv / (ms_hv * 1e-3) / 1e9
    out = {
        "metric": "Lanczos H*v iterations/sec, largest (Nup,Ndw) sector",
        "value": 1e3 / ms_step, "unit": "it/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "c128" if h.is_complex else "f64", "data": "synthetic",
        "config": {"workload": f"{w.name}: {w.ed_mode} mode, bath={w.bath_type}, Norb={w.norb}, Nbath={w.nbath}, "
                               f"sector={w.sector}, Dim={h.dim} ({w.note})",
                   "storage": {0: "Kronecker (Hd,Hup,Hdw,Hnd)", 1: "flat CSR", 2: "direct (on-the-fly)"}[h.kind],
                   "parallelism": "1 GPU, device-resident Lanczos", "build_s": round(t_build, 3),
                   "hv_only_ms": ms_hv_only, "lanczos_step_GBs": bytes_step / (ms_step * 1e-3) / 1e9},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": _traffic_from_profiles(w.name),
                     "kernel": {0: "normal_rows_kernel + normal_dw_panel2_kernel (normal_dw_panel_kernel for odd DimUp / small sectors)", 1: "sell_rows_packed_kernel (SELL-64 + value dictionary; csr_rows_kernel fallback)",
                                2: "direct_rows_kernel"}[h.kind],
                     "algorithmic_bytes_per_launch": bytes_hv, "ms_per_launch": ms_hv},
    }
    if not args.no_cpu and h.kind != 2:
        out["cpu_baseline"] = cpu_baseline(h, w.name, args.cpu_seconds)
    h.destroy()
    print(json.dumps(out), flush=True)


def run_multi(args):
    import torch
    import torch.distributed as dist
    from edipack_amd import capi
    from edipack_amd.sharding import gpu_sharded_hamiltonian, gpu_transposed_hamiltonian
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
    # one rank per GPU; EDIGPU_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsal of the
    # N>1 data flow on a single-GPU box; the timed configuration is always nccl = RCCL over xGMI)
    backend = os.environ.get("EDIGPU_DIST_BACKEND", "nccl")
    dev = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(backend)
    capi.init(dev)
    w = WORKLOADS[args.workload]
    model = synthetic_model(w)
    exchange = "allgather"
    if w.ed_mode == "normal" and os.environ.get("EDIGPU_EXCHANGE", "transpose") != "allgather":
        try:
            plan, h, lz = gpu_transposed_hamiltonian(model, w.sector, world, rank, stage_host=backend != "nccl")
            exchange = "transpose"
        except RuntimeError:      # sector not servable this way (explicit spH0nd, phonons)
            pass
    if exchange == "allgather":
        plan, h, lz = gpu_sharded_hamiltonian(model, w.sector, world, rank, direct=w.direct)
    bytes_hv, _ = h.algorithmic_bytes()   # this shard's share of the algorithmic bytes
    if exchange == "transpose":
        bytes_hv /= world                 # every rank holds the whole sector's (small) tables
    sent = lz.exchange_bytes if exchange == "transpose" else 8 * plan.chunk * (world - 1) * (2 if lz.dtype.is_complex else 1)
    gen = torch.Generator(device="cuda").manual_seed(12345 + rank)
    v0 = torch.randn(plan.nloc, dtype=torch.float64, device="cuda", generator=gen)
    if lz.dtype.is_complex:
        v0 = v0.to(lz.dtype)
    lz.tridiag(v0, max(1, args.warmup))
    # timed region: K Lanczos steps, barrier + synchronize on both sides
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    # transposed exchange: the product comes fused with the vector updates and the one all-reduce of the step, and
    # the loop runs on pre-bound launches -- no per-product events there, the step time is the product time
    fused = bool(getattr(lz, "fused", False))
    counter = {"k": 0}
    if not fused:
        hv_orig = lz.hv

        def hv_timed(*a):
            k = counter["k"]
            ev[k][0].record()
            hv_orig(*a)
            ev[k][1].record()
            counter["k"] = k + 1

        lz.hv = hv_timed
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lz.tridiag(v0, args.steps)
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    ms_hv = dt * 1e3 / args.steps if fused else sum(a.elapsed_time(b) for a, b in ev) / args.steps
    t = torch.tensor([dt, ms_hv, bytes_hv], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    tmax = t.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tsum = t.clone()
    dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    if rank == 0:
        dt_max, ms_hv_max, bytes_total = float(tmax[0]), float(tmax[1]), float(tsum[2])
        ms_step = dt_max * 1e3 / args.steps
        achieved = bytes_total / (ms_hv_max * 1e-3) / 1e9
        out = {
            "metric": "Lanczos H*v iterations/sec, largest (Nup,Ndw) sector",
            "value": 1e3 / ms_step, "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "c128" if lz.dtype.is_complex else "f64", "data": "synthetic",
            "config": {"workload": f"{w.name}: {w.ed_mode} mode, bath={w.bath_type}, Norb={w.norb}, "
                                   f"Nbath={w.nbath}, sector={w.sector}, Dim={h.dim} ({w.note})",
                       "parallelism": (f"row-sharded over {world} GPUs, transposed exchange: two RCCL all-to-alls per "
                                       f"H*v, the first overlapped with the row half (Hd + Hup)"
                                       if exchange == "transpose" else
                                       f"row-sharded over {world} GPUs, RCCL all-gather of v overlapped with the "
                                       f"shard-local part of H*v"),
                       "exchange_bytes_per_rank_per_hv": int(sent)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                         "frac": achieved / (HBM_PEAK_GBS * world), "traffic": None,
                         "note": "whole-job: sum of shard algorithmic bytes / slowest rank's H*v time "
                                 "(exchange included)", "ms_per_launch": ms_hv_max},
        }
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    h.destroy()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()
    # EDIGPU_FORCE_MULTI=1: take the N > 1 code path with a single rank (RCCL world of one; with
    # EDIGPU_FORCE_COLLECTIVES=1 the collectives are issued too) -- a one-GPU rehearsal of the nccl calls
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        # a scaling script that forgets the launcher must not record one-GPU numbers as N-GPU ones
        sys.exit(f"bench.py: --gpus {args.gpus} needs one rank per GPU (WORLD_SIZE={world}): launch it as\n"
                 f"  python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 "
                 f"--master-port 29533 bench.py --gpus {args.gpus} --steps {args.steps} --warmup {args.warmup}")
    if args.gpus > 1 or world > 1 or os.environ.get("EDIGPU_FORCE_MULTI"):
        run_multi(args)
    else:
        run_single(args)


if __name__ == "__main__":
    main()
