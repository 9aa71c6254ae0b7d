#!/usr/bin/env python3
"""Per-rank kernel times of the transposed-exchange Lanczos step on ONE GPU (no collectives): what a rank of an
N-GPU run computes per step, for the byte-count model in DESIGN.md section 5.

    python scripts/probe_transpose.py [--workload cfg2] [--worlds 2,4,8]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--worlds", default="2,4,8")
    ap.add_argument("--reps", type=int, default=50)
    args = ap.parse_args()
    import torch
    from edipack_amd import capi
    from edipack_amd.hamiltonian import SectorHamiltonian
    from edipack_amd.sharding import ShardPlan
    from tests.torch_sharded_loop import TransposedKernels, TransposedLanczos
    from edipack_amd.synthetic import WORKLOADS, synthetic_model
    capi.init(0)
    w = WORKLOADS[args.workload]
    h = SectorHamiltonian.normal_from_model(synthetic_model(w), *w.sector)
    out = {"workload": w.name, "dim": h.dim}
    for world in [int(x) for x in args.worlds.split(",")]:
        plan = ShardPlan(units=h.dim_dw, unit_len=h.dim_up, world=world, rank=0)
        lz = TransposedLanczos(plan, TransposedKernels(h, plan), vec_ops=None)
        lz.plan = plan
        lz.vin.normal_()
        lz.recv.normal_()
        ab = torch.tensor([0.5, 4.0], dtype=torch.float64, device="cuda")
        out2 = torch.zeros(2, dtype=torch.float64, device="cuda")
        steps = {
            "rotate_pack": lambda: lz.k.rotate_pack(lz, False, lz.vin, lz.vout, ab, lz.send),
            "rows": lambda: lz.k.rows(lz, lz.vin, lz.tmp),
            "cols": lambda: lz.k.cols(lz, lz.recv, lz.hvc),
            "unpack_add_dot2": lambda: lz.k.unpack_add_dot2(lz, lz.vin, lz.vout, lz.tmp, lz.back, out2),
        }
        res = {}
        for name, fn in steps.items():
            for _ in range(5):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res[name + "_us"] = round(e0.elapsed_time(e1) * 1e3 / args.reps, 2)
        res["sum_us"] = round(sum(res.values()), 2)
        res["sent_bytes_per_step"] = lz.exchange_bytes
        res["halo"] = lz.halo
        out[f"N={world}"] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
