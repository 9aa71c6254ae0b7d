#!/bin/bash
# ib parity, then the plain product and the Lanczos step of the three ladder sectors
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "impurity_block or split_rows" > gpurun_out/ib_tests.log 2>&1 || { tail -5 gpurun_out/ib_tests.log; exit 1; }
tail -1 gpurun_out/ib_tests.log
for w in cfg3_ns15 cfg3_ns16 cfg3_ns17; do
  timeout -k 10 300 python scripts/probe_hv.py --workload $w --steps 30 --warmup 5 || exit 1
  timeout -k 10 300 python scripts/probe_hv.py --workload $w --steps 30 --warmup 5 --lanczos || exit 1
done
