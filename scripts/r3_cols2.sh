#!/bin/bash
# A/B of the two columns kernels: parity with the pipelined form forced, then probe times with it off / on
set -o pipefail
mkdir -p gpurun_out/cols2
EDIGPU_IB_COLS2=1 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k impurity_block > gpurun_out/cols2/ib_tests.log 2>&1
rc=$?
echo "ib tests (cols2) rc=$rc: $(tail -1 gpurun_out/cols2/ib_tests.log)"
[ $rc -ne 0 ] && exit $rc
for w in cfg3_ns16 cfg3_ns15; do
  for c in 0 1; do
    echo "== $w COLS2=$c"
    EDIGPU_IB_COLS2=$c timeout -k 10 300 python scripts/probe_hv.py --workload $w --steps 30 --warmup 5 || exit 1
    EDIGPU_IB_COLS2=$c timeout -k 10 300 python scripts/probe_hv.py --workload $w --steps 30 --warmup 5 --lanczos || exit 1
  done
done
