#!/bin/bash
# round 4, first look at the local-block kernels: parity (the impurity-block tests take them where they fit), then H*v times
set -o pipefail
tag=${1:-r4a}; shift
mkdir -p gpurun_out/$tag
EDIGPU_SB_VERBOSE=1 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "impurity_block_image" > gpurun_out/$tag/tests.log 2>&1
rc=$?
echo "sb tests rc=$rc: $(tail -1 gpurun_out/$tag/tests.log)"
[ $rc -ne 0 ] && { grep -m5 -n "Error\|error\|assert" gpurun_out/$tag/tests.log; exit 1; }
for w in "$@"; do
  for sb in 1 0; do
    EDIGPU_SB=$sb EDIGPU_SB_VERBOSE=1 EDIGPU_IB_MINROW=0 timeout -k 10 300 python scripts/probe_hv.py --workload $w --steps 10 --warmup 2 > gpurun_out/$tag/${w}_sb$sb.log 2>&1 || { echo "$w sb=$sb FAILED"; tail -5 gpurun_out/$tag/${w}_sb$sb.log; exit 1; }
    echo "sb=$sb $(tail -1 gpurun_out/$tag/${w}_sb$sb.log)"
  done
done
