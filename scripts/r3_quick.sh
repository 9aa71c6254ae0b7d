#!/bin/bash
# quick look: H*v probe times of the ib kernels per workload (rocprof kernel stats), after the ib parity tests
set -o pipefail
tag=${1:-r3q}; shift
mkdir -p gpurun_out/$tag
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k impurity_block > gpurun_out/$tag/ib_tests.log 2>&1
echo "ib tests rc=$?: $(tail -1 gpurun_out/$tag/ib_tests.log)"
for w in "$@"; do
  bash scripts/r3_prof.sh $w $tag/$w EDIGPU_IB=1 | grep "ib_\|finalize"
  python - <<PY
import json
d=json.load(open("gpurun_out/$tag/$w/bench.json"))
print("$w: it/s %.1f ms_step %.4f ms_hv %.4f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["ms_per_launch"], d["roofline"]["frac"]))
PY
done
