#!/bin/bash
# round 3, first GPU pass of the impurity-block kernels: parity, then the two workloads that matter with and without it
set -o pipefail
mkdir -p gpurun_out/r3a
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k impurity_block > gpurun_out/r3a/ib_tests.log 2>&1
echo "ib tests rc=$?: $(tail -1 gpurun_out/r3a/ib_tests.log)"
for w in cfg3_ns16 cfg2 cfg3_ns15; do
  for ib in 1 0; do
    EDIGPU_IB=$ib timeout -k 10 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu --no-resident \
      > gpurun_out/r3a/bench_${w}_ib${ib}.json 2> gpurun_out/r3a/bench_${w}_ib${ib}.err
    echo "$w ib=$ib rc=$?: $(python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r3a/bench_${w}_ib${ib}.json"))
    print("it/s %.1f ms_step %.4f ms_hv %.4f frac %.3f image %s" % (d["value"], d["ms_per_step"], d["roofline"]["ms_per_launch"], d["roofline"]["frac"], d["config"]["image"]))
except Exception as e:
    print("no json", e)
PY
)"
  done
done
