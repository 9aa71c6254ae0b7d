#!/bin/bash
# One GPU-box call that refreshes the round's evidence: GPU tests, one bench line per workload, rocprofv3 summaries.
#   bash scripts/refresh_evidence.sh r01   (results under gpurun_out/; copy what should be judged into profiles/)
set -u
TAG=${1:-r01}
mkdir -p gpurun_out/bench
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
for wl in cfg2 cfg1 cfg3 cfg3_ns15 cfg3_ns16 cfg4 cfg4_ns12 cfg5 cfg5_stored cfg5_stored_ns11; do
  timeout -k 10 600 python bench.py --workload $wl > gpurun_out/bench/${TAG}_bench_$wl.json 2> gpurun_out/bench/${TAG}_bench_$wl.err || { echo "bench $wl failed"; tail -5 gpurun_out/bench/${TAG}_bench_$wl.err; exit 1; }
  python - $wl gpurun_out/bench/${TAG}_bench_$wl.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], round(d["value"], 1), "it/s  hv_ms", round(d["roofline"]["ms_per_launch"], 4), "frac", round(d["roofline"]["frac"], 3),
      "cpu", d.get("cpu_baseline", {}).get("value"))
PY
done
bash scripts/collect_profiles.sh $TAG cfg2 > gpurun_out/collect_cfg2.log 2>&1 && tail -25 gpurun_out/collect_cfg2.log
