#!/bin/bash
# One bench line per workload into gpurun_out/bench/<tag>_bench_<workload>.json (run on the GPU box):
#   bash scripts/bench_lines.sh r02 [workloads...]
TAG=${1:-r02}; shift
WLS=${@:-cfg2 cfg1 cfg3 cfg3_ns15 cfg3_ns16 cfg4 cfg4_ns12 cfg5 cfg5_stored cfg5_stored_ns11}
mkdir -p gpurun_out/bench
for wl in $WLS; do
  timeout -k 10 600 python bench.py --workload $wl > gpurun_out/bench/${TAG}_bench_$wl.json 2> gpurun_out/bench/${TAG}_bench_$wl.err || { echo "bench $wl failed"; tail -5 gpurun_out/bench/${TAG}_bench_$wl.err; continue; }
  python - $wl gpurun_out/bench/${TAG}_bench_$wl.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
r = d["roofline"]
print(sys.argv[1], round(d["value"], 1), "it/s  hv_ms", round(r["ms_per_launch"], 4), "frac", r["frac"], "traffic_frac", r.get("traffic_frac"),
      "cpu", d.get("cpu_baseline", {}).get("value"), "resident", (d["config"].get("hbm_resident") or {}).get("frac"))
PY
done
