#!/bin/bash
# one-rank rehearsal of the N > 1 step with the collectives forced (RCCL world of one): block exchange vs column-block exchange
for w in "$@"; do
  for g in 0 1; do
    extra=""; [ $g = 1 ] && extra="EDIGPU_SHARD_GENERIC=1"
    env EDIGPU_FORCE_MULTI=1 EDIGPU_FORCE_COLLECTIVES=1 EDIGPU_IB_MINROW=0 $extra timeout -k 10 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu > gpurun_out/r5_multi_${w}_g$g.json 2> gpurun_out/r5_multi_${w}_g$g.err || { echo "$w generic=$g FAILED"; tail -3 gpurun_out/r5_multi_${w}_g$g.err; continue; }
    python - $w $g gpurun_out/r5_multi_${w}_g$g.json <<'PY'
import json, sys
d = json.load(open(sys.argv[3]))
print(sys.argv[1], "generic" if sys.argv[2] == "1" else "block  ", "ms/step %.4f" % d["ms_per_step"], "exchange_ms", d["config"]["exchange_ms_per_step"], "bytes", d["config"]["exchange_bytes_per_rank_per_hv"])
PY
  done
done
