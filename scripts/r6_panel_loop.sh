#!/bin/bash
# Round 4 (late): the sharded recurrence kept in the padded panel layout (EDIGPU_SHARD_PANEL_LOOP, default on) against the
# row-layout loop with a conversion around every product (=0), one-rank rehearsal with the collectives forced through RCCL.
mkdir -p gpurun_out
for w in "$@"; do
  for pl in 1 0; do
    env EDIGPU_FORCE_MULTI=1 EDIGPU_FORCE_COLLECTIVES=1 EDIGPU_IB_MINROW=0 EDIGPU_SHARD_PANEL_LOOP=$pl timeout -k 10 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu > gpurun_out/r6_multi_${w}_pl$pl.json 2> gpurun_out/r6_multi_${w}_pl$pl.err || { echo "$w panel_loop=$pl FAILED"; tail -3 gpurun_out/r6_multi_${w}_pl$pl.err; continue; }
    python - $w $pl gpurun_out/r6_multi_${w}_pl$pl.json <<'PY'
import json, sys
d = json.load(open(sys.argv[3]))
print(sys.argv[1], "panel loop" if sys.argv[2] == "1" else "row loop  ", "ms/step %.4f" % d["ms_per_step"], "exchange_ms", d["config"]["exchange_ms_per_step"], "bytes", d["config"]["exchange_bytes_per_rank_per_hv"])
PY
  done
done
