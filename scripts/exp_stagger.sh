#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/exp_stagger
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
WL=${1:-cfg2}
run() { # name, env...
  local name=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_${WL}_$name -- python3 $R/scripts/probe_hv.py --workload $WL --steps 20 --warmup 3 > $O/st_${WL}_$name.log 2>&1
  f=$(find $O/st_${WL}_$name -name "*kernel_stats.csv" | head -1)
  echo "== $WL $name"; python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:2]:
    print("   %-70s calls=%s avg_us=%.1f" % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
}
for t in 36 24; do for s in 0 1 2 4 8; do run t${t}_s$s EDIGPU_TILE_ROWS=$t EDIGPU_STAGGER=$s; done; done
