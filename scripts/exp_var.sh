#!/bin/bash
# usage: exp_var.sh WL "name ENV=.. ENV=.." "name2 ..." : kernel durations of probe_hv under rocprofv3 per variant
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/exp_var
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
WL=$1; shift
for spec in "$@"; do
  set -- $spec
  name=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_${WL}_$name -- python3 $R/scripts/probe_hv.py --workload $WL --steps 20 --warmup 3 ${PROBE_ARGS:-} > $O/st_${WL}_$name.log 2>&1
  f=$(find $O/st_${WL}_$name -name "*kernel_stats.csv" | head -1)
  echo "== $WL $name"; python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:3]:
    print("   %-70s calls=%s avg_us=%.1f" % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
