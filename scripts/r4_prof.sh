#!/bin/bash
# per-kernel times of the plain product of a workload: scripts/r4_prof.sh <workload> <tag> [ENV=VAL ...]
set -o pipefail
w=$1; tag=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python3 $R/scripts/probe_hv.py --workload $w --steps 10 --warmup 2 > $out/probe.log 2> $out/probe.err
cd $R
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:6]:
    print("  %-70s calls %5s avg_us %10.1f pct %5s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
tail -1 $out/probe.log
