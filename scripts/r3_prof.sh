#!/bin/bash
# per-kernel times of a workload: scripts/r3_prof.sh <workload> <tag> [ENV=VAL ...]
set -o pipefail
w=$1; tag=$2; shift 2
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o p -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 10 --warmup 2 --no-cpu --no-resident > $GRAFT_REPO_ROOT/$out/bench.json 2> $GRAFT_REPO_ROOT/$out/bench.err
cd $GRAFT_REPO_ROOT
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-90s calls %5s avg_us %10.1f pct %5s" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
