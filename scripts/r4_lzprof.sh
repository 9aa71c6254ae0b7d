#!/bin/bash
# per-kernel times of the fused Lanczos step: scripts/r4_lzprof.sh <workload> <tag> [ENV=VAL ...]
w=$1; tag=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag; mkdir -p $out
( cd /tmp && export TMPDIR=/tmp && env EDIGPU_IB_MINROW=0 "$@" timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python3 $R/scripts/probe_hv.py --workload $w --steps 10 --warmup 2 --lanczos > $out/probe.log 2> $out/probe.err )
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:7]:
    print("  %-90s calls %5s avg_us %10.1f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
tail -1 $out/probe.log
