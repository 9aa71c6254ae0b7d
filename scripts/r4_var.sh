#!/bin/bash
# kernel times of the plain product under several settings: scripts/r4_var.sh <workload> <tag> "name ENV=VAL ..." ...
w=$1; tag=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
for spec in "$@"; do
  set -- $spec; name=$1; shift
  out=$R/gpurun_out/$tag/$name; mkdir -p $out
  ( cd /tmp && export TMPDIR=/tmp && env EDIGPU_IB_MINROW=0 "$@" timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python3 $R/scripts/probe_hv.py --workload $w --steps 10 --warmup 2 > $out/probe.log 2> $out/probe.err ) || { echo "$name FAILED"; tail -3 $out/probe.err; continue; }
  f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$name" "$(tail -1 $out/probe.log)" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rows" in r["Name"] or "cols" in r["Name"] or "normal_" in r["Name"]]
s = "  ".join("%s %.1f" % (("rows" if "rows" in r["Name"] else "cols" if "cols" in r["Name"] else r["Name"][:24]), float(r["AverageNs"]) / 1e3) for r in rows[:3])
print("%-22s %s | %s" % (sys.argv[2], s, sys.argv[3].split("H*v")[-1].strip()))
PY
done
