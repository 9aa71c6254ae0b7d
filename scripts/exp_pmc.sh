#!/bin/bash
# usage: exp_pmc.sh WL "counters" "name ENV=.. ENV=.." ... : per-kernel mean counter values of probe_hv under rocprofv3 --pmc
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/exp_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
WL=$1; shift
CTR=$1; shift
for spec in "$@"; do
  set -- $spec
  name=$1; shift
  n=$(echo $CTR | tr " " "_" | cut -c1-20)
  env "$@" rocprofv3 --pmc $CTR --output-format csv -d $O/${WL}_${name}_$n -- python3 $R/scripts/probe_hv.py --workload $WL --steps 3 --warmup 1 > $O/${WL}_${name}_$n.log 2>&1
  f=$(find $O/${WL}_${name}_$n -name "*counter_collection.csv" | head -1)
  echo "== $WL $name"; python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name']
    if 'panel' in k or 'tile' in k or 'rows_kernel' in k or 'ib_' in k or 'sb_' in k:
        acc[k[:48]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in acc.items():
    print("  ",k, "  ".join("%s=%.4g" % (c, sum(v)/len(v)) for c,v in d.items()))
PY
done
