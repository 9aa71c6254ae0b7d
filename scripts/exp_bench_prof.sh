#!/bin/bash
# usage: exp_bench_prof.sh WL "name ENV=.." ... : kernel durations (stats pass) and FETCH/WRITE/TCC counters (pmc passes) of
# bench.py's Lanczos loop per variant
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/exp_bp
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
WL=$1; shift
for spec in "$@"; do
  set -- $spec
  name=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_${WL}_$name -- python3 $R/bench.py --workload $WL --steps 20 --warmup 3 --no-cpu --no-resident > $O/st_${WL}_$name.log 2>&1
  f=$(find $O/st_${WL}_$name -name "*kernel_stats.csv" | head -1)
  echo "== $WL $name"; python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:5]:
    print("   %-70s calls=%s avg_us=%.1f" % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
  for CTR in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $CTR | tr " " "_" | cut -c1-20)
    env "$@" rocprofv3 --pmc $CTR --output-format csv -d $O/${WL}_${name}_$n -- python3 $R/bench.py --workload $WL --steps 4 --warmup 1 --no-cpu --no-resident > $O/${WL}_${name}_$n.log 2>&1
    f=$(find $O/${WL}_${name}_$n -name "*counter_collection.csv" | head -1)
    python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name']
    if 'panel' in k or 'tile' in k or 'rows_kernel' in k or 'blk' in k:
        acc[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in acc.items():
    print("     ",k, "  ".join("%s=%.4g (n=%d)" % (c, sum(v)/len(v), len(v)) for c,v in d.items()))
PY
  done
done
