#!/bin/bash
# kernel times of the N > 1 step rehearsed on one GPU (RCCL world of one, collectives forced)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/multi_prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export EDIGPU_FORCE_MULTI=1 EDIGPU_FORCE_COLLECTIVES=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --workload ${1:-cfg2} --steps 50 --warmup 5 > $O/bench.json 2> $O/bench.err
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print("%-100s %6s %9.1f us %6.2f%%" % (r['Name'][:100], r['Calls'], float(r['AverageNs'])/1000, float(r['Percentage'])))
PY
