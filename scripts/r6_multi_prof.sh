#!/bin/bash
# kernel times of the N > 1 step rehearsed on one GPU (RCCL world of one, collectives forced), block exchange with the
# recurrence in the panel layout; $1 = workload
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=${1:-cfg2}
O=$R/gpurun_out/r6_multi_prof_$W; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export EDIGPU_FORCE_MULTI=1 EDIGPU_FORCE_COLLECTIVES=1 EDIGPU_IB_MINROW=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --workload $W --steps 50 --warmup 5 --no-cpu > $O/bench.json 2> $O/bench.err
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print("%-100s %6s %9.1f us %6.2f%%" % (r['Name'][:100], r['Calls'], float(r['AverageNs'])/1000, float(r['Percentage'])))
PY
