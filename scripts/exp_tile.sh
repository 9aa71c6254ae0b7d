#!/bin/bash
# experiment: panel sweep variants under rocprofv3 (kernel durations + SQ/TCC counters)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/exp_tile
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
for r in rows[:4]:
    print("   %-70s calls=%s avg_us=%.1f" % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
}
run tile72 EDIGPU_TILE_ROWS=72
run tile36 EDIGPU_TILE_ROWS=36
run tile152 EDIGPU_TILE_ROWS=152
run notile EDIGPU_PANEL_TILE=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
  n=$(echo $c | tr " " "_" | cut -c1-24)
  for v in 1 0; do
    EDIGPU_PANEL_TILE=$v rocprofv3 --pmc $c --output-format csv -d $O/pmc_${WL}_t${v}_$n -- python3 $R/scripts/probe_hv.py --workload $WL --steps 3 --warmup 1 > $O/pmc_${WL}_t${v}_$n.log 2>&1
    f=$(find $O/pmc_${WL}_t${v}_$n -name "*counter_collection.csv" | head -1)
    echo "== pmc $WL tile=$v"; python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
try:
    for r in csv.DictReader(open(sys.argv[1])):
        k=r['Kernel_Name']
        if 'panel' in k or 'tile' in k or 'rows_kernel' in k:
            acc[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,d in acc.items():
        print("  ",k)
        for c,v in d.items(): print("      %-32s %.4g" % (c, sum(v)/len(v)))
except Exception as e: print("   ERR",e)
PY
  done
done
