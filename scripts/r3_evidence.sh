#!/bin/bash
# round-3 evidence, part by part (a gpurun call is limited to 20 minutes):
#   bash scripts/r3_evidence.sh prof cfg2 cfg3_ns15 ...   rocprofv3 stats + counters per workload (collect_profiles.sh)
#   bash scripts/r3_evidence.sh bench cfg2 cfg1 ...       one bench line per workload (bench_lines.sh)
what=$1; shift
case $what in
  prof) for w in "$@"; do bash scripts/collect_profiles.sh r03 $w > gpurun_out/collect_$w.log 2>&1; echo "$w: $(tail -1 gpurun_out/collect_$w.log)"; done;;
  bench) bash scripts/bench_lines.sh r03 "$@";;
esac
