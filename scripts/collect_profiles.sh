#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box:  bash scripts/collect_profiles.sh r01 cfg2
# 1. kernel-trace + stats of the same command bench.py's roofline comes from (bench.py --no-cpu --no-resident: ONE
#    workload per summary -- the round-2 cfg2 summary mixed in the Ns=16 launches of the hbm_resident leg)
# 2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE, L2 hit/miss) on the plain H*v probe
# Raw output goes to gpurun_out/prof_<tag>/; summaries are written by scripts/summarize_profiles.py.
set -u
TAG=${1:-r01}; WL=${2:-cfg2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_${TAG}_${WL}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --workload $WL --steps 50 --warmup 5 --no-cpu --no-resident > $O/bench.json 2> $O/bench.err
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  n=$(echo $c | tr " " "_" | cut -c1-30)
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$n -- python3 $R/scripts/probe_hv.py --workload $WL --steps 5 --warmup 1 > $O/pmc_$n.log 2>&1
done
python3 $R/scripts/summarize_profiles.py $O $TAG $WL
