#!/bin/bash
# Late round 4: counters and kernel stats re-collected on the final sources (the traffic figures are stamped with a hash of
# csrc/, which the sharded-loop change moved), default bench line, one-rank rehearsal of the N > 1 step.
set -u
TAG=r04
mkdir -p gpurun_out/bench
for wl in "$@"; do
  timeout -k 10 500 bash scripts/collect_profiles.sh $TAG $wl > gpurun_out/collect_$wl.log 2>&1; echo "collect $wl rc=$?"; tail -2 gpurun_out/collect_$wl.log
done
