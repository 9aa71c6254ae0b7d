#!/bin/bash
# rows-kernel geometry sweep on one workload: scripts/r3_nt.sh <workload> <tag> nt...
w=$1; tag=$2; shift 2
for nt in "$@"; do
  echo "== NT=$nt"
  bash scripts/r3_prof.sh $w $tag/nt$nt EDIGPU_IB=1 EDIGPU_IB_NT=$nt | grep "ib_"
done
