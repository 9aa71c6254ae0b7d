#!/bin/bash
# Ns = 17 product with variants of the split rows kernel (edipack_amd/lib/abl/*.so, built by hand)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in main B C D; do
  lib=$R/edipack_amd/lib/abl/libedigpu_$v.so
  [ $v = main ] && lib=$R/edipack_amd/lib/libedigpu.so
  [ -f $lib ] || continue
  echo "== $v"
  EDIGPU_LIB=$lib timeout -k 10 300 python scripts/probe_hv.py --workload cfg3_ns17 --steps 10 --warmup 2 || exit 1
done
