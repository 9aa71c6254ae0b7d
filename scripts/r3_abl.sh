#!/bin/bash
# kernel times of the ib kernels under compile-time ablations (edipack_amd/lib/abl/*.so, built by hand with -DIB_ABL=n)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/abl_cols
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in 0 1; do
for a in 0 1 2 4 7; do
  lib=$R/edipack_amd/lib/abl/libedigpu_abl$a.so
  [ $a = 0 ] && lib=$R/edipack_amd/lib/libedigpu.so
  export EDIGPU_LIB=$lib EDIGPU_IB_COLS2=$c
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/c${c}_a$a -- python3 $R/scripts/probe_hv.py --workload cfg3_ns16 --steps 10 --warmup 2 > $O/c${c}_a$a.log 2>&1
  f=$(find $O/c${c}_a$a -name "*kernel_stats.csv" | head -1)
  echo "== cols2=$c abl=$a: $(grep -E 'ib_(cols|rows)' $f | awk -F'","' '{printf "%s %.1f us | ", substr($1,15,28), $4/1000}')"
done
done
