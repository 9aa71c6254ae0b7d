#!/bin/bash
# rows kernel with its global traffic compiled out (Ns = 16): `git apply scripts/r3_abl_rows.patch`, build
# edipack_amd/lib/abl/libedigpu_abl{16,32,48}.so by hand with -DIB_ABL=16 (no loads of V) / 32 (no stores) / 48 (neither),
# `git checkout edipack_amd/csrc/kernels_ib.hip`, then run this on the GPU box.  Round 3: 1269 / 1086 / 1159 / 964 us.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/abl_rows; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for a in 0 16 32 48; do
  lib=$R/edipack_amd/lib/abl/libedigpu_abl$a.so; [ $a = 0 ] && lib=$R/edipack_amd/lib/libedigpu.so
  export EDIGPU_LIB=$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/a$a -- python3 $R/scripts/probe_hv.py --workload cfg3_ns16 --steps 10 --warmup 2 > $O/a$a.log 2>&1
done
