#!/bin/bash
# A/B builds of the local-block kernels: scripts/ab_build.sh NAME [-DFLAG=..]...  -> edipack_amd/lib/ab/libedigpu_NAME.so
# (the other objects are taken from the regular build; run a variant with EDIGPU_LIB=edipack_amd/lib/ab/libedigpu_NAME.so)
set -e
name=$1; shift
R=$(cd $(dirname $0)/.. && pwd)
C=$R/edipack_amd/csrc; O=$R/edipack_amd/lib/ab/$name; mkdir -p $O
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=off"
for f in kernels_sb kernels_sb2 kernels_sb3; do
  /opt/rocm/bin/hipcc $FL "$@" -c $C/$f.hip -o $O/$f.o &
done
wait
objs=$(ls $R/edipack_amd/lib/obj/*.o | grep -v "kernels_sb.hip.o\|kernels_sb2\|kernels_sb3")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/edipack_amd/lib/ab/libedigpu_$name.so $objs $O/kernels_sb.o $O/kernels_sb2.o $O/kernels_sb3.o -ldl -lrt
echo built $name
