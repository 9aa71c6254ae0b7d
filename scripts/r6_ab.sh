#!/bin/bash
# A/B of library builds on config 2: scripts/r6_ab.sh NAME... (edipack_amd/lib/ab/libedigpu_NAME.so), two rounds each
for i in 1 2; do
  for n in "$@"; do
    EDIGPU_LIB=edipack_amd/lib/ab/libedigpu_$n.so python bench.py --no-cpu --no-resident --steps 400 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$n',round(d['value'],1),round(d['ms_per_step'],5),round(d['roofline']['ms_per_launch'],5))"
  done
done
