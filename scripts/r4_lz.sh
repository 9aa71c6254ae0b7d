#!/bin/bash
# Lanczos step + product times of a workload: scripts/r4_lz.sh <workload> [ENV=VAL ...]
w=$1; shift
env EDIGPU_IB_MINROW=0 "$@" timeout -k 10 300 python scripts/probe_hv.py --workload $w --steps 10 --warmup 2 --lanczos 2>&1 | tail -1
