#!/bin/bash
# The GPU parity suite under every documented switch of the library ("Environment switches"):
#   gpurun -- 'bash scripts/check_switches.sh'            (all: ~100 s per switch, more than one gpurun call allows)
#   gpurun -- 'bash scripts/check_switches.sh 12'         (from the 13th switch on)
#   gpurun -- 'bash scripts/check_switches.sh 12 8'       (switches 13 .. 20)
SKIP=${1:-0}; COUNT=${2:-1000}; N=0
for sw in EDIGPU_LANCZOS_UNFUSED EDIGPU_NORMAL_EXPLICIT EDIGPU_FLAT_HOSTBUILD EDIGPU_LANCZOS_EXACTBETA EDIGPU_TRL_ONEPASS \
          EDIGPU_ELL_UNTYPED EDIGPU_CSR_NOSELL EDIGPU_CSR_UNPACKED EDIGPU_DIRECT_TERMORDER EDIGPU_PANEL_VEC2_MIN \
          "EDIGPU_ROW_SPLIT=2" "EDIGPU_PANEL_VEC2=0" EDIGPU_ND_IN_ROWS "EDIGPU_PANEL_TILE=0" "EDIGPU_TILE_ROWS=64" \
          EDIGPU_TILE_PERSIST "EDIGPU_HANDOVER_FACTOR=0" EDIGPU_ND_NO_MERGE EDIGPU_CMPLX_FOURPRODUCTS EDIGPU_DIRECT_NOSORT EDIGPU_LANCZOS_GRAPH EDIGPU_LANCZOS_INKERNEL_FINALIZE EDIGPU_EIGH_TWOPASS "EDIGPU_BLOCKED=1 EDIGPU_BLOCKED_MIN=0" \
          "EDIGPU_BLOCKED=1 EDIGPU_BLOCKED_MIN=0 EDIGPU_BLOCKED_W=16 EDIGPU_BLOCKED_LDS_KB=4" \
          "EDIGPU_IB=1 EDIGPU_IB_MIN=0" "EDIGPU_IB=1 EDIGPU_IB_MIN=0 EDIGPU_IB_SPLIT=1" \
          "EDIGPU_IB=1 EDIGPU_IB_MIN=0 EDIGPU_IB_COLS2=1" "EDIGPU_IB=1 EDIGPU_IB_MIN=0 EDIGPU_IB_ROWS=16" EDIGPU_TRL_FULL \
          "EDIGPU_IB=1 EDIGPU_IB_MIN=0 EDIGPU_SB=0" "EDIGPU_IB=1 EDIGPU_IB_MIN=0 EDIGPU_SB_STEP=0" "EDIGPU_IB=1 EDIGPU_IB_MIN=0 EDIGPU_SB_STEP=2" \
          "EDIGPU_IB=1 EDIGPU_IB_MIN=0 EDIGPU_SB_CW=1" "EDIGPU_IB=1 EDIGPU_IB_MIN=0 EDIGPU_SB_AMODE=1" \
          "EDIGPU_IB=1 EDIGPU_IB_MIN=0 EDIGPU_IB_SPLIT=1 EDIGPU_SB_SPLIT=1" "EDIGPU_IB=1 EDIGPU_IB_MIN=0 EDIGPU_IB_PAIRS=1" \
          "EDIGPU_IB=1 EDIGPU_IB_MIN=0 EDIGPU_POSROWS=1" "EDIGPU_IB=1 EDIGPU_IB_MIN=0 EDIGPU_IB_PSPAD=48"; do
  N=$((N+1)); [ $N -le $SKIP ] && continue
  [ $N -gt $((SKIP+COUNT)) ] && break
  case $sw in *=*) kv=$sw;; *) kv=$sw=1;; esac
  tag=$(echo "${kv}" | tr " =" "__")
  env $kv timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/sw_${tag}.log 2>&1
  echo "$kv: $(tail -1 gpurun_out/sw_${tag}.log)"
done
