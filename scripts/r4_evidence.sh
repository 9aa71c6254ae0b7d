#!/bin/bash
# Round-4 evidence in one GPU-box call (results under gpurun_out/; scripts/publish_profiles.py r04 copies them into profiles/):
#   bench lines, rocprofv3 kernel stats + traffic counters per workload, SQ counters of the two local-block kernels,
#   the one-rank rehearsal of the N > 1 step.
set -u
TAG=r04
bash scripts/bench_lines.sh $TAG cfg2 cfg1 cfg3 cfg3_ns15 cfg3_ns16 cfg3_ns17 cfg4 cfg4_ns12 cfg5 cfg5_stored cfg5_stored_ns11 2>&1 | tail -12
python bench.py --workload cfg2 --image handover > gpurun_out/bench/${TAG}_bench_cfg2_handover.json 2> gpurun_out/bench/${TAG}_bench_cfg2_handover.err
for wl in cfg2 cfg3_ns15 cfg3_ns16; do
  bash scripts/collect_profiles.sh $TAG $wl > gpurun_out/collect_$wl.log 2>&1; tail -3 gpurun_out/collect_$wl.log
done
bash scripts/exp_pmc.sh cfg3_ns16 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD" "sb EDIGPU_SB=1" > gpurun_out/${TAG}_sq1.txt 2>&1
bash scripts/exp_pmc.sh cfg3_ns16 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES" "sb EDIGPU_SB=1" > gpurun_out/${TAG}_sq2.txt 2>&1
bash scripts/exp_pmc.sh cfg3_ns16 "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "sb EDIGPU_SB=1" > gpurun_out/${TAG}_sq3.txt 2>&1
cat gpurun_out/${TAG}_sq1.txt gpurun_out/${TAG}_sq2.txt gpurun_out/${TAG}_sq3.txt | grep -A2 "== cfg3" > gpurun_out/${TAG}_cfg3_ns16_sq_counters.txt
bash scripts/r5_multi.sh cfg2 cfg3_ns15 cfg3_ns16 > gpurun_out/${TAG}_multi_one_rank.txt 2>&1; cat gpurun_out/${TAG}_multi_one_rank.txt
EDIGPU_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_gpus2_gloo.json 2> gpurun_out/${TAG}_bench_gpus2_gloo.err; echo "self-launched 2 ranks rc=$?"
