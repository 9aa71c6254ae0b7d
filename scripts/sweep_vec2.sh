#!/bin/bash
# One-column vs two-column panel kernel (kernel B of the normal-mode product) on the normal-mode workloads, library-built
# and hand-over images:  gpurun -- 'bash scripts/sweep_vec2.sh'   (prints Lanczos it/s, H*v us per launch pair, plain H*v us)
set -e
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['ms_per_launch']*1e3,1), round(d['config']['hv_only_ms']*1e3,1))"; }
for wl in cfg3 cfg3_ns15 cfg3_ns16 cfg2; do for v in 0 1; do
echo "$wl vec2=$v"; EDIGPU_PANEL_VEC2=$v python bench.py --no-cpu --steps 100 --workload $wl | pr
done; done
echo "cfg2 handover"; for v in 0 1; do EDIGPU_PANEL_VEC2=$v EDIGPU_NORMAL_EXPLICIT=1 python bench.py --no-cpu --steps 100 --workload cfg2 | pr; done
