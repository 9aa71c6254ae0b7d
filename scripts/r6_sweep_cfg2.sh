#!/bin/bash
# config 2, iterations per second of the fused step under the tile / row switches (one box, back to back)
run() { env "$@" python bench.py --no-cpu --no-resident --steps 400 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('%-40s' % '$*', round(d['value'],1), 'it/s  step', round(d['ms_per_step'],5), ' hv', round(d['roofline']['ms_per_launch'],5))"; }
run X=0
run EDIGPU_TILE_ROWS=16
run EDIGPU_TILE_ROWS=24
run EDIGPU_TILE_ROWS=40
run EDIGPU_TILE_ROWS=48
run EDIGPU_TILE_ROWS=64
run EDIGPU_LANCZOS_GRAPH=1 EDIGPU_LANCZOS_GRAPH_MAX=100000000
run EDIGPU_ROWS_TD=2
run EDIGPU_BLOCKED=0
run X=0
