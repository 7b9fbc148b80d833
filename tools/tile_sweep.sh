#!/bin/bash
# In-model tile sweep (GPU box): each line = bench.py with CF_TILE_OVERRIDE forcing tile kinds for named layers.
run() { name=$1; shift; CF_TILE_OVERRIDE="$1" CF_LAYER_REPORT=gpurun_out/sweep_$name.txt python bench.py --no-cpu-baseline --no-alt --steps 30 --warmup 5 > gpurun_out/sweep_$name.log 2>&1; echo "$name [$1] $(grep -o '"value": [0-9.]*' gpurun_out/sweep_$name.log | head -1) | $(grep -E 'cista\.(P|out_gates|D|gates) ' gpurun_out/sweep_$name.txt | awk '{printf "%s t%s %s; ", $1,$3,$9}')"; }
run A ""
run B "cista.P=23,cista.P0=23,cista.out_gates=23"
run C "cista.P=26,cista.P0=26,cista.out_gates=26"
run D "cista.P=25,cista.P0=25,cista.out_gates=25"
run A2 ""
