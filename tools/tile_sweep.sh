#!/bin/bash
# In-model tile sweep (GPU box): each line = bench.py with CF_TILE_OVERRIDE forcing tile kinds for named layers.
run() { name=$1; shift; CF_TILE_OVERRIDE="$1" python bench.py --no-cpu-baseline --no-alt --no-roofline --steps 30 --warmup 5 > gpurun_out/sweep_$name.log 2>&1; echo "$name [$1] $(grep -o '"value": [0-9.]*' gpurun_out/sweep_$name.log)"; }
run A ""
run B "cista.D=23,cista.Dg=23"
run C "cista.P=20,cista.P0=20"
run D "cista.gates=23,cista.Gates=23"
run E "cista.out_gates=26"
run F "cista.W0=23"
run G "cista.upsamp=4"
run H "cista.gates=25,cista.Gates=25,cista.out_gates=25"
run A2 ""
