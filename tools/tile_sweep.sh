#!/bin/bash
# In-model tile sweep (GPU box): each line = bench.py with CF_TILE_OVERRIDE forcing tile kinds for named layers.
run() { name=$1; shift; CF_TILE_OVERRIDE="$1" python bench.py --no-cpu-baseline --no-alt --no-roofline --steps 30 --warmup 5 > gpurun_out/sweep_$name.log 2>&1; echo "$name $(grep -o '"value": [0-9.]*' gpurun_out/sweep_$name.log | head -1) [$1]" | cut -c1-150; }
L1=""
for e in enet fnet cnet; do for b in 0 1; do for c in 1 2; do L1="$L1,event_flownet.$e.layer1.$b.conv$c=23"; done; done; done
L1=${L1#,}
run A ""
run B "cista.gates=28,cista.Gates=28"
run C "cista.gates=23,cista.Gates=23"
run D "cista.D=23,cista.Dg=23"
run E "$L1"
run F "cista.gates=28,cista.Gates=28,cista.D=23,cista.Dg=23,$L1"
run A2 ""
