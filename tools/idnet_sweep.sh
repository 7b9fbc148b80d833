run() { name=$1; shift; CF_TILE_OVERRIDE="$1" CF_LAYER_REPORT=gpurun_out/ids_$name.txt python bench.py --no-cpu-baseline --no-alt --model idnet --batch 16 --height 260 --width 346 > gpurun_out/ids_$name.log 2>&1; echo "$name [$1] $(grep -o '"value": [0-9.]*' gpurun_out/ids_$name.log | head -1) $(grep -E 'idn.gru.q|idn.gru.zr' gpurun_out/ids_$name.txt | awk '{printf "%s t%s %sus %sTF; ", $1,$3,$9,$11}')"; }
run A ""
run B "idn.gru.q=22"
run C "idn.gru.q=29"
run D "idn.gru.q=20"
run E "idn.gru.q=22,idn.gru.zr=22"
run F "idn.gru.q=29,idn.gru.zr=29"
