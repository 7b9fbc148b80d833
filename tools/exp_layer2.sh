L2=""
for e in enet fnet cnet; do for n in layer2.0.conv1 layer2.0.conv2 layer2.0.downsample.0 layer2.1.conv1 layer2.1.conv2; do L2="$L2,event_flownet.$e.$n=23"; done; done
L2=${L2#,}
for i in 1 2; do
CF_TILE_OVERRIDE="$L2" CF_PHASES=1 python bench.py --no-cpu-baseline --no-alt --no-roofline --steps 30 --warmup 5 > gpurun_out/l2a.log 2>&1; echo "t23 $(grep -o '"value": [0-9.]*' gpurun_out/l2a.log) $(grep -a phases gpurun_out/l2a.log | tail -1 | cut -c30-100)"
CF_PHASES=1 python bench.py --no-cpu-baseline --no-alt --no-roofline --steps 30 --warmup 5 > gpurun_out/l2b.log 2>&1; echo "t34 $(grep -o '"value": [0-9.]*' gpurun_out/l2b.log) $(grep -a phases gpurun_out/l2b.log | tail -1 | cut -c30-100)"
done
