# the other BASELINE workloads + phase split (GPU box); writes gpurun_out/r4w/*.log
mkdir -p gpurun_out/r4w
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --model eraft > gpurun_out/r4w/eraft.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --model idnet --batch 16 --height 260 --width 346 > gpurun_out/r4w/idnet_f32.log 2>&1
CF_PRECISION=f16 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-alt --model idnet --batch 16 --height 260 --width 346 > gpurun_out/r4w/idnet_f16.log 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency --batch 4 --height 480 --width 640 > gpurun_out/r4w/eiflow_480x640_b4.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency > gpurun_out/r4w/eiflow.log 2>&1
CF_PHASES=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-alt --no-roofline > gpurun_out/r4w/phases.log 2>&1
for b in 1 2 4; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-alt --no-roofline --batch $b > gpurun_out/r4w/eiflow_b$b.log 2>&1; done
for f in gpurun_out/r4w/*.log; do echo "$f $(grep -o '"value": [0-9.]*' $f | head -2 | tr '\n' ' ')"; done
grep -i "phase" gpurun_out/r4w/phases.log | tail -5
