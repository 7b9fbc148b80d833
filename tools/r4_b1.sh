#!/bin/bash
# B = 1 / 2 / 4 latency regime: knobs that change the launch structure (one box)
mkdir -p gpurun_out/r4
run() { # label, batch, env...
  local label=$1; shift; local b=$1; shift
  env "$@" timeout -k 10 300 python bench.py --batch $b --steps 60 --warmup 10 --repeat 1 --no-cpu-baseline --no-alt --no-roofline --no-latency > gpurun_out/r4/b1_tmp.log 2>&1
  echo "B=$b $label: $(grep -o '"value": [0-9.]*' gpurun_out/r4/b1_tmp.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4/b1_tmp.log | head -1)"
}
for b in 1 2 4; do
  run default $b CF_X=0
  run enc_pair $b CF_ENC_PAIR=1
  run enc_pair+inorm_fused $b CF_ENC_PAIR=1 CF_INORM_FUSED=1
  run serial $b CF_SERIAL=1
  run default_again $b CF_X=0
done
