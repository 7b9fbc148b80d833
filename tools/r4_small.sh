#!/bin/bash
# round 4: small-batch regime A/B of launcher thresholds: usage  bash tools/r4_small.sh "ENV=.. ENV=.." [batches] [pairs]
set -o pipefail
VAR=$1; BATCHES=${2:-"1 2 4"}; PAIRS=${3:-2}
mkdir -p gpurun_out/r4
for b in $BATCHES; do for i in $(seq 1 $PAIRS); do
  for v in base var; do
    if [ $v = var ]; then E="$VAR"; else E=""; fi
    r=$(env $E timeout -k 10 300 python bench.py --batch $b --steps 60 --warmup 10 --repeat 1 --no-cpu-baseline --no-alt --no-roofline --no-latency 2>gpurun_out/r4/small_err.log | grep -o '"value": [0-9.]*' | head -1)
    echo "B=$b run $i $v: $r"
  done
done; done
