#!/bin/bash
# alternating A/B of libraries on one box: tools/r4_abn.sh "<lib paths, '-' = in-tree>" [pairs] [extra bench args]
LIBS=${1:-"- build_var/lib_prev.so"}; PAIRS=${2:-3}; EXTRA=${3:-}
mkdir -p gpurun_out/r4
for i in $(seq 1 $PAIRS); do
  for l in $LIBS; do
    if [ "$l" == "-" ]; then unset CF_LIB_PATH; else export CF_LIB_PATH=$PWD/$l; fi
    timeout -k 10 300 python bench.py --steps 40 --warmup 8 --repeat 1 --no-cpu-baseline --no-alt --no-roofline --no-latency $EXTRA > gpurun_out/r4/abn_tmp.log 2>&1 || { tail -20 gpurun_out/r4/abn_tmp.log; exit 1; }
    echo "$l run $i: $(grep -o '"value": [0-9.]*' gpurun_out/r4/abn_tmp.log | head -1)"
  done
done
