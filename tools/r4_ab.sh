#!/bin/bash
# round-4 A/B of the in-tree library against build_var/lib_prev.so on ONE box: parity tests first, then per-layer timings, then alternating step runs
# usage: bash tools/r4_ab.sh "<pytest -k expression>" "<TILES>" "<SHAPES>" [pairs] [extra bench args]
set -o pipefail
K=${1:-winograd}; TILES=${2:-0,47}; SHAPES=${3:-convc2,fh.conv1,layer2,menc,layer3,convf2}; PAIRS=${4:-3}; EXTRA=${5:-}
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -x -q -k "$K" > gpurun_out/r4/ops.log 2>&1 || { tail -30 gpurun_out/r4/ops.log; exit 1; }
tail -2 gpurun_out/r4/ops.log
echo "== prev"; CF_LIB_PATH=build_var/lib_prev.so TILES=$TILES SHAPES=$SHAPES timeout -k 10 300 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/conv_prev.log
echo "== tree"; TILES=$TILES SHAPES=$SHAPES timeout -k 10 300 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/conv_tree.log
for i in $(seq 1 $PAIRS); do
  CF_LIB_PATH=build_var/lib_prev.so timeout -k 10 300 python bench.py --steps 40 --warmup 8 --repeat 1 --no-cpu-baseline --no-alt --no-roofline --no-latency $EXTRA > gpurun_out/r4/ab_prev_$i.log 2>&1 || { tail -20 gpurun_out/r4/ab_prev_$i.log; exit 1; }
  echo "prev run $i: $(grep -o '"value": [0-9.]*' gpurun_out/r4/ab_prev_$i.log | head -1)"
  timeout -k 10 300 python bench.py --steps 40 --warmup 8 --repeat 1 --no-cpu-baseline --no-alt --no-roofline --no-latency $EXTRA > gpurun_out/r4/ab_tree_$i.log 2>&1 || { tail -20 gpurun_out/r4/ab_tree_$i.log; exit 1; }
  echo "tree run $i: $(grep -o '"value": [0-9.]*' gpurun_out/r4/ab_tree_$i.log | head -1)"
done
