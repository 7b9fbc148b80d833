#!/bin/bash
# round-4 check of the persistent Winograd kernel (tile 48): parity tests, per-layer timings against tile 40, step A/B
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -k "winograd or inorm_stats" > gpurun_out/r4/ops.log 2>&1 || { tail -30 gpurun_out/r4/ops.log; exit 1; }
tail -3 gpurun_out/r4/ops.log
TILES=40,48,49 SHAPES=cista.D,cista.P,gates,out_gates,Gates,layer1,hs.gates,hs.P,big.P timeout -k 10 300 python tools/conv_bench.py > gpurun_out/r4/conv.log 2>&1 || { tail -30 gpurun_out/r4/conv.log; exit 1; }
cat gpurun_out/r4/conv.log
BATCH=4 TILES=40,48,49 SHAPES=cista.D,cista.P,gates,out_gates,Gates timeout -k 10 300 python tools/conv_bench.py > gpurun_out/r4/conv_b4.log 2>&1 || { tail -30 gpurun_out/r4/conv_b4.log; exit 1; }
cat gpurun_out/r4/conv_b4.log
for i in 1 2 3; do
  for v in 0 2; do
    CF_WINOP=$v timeout -k 10 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-alt --no-roofline > gpurun_out/r4/ab_winop${v}_$i.log 2>&1 || { tail -20 gpurun_out/r4/ab_winop${v}_$i.log; exit 1; }
    echo "CF_WINOP=$v run $i: $(grep -o '"value": [0-9.]*' gpurun_out/r4/ab_winop${v}_$i.log | head -1)"
  done
done
