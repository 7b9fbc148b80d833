#!/bin/bash
# A/B of library builds on one GPU box: tools/ab_libs.sh name1 name2 ... (build_var/lib_<name>.so)
for v in "$@"; do
  CF_LIB_PATH=$PWD/build_var/lib_$v.so python bench.py --no-cpu-baseline --no-alt --no-roofline --steps 30 --warmup 5 > gpurun_out/ab_$v.log 2>&1
  echo "$v $(grep -o '"value": [0-9.]*' gpurun_out/ab_$v.log)"
done
