for v in head eb2_w4 eb2_w3 eb1_w4 eb4_w3 head eb2_w4; do
  CF_LIB_PATH=$PWD/build_var/lib_$v.so python bench.py --no-cpu-baseline --no-alt --no-roofline --steps 30 --warmup 5 > gpurun_out/ab_$v.log 2>&1
  echo "$v $(grep -o '"value": [0-9.]*' gpurun_out/ab_$v.log)"
done
