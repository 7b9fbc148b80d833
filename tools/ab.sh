# same-box A/B of tuning builds (build_var/*.so through CF_LIB_PATH) against the in-tree library
LIBS=${LIBS:-"build_var/lib_prev.so"}
for i in 1 2; do
  for l in $LIBS; do CF_LIB_PATH=$l python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt > gpurun_out/ab_$(basename $l .so)_$i.log 2>&1; done
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt > gpurun_out/ab_tree_$i.log 2>&1
done
for f in gpurun_out/ab_*.log; do echo "$f $(grep -o '"value": [0-9.]*' $f | head -1)"; done
for l in $LIBS; do echo $l; CF_LIB_PATH=$l TILES=0,40 SHAPES=${SHAPES:-cista.D,cista.P,gates,out_gates,Gates} python tools/conv_bench.py 2>&1 | tail -5; done
echo tree; TILES=0,40 SHAPES=${SHAPES:-cista.D,cista.P,gates,out_gates,Gates} python tools/conv_bench.py 2>&1 | tail -5
