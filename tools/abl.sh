for a in 0 1 2 4 8 15; do
  if [ $a = 0 ]; then unset CF_LIB_PATH; else export CF_LIB_PATH=build_var/lib_abl$a.so; fi
  echo "ablation $a"; TILES=40 SHAPES=cista.D,cista.P,gates python tools/conv_bench.py 2>&1 | grep -v "out_gates\|big\|hs" | tail -3
done
