# tuning: conv_bench timings of the Winograd kernel with parts compiled out (-DWG_ABL bit mask: 1 transform, 2 U loads, 4 tail, 8 raw DMA);
# build first: for a in 1 2 4 8 15; do hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -DWG_ABL=$a -o build_var/lib_abl$a.so cista_flow_amd/csrc/*.hip; done
# results of ablated builds are WRONG by construction: timing only.  Run on the GPU box from the repo root.
for a in 0 1 2 4 8 15; do
  if [ $a = 0 ]; then unset CF_LIB_PATH; else export CF_LIB_PATH=build_var/lib_abl$a.so; fi
  echo "ablation $a"; TILES=40 SHAPES=cista.D,cista.P,gates python tools/conv_bench.py 2>&1 | grep -v "out_gates\|big\|hs" | tail -3
done
