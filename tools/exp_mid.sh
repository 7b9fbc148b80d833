for v in base mid; do echo "== $v"; CF_LIB_PATH=$PWD/build_var/lib_$v.so SHAPES=gru,convc2,fh.conv1,layer1,cista,gates TILES=20,22,23,26 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids; done
bash tools/ab_libs.sh base mid base mid
