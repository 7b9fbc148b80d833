#!/bin/bash
# build_var/lib_stamp.so: the library with the Winograd kernels' cycle stamps compiled in (-DCF_STAMP on conv_igemm.hip, conv_wino_p.hip,
# conv_wino16.hip, conv_wino1d.hip); the other objects come from the last regular build.  CF_LIB_PATH=build_var/lib_stamp.so python tools/stamp_probe.py
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.compile_objects()" >/dev/null
objs=""
for f in conv_igemm conv_wino4 conv_wino_sk conv_wino1d conv_wino16 conv_wino_p pointwise metrics cf_api; do
  case $f in
    conv_igemm|conv_wino_p|conv_wino16|conv_wino1d)
      hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter -DCF_STAMP -c cista_flow_amd/csrc/$f.hip -o build_var/stamp.$f.o &
      objs="$objs build_var/stamp.$f.o";;
    *) objs="$objs build_var/obj/$f.o";;
  esac
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o build_var/lib_stamp.so $objs
echo build_var/lib_stamp.so
