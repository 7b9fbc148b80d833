#!/bin/bash
# tools/build_flagged.sh NAME "-Dflags" src1.hip src2.hip ...: build_var/NAME.so with the named sources compiled with the extra flags
# (in parallel), the other objects from the last regular build.  CF_LIB_PATH=build_var/NAME.so selects it.
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2; shift 2
python -c "import __graft_entry__ as g; g.compile_objects()" >/dev/null
objs=""
for f in conv_igemm conv_wino4 conv_wino_sk conv_wino1d conv_wino16 conv_wino_p pointwise metrics cf_api; do
  if [[ " $* " == *" $f.hip "* ]]; then
    hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter $flags -c cista_flow_amd/csrc/$f.hip -o build_var/$name.$f.o &
    objs="$objs build_var/$name.$f.o"
  else
    objs="$objs build_var/obj/$f.o"
  fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o build_var/$name.so $objs
echo build_var/$name.so
