#!/bin/bash
# tools/build_variant.sh NAME SOURCE.hip [-Dflags...]: a tuning build of the library under build_var/NAME.so in which ONE
# source file is compiled with extra flags (the other objects come from build_var/obj, i.e. from the last regular build).
# Load it with CF_LIB_PATH=build_var/NAME.so (cista_flow_amd/lib.py).
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p build_var/obj
python -c "import __graft_entry__ as g; g.compile_objects()" >/dev/null
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter "$@" -c cista_flow_amd/csrc/$src -o build_var/$name.${src%.hip}.o
objs=""
for f in conv_igemm conv_wino4 conv_wino_sk conv_wino1d conv_wino16 conv_wino_p pointwise metrics cf_api; do
  if [ "$f.hip" == "$src" ]; then objs="$objs build_var/$name.$f.o"; else objs="$objs build_var/obj/$f.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o build_var/$name.so $objs
echo build_var/$name.so
