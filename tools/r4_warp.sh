#!/bin/bash
# warp_kernel with 2-D pixel tiles (in-tree) against the strip version (build_var/lib_prev.so): parity, cold / warm time, fabric bytes
mkdir -p gpurun_out/r4
true
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for l in prev tree; do
  if [ $l == prev ]; then export CF_LIB_PATH=$R/build_var/lib_prev.so; else unset CF_LIB_PATH; fi
  echo "== $l"; (cd $R && python tools/warp_probe.py 2>&1 | grep -v amdgpu.ids)
  for c in FETCH_SIZE WRITE_SIZE; do
    PYTHONPATH=$R rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/r4/warp_${l}_$c -- python $R/tools/warp_probe.py > $R/gpurun_out/r4/warp_${l}_$c.log 2>&1
    python - "$R/gpurun_out/r4/warp_${l}_$c" $c <<'PY'
import csv, glob, sys
v = [float(r["Counter_Value"]) for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "warp_kernel" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]]
print("   %s per warp_kernel launch: mean %.1f (n=%d)%s" % (sys.argv[2], sum(v) / max(len(v), 1), len(v), "  [KB; FETCH_SIZE x2 for the gfx950 correction]" if sys.argv[2] == "FETCH_SIZE" else " [KB]"))
PY
  done
done
find $R/gpurun_out/r4 -name "*counter_collection.csv" -delete
