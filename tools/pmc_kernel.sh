#!/bin/bash
# tuning: SQ counters of one conv_bench shape/tile (GPU box).  usage: tools/pmc_kernel.sh <shape> <tile> <outdir> [counter sets...]
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
SHAPE=$1; TILE=$2; OUT=$ROOT/$3; shift 3
mkdir -p $OUT
export SHAPES=$SHAPE TILES=$TILE
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/set$i -- python $ROOT/tools/conv_bench.py > $OUT/set$i.log 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/set*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    if "conv" not in k: continue
    print(k)
    for n, v in sorted(c.items()):
        print("   %-34s n=%3d  mean %.4g" % (n, len(v), sum(v) / len(v)))
PY
