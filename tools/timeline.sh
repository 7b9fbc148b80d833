#!/bin/bash
# tuning: kernel timeline of one concurrent step (GPU box): rocprofv3 --kernel-trace, last step printed per queue
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/timeline
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $ROOT/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-alt --no-roofline > $OUT/run.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step: find the last 'coords_init_kernel' (one per step)
idx = [i for i, r in enumerate(rows) if "coords_init" in r["Kernel_Name"]]
start = idx[-1]
# walk back to the first kernel of that step: the first conv of the encoders precedes coords_init by < 2 ms
t_ci = int(rows[start]["Start_Timestamp"])
first = start
while first > 0 and t_ci - int(rows[first - 1]["Start_Timestamp"]) < 1_800_000:
    first -= 1
t0 = int(rows[first]["Start_Timestamp"])
qs = {}
with open(out + "/timeline.txt", "w") as o:
    for r in rows[first:]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        q = qs.setdefault(r["Queue_Id"], len(qs))
        o.write("%8.1f %8.1f %6.1f q%d %s grid %s\n" % (s / 1e3, e / 1e3, (e - s) / 1e3, q, r["Kernel_Name"][:60], r.get("Grid_Size", "?")))
print("rows", len(rows) - first)
PY
find $OUT -name "*kernel_trace.csv" -delete
