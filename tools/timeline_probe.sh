#!/bin/bash
# Timeline of ONE steady-state step (rocprofv3 --kernel-trace of the concurrent bench command): every kernel with its queue, start relative to
# the step's first kernel and duration, plus gaps on the critical path.  usage: bash tools/timeline_probe.sh [B]   (GPU box)
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-1}
OUT=$ROOT/gpurun_out/timeline_b$B
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $ROOT/bench.py --batch $B --steps 12 --warmup 6 --repeat 1 --no-cpu-baseline --no-alt --no-roofline --no-latency > $OUT/run.log 2>&1
python - "$OUT" $B <<'PY'
import csv, glob, sys, re
out, B = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True)[0]
rd = list(csv.DictReader(open(f)))
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?"))) for r in rd]
rows.sort()
marks = [s for s, e, n, q, g in rows if "corr_pyramid" in n]
# a step = from the first kernel after the previous step's last CISTA kernel; approximate by the window between two pyramid marks, shifted to
# the first encoder kernel: print the window [marks[-3], marks[-2]) -- one whole period of the steady state
t0, t1 = marks[-3], marks[-2]
qs = {}
def short(n):
    n = re.sub(r"^void cf::", "", n); n = re.sub(r"\(.*$", "", n); return n[:44]
with open(out + "/timeline.txt", "w") as fo:
    fo.write("# B=%s, one period of the steady state (pyramid mark to pyramid mark): %.3f ms\n" % (B, (t1 - t0) / 1e6))
    fo.write("# %8s %8s  q  %-44s %s\n" % ("start_us", "dur_us", "kernel", "grid"))
    last_end = t0
    for s, e, n, q, g in rows:
        if s < t0 or s >= t1: continue
        qi = qs.setdefault(q, len(qs))
        gap = (s - last_end) / 1e3
        fo.write("%10.1f %8.1f  %d  %-44s %s%s\n" % ((s - t0) / 1e3, (e - s) / 1e3, qi, short(n), g, ("   <- nothing ran for %.1f us" % gap) if gap > 1.5 else ""))
        last_end = max(last_end, e)
print(open(out + "/timeline.txt").read()[:300])
PY
find $OUT -name "*kernel_trace.csv" -delete; rm -rf $OUT/kt
