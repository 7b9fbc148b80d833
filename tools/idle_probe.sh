#!/bin/bash
# How much of a timed step is the chip idle (no kernel running on any queue), and how is the busy time split by the number of kernels that
# overlap?  rocprofv3 --kernel-trace of the default (concurrent) bench command; the last STEPS steps are analysed.  GPU box.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-8}
OUT=$ROOT/gpurun_out/idle_b$B
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $ROOT/bench.py --batch $B --steps 12 --warmup 6 --repeat 1 --no-cpu-baseline --no-alt --no-roofline --no-latency > $OUT/run.log 2>&1
python - "$OUT" $B <<'PY'
import csv, glob, sys
out, B = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# step boundaries: one corr_pyramid_kernel per step
marks = [s for s, e, n in rows if "corr_pyramid" in n]
marks = marks[-9:]                     # the last 8 whole steps
t0, t1 = marks[0], marks[-1]
ev = []
for s, e, n in rows:
    if e <= t0 or s >= t1: continue
    ev.append((max(s, t0), 1)); ev.append((min(e, t1), -1))
ev.sort()
depth, last, hist = 0, t0, {}
for t, d in ev:
    hist[depth] = hist.get(depth, 0) + (t - last)
    depth += d; last = t
hist[depth] = hist.get(depth, 0) + (t1 - last)
tot = float(t1 - t0)
nst = len(marks) - 1
print("B=%s: %d steps, %.3f ms per step (kernel-trace clock); kernels per step %.0f" % (B, nst, tot / nst / 1e6, sum(1 for s, e, n in rows if t0 <= s < t1) / nst))
for k in sorted(hist):
    print("   %d kernel(s) running: %5.1f %% of the time (%.3f ms per step)" % (k, 100.0 * hist[k] / tot, hist[k] / nst / 1e6))
busy = sum(e - s for s, e, n in rows if t0 <= s < t1)
print("   sum of kernel durations per step: %.3f ms" % (busy / nst / 1e6))
PY
find $OUT -name "*kernel_trace.csv" -delete
