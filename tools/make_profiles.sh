#!/bin/bash
# Regenerates the judged profile artefacts on a GPU box (run through gpurun from the repo root):
#   gpurun_out/prof/ktrace       rocprofv3 --kernel-trace of the timed loop in SERIAL mode (CF_SERIAL=1: every kernel alone
#                                on the chip, same grids as the timed step) -> per (kernel, grid) durations a judge can
#                                recompute the roofline fractions from (tools/collect_ktrace.py joins them with the
#                                library's own per-launch-site table of algorithmic flops / bytes)
#   gpurun_out/prof/stats        rocprofv3 --kernel-trace --stats of the default (concurrent) bench command
#   gpurun_out/prof/fetch|write  two separate PMC passes (FETCH_SIZE, WRITE_SIZE) -> hbm_traffic.json
#   gpurun_out/prof/mfma         PMC pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE) -> mfma_busy.txt
#   gpurun_out/prof/layers.txt   per-launch-site HIP-event table (+ .json)
# tools/collect_profiles.py <round> then turns them into profiles/<round>_*.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="$ROOT/bench.py --steps 20 --warmup 4 --repeat 1 --no-latency --no-cpu-baseline --no-alt"
CF_SERIAL=1 rocprofv3 --kernel-trace --output-format csv -d "$OUT/ktrace" -- python $ROOT/bench.py --steps 10 --warmup 3 --repeat 1 --no-latency --no-cpu-baseline --no-alt --no-roofline > "$OUT/ktrace.log" 2>&1
echo "serial kernel-trace pass done"
# the same with TWICE the timed steps: launches whose count does not grow with the step count are set-up work (weight upload / packing),
# not part of a step (the copyBuffer / fillBuffer launches)
CF_SERIAL=1 rocprofv3 --kernel-trace --output-format csv -d "$OUT/ktrace20" -- python $ROOT/bench.py --steps 20 --warmup 3 --repeat 1 --no-latency --no-cpu-baseline --no-alt --no-roofline > "$OUT/ktrace20.log" 2>&1
python - "$OUT" <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
def counts(d):
    c = collections.Counter()
    for f in glob.glob(out + "/" + d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            c[(r["Kernel_Name"].split("(")[0][:60], r.get("Grid_Size", "?"))] += 1
    return c
a, b = counts("ktrace"), counts("ktrace20")
with open(out + "/setup_launches.txt", "w") as f:
    f.write("# launches per (kernel, grid) in two serial kernel traces of bench.py that differ ONLY in the number of timed steps: 3 warm-up + 10 timed\n")
    f.write("# against 3 + 20.  A count that is the same in both belongs to set-up (model.to(device): one small host-to-device\n")
    f.write("# copy per parameter tensor = __amd_rocclr_copyBuffer; weight packing: one hipMemsetAsync per packed matrix / bias = fillBufferAligned,\n")
    f.write("# pack_weight / wino*_weight kernels), not to a step; per-step launches grow by 10 x their per-step count.\n")
    f.write("%-62s %10s %8s %8s %12s\n" % ("kernel", "grid", "13 steps", "23 steps", "per step"))
    for k in sorted(set(a) | set(b), key=lambda k: (-(b[k] - a[k]), k)):
        per = (b[k] - a[k]) / 10.0
        if "rocclr" in k[0] or "weight" in k[0] or per == 0:
            f.write("%-62s %10s %8d %8d %12.1f\n" % (k[0], k[1], a[k], b[k], per))
PY
echo "setup-launch comparison done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python $BENCH --no-roofline > "$OUT/stats.log" 2>&1
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python $ROOT/bench.py --steps 4 --warmup 2 --repeat 1 --no-latency --no-cpu-baseline --no-alt --no-roofline > "$OUT/fetch.log" 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python $ROOT/bench.py --steps 4 --warmup 2 --repeat 1 --no-latency --no-cpu-baseline --no-alt --no-roofline > "$OUT/write.log" 2>&1
echo "write pass done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/mfma" -- python $ROOT/bench.py --steps 4 --warmup 2 --repeat 1 --no-latency --no-cpu-baseline --no-alt --no-roofline > "$OUT/mfma.log" 2>&1
echo "mfma pass done"
cd "$ROOT"
CF_LAYER_REPORT="$OUT/layers.txt" python bench.py --steps 20 --warmup 4 > "$OUT/bench.log" 2>&1
python tools/collect_ktrace.py "$OUT/ktrace" "$OUT/layers.txt.json" "$OUT/ktrace_serial.txt" > /dev/null
python tools/collect_traffic.py "$OUT/fetch" "$OUT/write" "$OUT/hbm_traffic.json" > "$OUT/traffic_top.txt"
python tools/collect_mfma_busy.py "$OUT/mfma" "$OUT/mfma_busy.txt" > /dev/null
CASES=convc2:47,fh.conv1:47,layer2:47,menc:47,zr.h:46,q.h:46,zr.v:46,cista.D:40,cista.D:48,cista.D:49,cista.P:40,cista.P:48,cista.P:49,gates:40,gates:48,gates:49,out_gates:40,out_gates:49,hs.gates:40,hs.gates:49,cista.D:23,cista.P:28,gates:25,layer1:23 CF_LIB_PATH=$ROOT/build_var/lib_stamp.so python tools/stamp_probe.py > "$OUT/stamps.txt" 2>&1 || true
[ -x tools/probe/mfma_shape_probe.bin ] && tools/probe/mfma_shape_probe.bin > "$OUT/mfma_clock_probe.txt" 2>&1 || true
# keep only the small summaries (the merge back is capped at 64 MiB)
find "$OUT" -name "*counter_collection.csv" -delete
find "$OUT" -name "*kernel_trace.csv" -delete
rm -rf "$OUT/ktrace20"
tail -1 "$OUT/bench.log" | cut -c1-200
