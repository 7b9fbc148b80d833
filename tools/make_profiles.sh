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
BENCH="$ROOT/bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-alt"
CF_SERIAL=1 rocprofv3 --kernel-trace --output-format csv -d "$OUT/ktrace" -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-roofline > "$OUT/ktrace.log" 2>&1
echo "serial kernel-trace pass done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python $BENCH --no-roofline > "$OUT/stats.log" 2>&1
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt --no-roofline > "$OUT/fetch.log" 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt --no-roofline > "$OUT/write.log" 2>&1
echo "write pass done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/mfma" -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt --no-roofline > "$OUT/mfma.log" 2>&1
echo "mfma pass done"
cd "$ROOT"
CF_LAYER_REPORT="$OUT/layers.txt" python bench.py --steps 20 --warmup 4 > "$OUT/bench.log" 2>&1
python tools/collect_ktrace.py "$OUT/ktrace" "$OUT/layers.txt.json" "$OUT/ktrace_serial.txt" > /dev/null
python tools/collect_traffic.py "$OUT/fetch" "$OUT/write" "$OUT/hbm_traffic.json" > "$OUT/traffic_top.txt"
python tools/collect_mfma_busy.py "$OUT/mfma" "$OUT/mfma_busy.txt" > /dev/null
CASES=convc2:47,fh.conv1:47,layer2:47,zr.h:46,q.h:46,zr.v:46,zr.h:20,cista.D:40,cista.P:40,gates:40,out_gates:40,gates:42,cista.P:42,cista.D:42,cista.D:23,cista.P:28,gates:25,layer1:23,gru.zr:20,gru.q:22,convc2:20 CF_LIB_PATH=$ROOT/build_var/lib_stamp.so python tools/stamp_probe.py > "$OUT/stamps.txt" 2>&1 || true
[ -x tools/probe/mfma_shape_probe.bin ] && tools/probe/mfma_shape_probe.bin > "$OUT/mfma_clock_probe.txt" 2>&1 || true
# keep only the small summaries (the merge back is capped at 64 MiB)
find "$OUT" -name "*counter_collection.csv" -delete
find "$OUT" -name "*kernel_trace.csv" -delete
tail -1 "$OUT/bench.log" | cut -c1-200
