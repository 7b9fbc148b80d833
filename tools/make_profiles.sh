#!/bin/bash
# Regenerates the judged profile artefacts on a GPU box (run through gpurun from the repo root):
#   gpurun_out/prof/stats      rocprofv3 --kernel-trace --stats of the default bench command
#   gpurun_out/prof/fetch|write  two separate PMC passes (FETCH_SIZE, WRITE_SIZE) -> hbm_traffic.json
#   gpurun_out/prof/mfma       PMC pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE) -> mfma_busy.txt
#   gpurun_out/prof/layers.txt per-layer HIP-event table
# tools/collect_profiles.py then turns them into profiles/<round>_*.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="$ROOT/bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-alt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python $BENCH > "$OUT/stats.log" 2>&1
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt --no-roofline > "$OUT/fetch.log" 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt --no-roofline > "$OUT/write.log" 2>&1
echo "write pass done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/mfma" -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt --no-roofline > "$OUT/mfma.log" 2>&1
echo "mfma pass done"
cd "$ROOT"
CF_LAYER_REPORT="$OUT/layers.txt" python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-alt > "$OUT/bench.log" 2>&1
python tools/collect_traffic.py "$OUT/fetch" "$OUT/write" "$OUT/hbm_traffic.json" > "$OUT/traffic_top.txt"
python tools/collect_mfma_busy.py "$OUT/mfma" "$OUT/mfma_busy.txt" > /dev/null
# keep only the small summaries (the merge back is capped at 64 MiB)
find "$OUT" -name "*counter_collection.csv" -delete
find "$OUT" -name "*kernel_trace.csv" -delete
tail -1 "$OUT/bench.log" | cut -c1-200
