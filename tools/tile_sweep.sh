set -e
run() { name=$1; shift; CF_TILE_OVERRIDE="$1" CF_LAYER_REPORT=gpurun_out/sweep_$name.txt python bench.py --no-cpu-baseline --no-alt --steps 10 --warmup 3 > gpurun_out/sweep_$name.log 2>&1; python - <<PY
import json
d=json.loads(open("gpurun_out/sweep_$name.log").read().strip().splitlines()[-1])
print("$name", "$1", d["value"], d["roofline"]["all_conv"]["ms_per_step"])
PY
grep -E "cista\.(D|P|P0|Dg|out_gates|gates|Gates|W0|upsamp) " gpurun_out/sweep_$name.txt; }
run A ""
run B "cista.out_gates=23,cista.P=23,cista.P0=23"
run C "cista.out_gates=26,cista.P=26,cista.P0=26"
run D "cista.D=23,cista.Dg=23,cista.W0=23,cista.gates=23,cista.Gates=23"
run E "cista.D=24,cista.Dg=24,cista.gates=25,cista.Gates=25"
