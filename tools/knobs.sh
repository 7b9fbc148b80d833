# tuning: whole-step throughput under library knobs (GPU box).  usage: tools/knobs.sh "A=1" "B=2 C=3" ...
for kv in "$@"; do
  for rep in 1 2; do
    echo "== $kv: $(env $kv python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-roofline ${BENCH_ARGS} 2>&1 | grep -o '"value": [0-9.]*' | head -1)"
  done
done
