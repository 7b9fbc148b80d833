# tuning: whole-step throughput under library knobs (GPU box)
for kv in "X=0" "CF_CISTA_CHAINS=1" "CF_CISTA_CHAINS=2" "CF_CISTA_CHAINS=4" "CF_ENC_PAIR=1" "CF_PHASES=1"; do
  echo "== $kv"; env $kv python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|phases[^}]*}' | head -3
done
