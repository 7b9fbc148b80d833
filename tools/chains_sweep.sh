for c in 1 2; do
  export CF_CISTA_CHAINS=$c
  echo "chains=$c idnet16 $(python bench.py --no-cpu-baseline --no-alt --no-roofline --model idnet --batch 16 --height 260 --width 346 2>&1 | grep -o '"value": [0-9.]*')"
  echo "chains=$c hs4 $(python bench.py --no-cpu-baseline --no-alt --no-roofline --batch 4 --height 480 --width 640 --steps 10 --warmup 3 2>&1 | grep -o '"value": [0-9.]*')"
  echo "chains=$c b4 $(python bench.py --no-cpu-baseline --no-alt --no-roofline --batch 4 --steps 30 --warmup 5 2>&1 | grep -o '"value": [0-9.]*')"
  echo "chains=$c b2 $(python bench.py --no-cpu-baseline --no-alt --no-roofline --batch 2 --steps 30 --warmup 5 2>&1 | grep -o '"value": [0-9.]*')"
  echo "chains=$c b16 $(python bench.py --no-cpu-baseline --no-alt --no-roofline --batch 16 --steps 20 --warmup 5 2>&1 | grep -o '"value": [0-9.]*')"
  echo "chains=$c eraft8 $(python bench.py --no-cpu-baseline --no-alt --no-roofline --model eraft 2>&1 | grep -o '"value": [0-9.]*')"
done
