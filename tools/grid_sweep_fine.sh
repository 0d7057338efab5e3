for g in 32768 36409 39322 40960 43691 46811 49152 52429 58254; do
  for r in 1 2; do GYMX_GRID_BLOCKS=$g timeout -k 10 200 python bench.py --no-extra --no-cpu-baseline --steps 100 $* 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'grid', $g, round(j['roofline']['kernel_ms'],4))" || exit 1; done
done
