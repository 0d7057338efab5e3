for g in 65536 43691 32768 21846 16384 8192 4096; do
  for r in 1 2; do GYMX_GRID_BLOCKS=$g timeout -k 10 200 python bench.py --no-extra --no-cpu-baseline --steps 100 | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grid', $g, round(j['roofline']['kernel_ms'],4))" || exit 1; done
done
