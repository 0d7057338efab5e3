# kernel time per step against the number of workgroups (trips per workgroup = lane groups / workgroups)
# usage: tools/grid_sweep_env.sh <bench.py workload flags...>
for g in 65536 32768 21846 16384 10923 8192 4096; do
  GYMX_GRID_BLOCKS=$g timeout -k 10 200 python bench.py --no-extra --no-cpu-baseline --steps 100 "$@" | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grid', $g, round(j['roofline']['kernel_ms'],4))" || exit 1
done
