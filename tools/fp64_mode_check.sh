set -e
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q -k "fp64" --timeout 120 --timeout-method thread
for e in cartpole mountaincar mountaincarcontinuous pendulum acrobot; do
  timeout -k 10 200 python bench.py --env $e --mode fp64 --no-extra --no-cpu-baseline --steps 100 | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('$e', round(r['kernel_ms'],4), r['bytes_per_env_step'], round(r['frac'],3))"
done
