for e in "frozenlake --num-envs 67108864" "frozenlake --map 8x8 --slippery --num-envs 67108864 --policy random" "frozenlake --map 8x8 --num-envs 67108864" "taxi --num-envs 67108864" "taxi --taxi-full --num-envs 67108864 --policy random"; do
  for r in 1 2; do for b in 128 256; do
    GYMX_TABULAR_BLOCK=$b timeout -k 10 300 python bench.py --env $e --no-extra --no-cpu-baseline --steps 100 | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"$e\", $b, round(j[\"roofline\"][\"kernel_ms\"],4))" || exit 1
  done; done
done
