"""A/B runner: python tools_ab.py variant1 variant2 ... [-- bench args]; interleaved rounds in separate processes."""
import os, sys, json, subprocess
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--"); extra = args[i + 1:]; args = args[:i]
rounds = 3
res = {v: [] for v in args}
for r in range(rounds):
    for v in args:
        e = dict(os.environ)
        if v != "default":
            e["GYMX_LIB"] = os.path.abspath("gym_amd/lib/variants/libgymx_%s.so" % v)
        p = subprocess.run([sys.executable, "bench.py", "--steps", "100", "--warmup", "10", "--no-cpu-baseline", "--no-extra"] + extra, env=e, capture_output=True, text=True)
        try:
            j = json.loads(p.stdout.strip().splitlines()[-1])
            res[v].append(j["roofline"]["kernel_ms"])
        except Exception:
            res[v].append(None); print(v, p.stderr[-500:], flush=True)
for v in args:
    xs = [x for x in res[v] if x]
    print("%-12s kernel ms per step: %s  best %.4f" % (v, ["%.4f" % x for x in xs], min(xs) if xs else 0), flush=True)
