import os, sys, json, subprocess
env = dict(os.environ)
def run(extra_env, args):
    e = dict(env); e.update(extra_env)
    r = subprocess.run([sys.executable, "bench.py", "--steps", "50", "--warmup", "5", "--no-cpu-baseline", "--no-extra"] + args, env=e, capture_output=True, text=True)
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1])
        return j["roofline"]["kernel_ms"], j["roofline"]["achieved"]
    except Exception as ex:
        return r.stderr[-300:], None
for blocks in ["512", "1024", "2048", "4096", "8192", "16384", "65536"]:
    print("grid", blocks, run({"GYMX_GRID_BLOCKS": blocks}, []), flush=True)
print("fast32", run({}, ["--mode", "fast32"]), flush=True)
print("random policy", run({}, ["--policy", "random"]), flush=True)
for envn in ["mountaincar", "pendulum", "acrobot", "frozenlake", "pong"]:
    print(envn, run({}, ["--env", envn]), flush=True)
    print(envn, "random", run({}, ["--env", envn, "--policy", "random"]), flush=True)
