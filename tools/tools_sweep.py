"""Single-step kernel times of every env at its headline size."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def run(args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "200", "--warmup", "20", "--no-cpu-baseline", "--no-extra"] + args, capture_output=True, text=True)
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1])
        return "%.4f ms  %.3e steps/s  %.0f %s (%.1f%% of peak), %.4g B/env-step" % (
            j["roofline"]["kernel_ms"], j["value"], j["roofline"]["achieved"], j["roofline"]["unit"], 100 * j["roofline"]["frac"],
            j["roofline"].get("bytes_per_env_step", 0))
    except Exception:
        return r.stderr[-300:]
for env, n in [("cartpole", 1 << 24), ("mountaincar", 1 << 24), ("mountaincarcontinuous", 1 << 24), ("pendulum", 1 << 24), ("acrobot", 1 << 24), ("frozenlake", 1 << 26), ("pong", 65536), ("breakout", 65536), ("spaceinvaders", 65536)]:
    for pol in ("device", "random"):
        print(env, n, pol, run(["--env", env, "--num-envs", str(n), "--policy", pol]), flush=True)
print("cartpole fast32", run(["--mode", "fast32"]), flush=True)
print("acrobot fast32", run(["--env", "acrobot", "--mode", "fast32", "--policy", "random"]), flush=True)
print("acrobot rk4 single step", run(["--env", "acrobot", "--rk4", "--policy", "random"]), flush=True)
print("acrobot rk4 rollout 64", run(["--env", "acrobot", "--rk4", "--rollout", "64", "--steps", "5", "--warmup", "1"]), flush=True)
print("cartpole rollout 64", run(["--env", "cartpole", "--rollout", "64", "--steps", "5", "--warmup", "1"]), flush=True)
for e in ("mountaincar", "mountaincarcontinuous", "pendulum", "acrobot"):
    print(e + " rollout 64", run(["--env", e, "--rollout", "64", "--steps", "5", "--warmup", "1"]), flush=True)
for env in ("cartpole", "mountaincar", "mountaincarcontinuous", "pendulum", "acrobot"):
    print(env, "fp64 mode", run(["--env", env, "--mode", "fp64"]), flush=True)
print("taxi stub 64Mi", run(["--env", "taxi", "--num-envs", str(1 << 26)]), flush=True)
print("taxi full 64Mi random", run(["--env", "taxi", "--taxi-full", "--num-envs", str(1 << 26), "--policy", "random"]), flush=True)
print("frozenlake 8x8 slippery 64Mi random", run(["--env", "frozenlake", "--map", "8x8", "--slippery", "--num-envs", str(1 << 26), "--policy", "random"]), flush=True)
print("cartpole rollout 64, every step's reward and done rows", run(["--env", "cartpole", "--rollout", "64", "--rollout-out", "all", "--steps", "5", "--warmup", "1"]), flush=True)
print("cartpole rollout 64, statistics only", run(["--env", "cartpole", "--rollout", "64", "--rollout-out", "none", "--steps", "5", "--warmup", "1"]), flush=True)
for extra in ([], ["--policy", "random"], ["--blackjack-full"], ["--blackjack-full", "--policy", "random"]):
    print("blackjack 64Mi", " ".join(extra), run(["--env", "blackjack", "--num-envs", str(1 << 26)] + extra), flush=True)
