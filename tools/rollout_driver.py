"""Fused K-step rollouts of the RK4/binary64 Acrobot (the FP64 vector-ALU workload), for profiling:
python tools/rollout_driver.py [num_envs] [K] [launches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gym_amd import _lib as L
from gym_amd.engine import Context, DeviceBatch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
k = int(sys.argv[2]) if len(sys.argv) > 2 else 64
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 3
torch.cuda.set_device(0)
ctx = Context(0)
b = DeviceBatch(ctx, L.ENV_ACROBOT, n, seed=1, flags=L.AUTORESET | L.RECORD_STATS | L.ACROBOT_RK4)
b.reset()
b.rollout(k); ctx.synchronize()
t0 = time.perf_counter()
for _ in range(launches):
    b.rollout(k)
ctx.synchronize()
dt = (time.perf_counter() - t0) / launches
s = b.stats()
print("acrobot rk4 fp64: %d envs x %d fused steps: %.3f ms per launch, %.3e env-steps/s, episodes %d mean length %.1f" %
      (n, k, dt * 1e3, n * k / dt, s.episodes, s.sum_length / max(s.episodes, 1)))
