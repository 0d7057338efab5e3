"""Long free-running comparison of a window of a large device batch with the CPU oracle keyed at the same global env
indices (in-kernel random policy, auto-reset): counts the state elements that differ in bits at checkpoints.
usage: python tools/soak_parity.py [steps] [window]      (runs on a GPU box; the oracle is the checker)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_common import L, bits, ctx  # noqa: E402
from gym_amd.engine import DeviceBatch, torch_views  # noqa: E402
from oracle import oracle as O  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
width = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
n, base, every = 1 << 20, 777216, 500
flags = L.AUTORESET | L.RECORD_STATS
for name, env in (("cartpole", L.ENV_CARTPOLE), ("mountaincar", L.ENV_MOUNTAINCAR), ("mountaincarcontinuous", L.ENV_MOUNTAINCAR_CONT),
                  ("pendulum", L.ENV_PENDULUM), ("acrobot", L.ENV_ACROBOT)):
    g = DeviceBatch(ctx(), env, n, seed=31, flags=flags)
    o = O.OracleVec(env, width, seed=31, flags=O.AUTORESET, env_index_base=base)
    g.reset(); o.reset()
    v = torch_views(g)
    worst = 0
    for t in range(0, steps, every):
        g.rollout(every) if (t // every) % 2 else [g.step_random() for _ in range(every)]
        for _ in range(every):
            o.step_random()
        g.ctx.synchronize()
        orr, oi = o.get_state()
        gr = v["state_real"][:orr.shape[0], base:base + width].cpu().numpy()
        gi = v["state_int"][:, base:base + width].cpu().numpy()
        bad = int((bits(gr) != bits(orr)).sum())
        worst = max(worst, bad)
        if bad or not np.array_equal(gi, oi):
            print(name, "step", t + every, "real elements differing", bad, "int rows equal", np.array_equal(gi, oi), flush=True)
            break
    print(name, "%d steps x %d envs: %s" % (steps, width, "bit-identical at every checkpoint" if worst == 0 else "DIFFERS"), flush=True)
    g.close()
