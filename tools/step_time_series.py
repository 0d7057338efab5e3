"""Kernel time per step against the number of steps since Reset(), in chunks: the fused auto-reset makes a step's
cost depend on how many envs finish in it, which for some envs takes hundreds of steps to settle.
usage: python tools/step_time_series.py [env] [chunks] [steps per chunk]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_amd import _lib as L  # noqa: E402
from gym_amd.engine import Context, DeviceBatch  # noqa: E402

ENV = {"acrobot": L.ENV_ACROBOT, "cartpole": L.ENV_CARTPOLE, "mountaincar": L.ENV_MOUNTAINCAR}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "acrobot"
    chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    n = 1 << 24
    ctx = Context(0)
    b = DeviceBatch(ctx, ENV[name], n, seed=1, flags=L.AUTORESET | L.RECORD_STATS)
    b.reset()
    stream = torch.cuda.Stream()  # a stream of its own: the library takes a null stream pointer to mean its own stream
    torch.cuda.set_stream(stream)
    sp = stream.cuda_stream
    last = 0
    for c in range(chunks):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(per):
            b.step_random(sp)
        e1.record(stream)
        torch.cuda.synchronize()
        st = b.stats()
        print("steps %5d..%5d: %.4f ms per step, %.3f %% of the envs finish per step" %
              (c * per, (c + 1) * per, e0.elapsed_time(e1) / per, 100.0 * (st.episodes - last) / (n * per)), flush=True)
        last = st.episodes


if __name__ == "__main__":
    main()
