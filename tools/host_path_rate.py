"""PCIe-inclusive rate of the host path (gx_vec_step_async + gx_vec_wait): actions in, reward +
done (+ observation) out, pinned buffers.  usage: host_path_rate.py [num_envs] (default 16 Mi)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gym_amd import _lib as L
from gym_amd.engine import Context, DeviceBatch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
ctx = Context(0)
b = DeviceBatch(ctx, L.ENV_CARTPOLE, n, seed=1, flags=L.AUTORESET | L.RECORD_STATS)
b.reset()
lib = L.load()
act = torch.randint(0, 2, (n,), dtype=torch.int32).pin_memory()
rew = torch.empty(n, dtype=torch.float32).pin_memory()
done = torch.empty(n, dtype=torch.uint8).pin_memory()
obs = torch.empty((4, n), dtype=torch.float32).pin_memory()
vp = lambda t: C.c_void_p(t.data_ptr())
for name, o in (("actions in, reward+done out", None), ("actions in, reward+done+obs out", obs)):
    for _ in range(3):
        L.check(lib.gx_vec_step_async(b._h, vp(act), vp(o) if o is not None else None, vp(rew), vp(done), None)); L.check(lib.gx_vec_wait(b._h))
    t0 = time.perf_counter(); k = 20 if n > 1 << 20 else 2000
    for _ in range(k):
        L.check(lib.gx_vec_step_async(b._h, vp(act), vp(o) if o is not None else None, vp(rew), vp(done), None)); L.check(lib.gx_vec_wait(b._h))
    dt = (time.perf_counter() - t0) / k
    byt = n * (4 + 4 + 1 + (16 if o is not None else 0))
    print("%s: %.3f ms/step, %.3e env-steps/s, %.1f GB/s over PCIe" % (name, dt * 1e3, n / dt, byt / dt / 1e9))
