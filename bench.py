#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the batched step path on MI355X.

One "step" = one VectorEnv.Step over the whole batch: CartPole-v1, 16 Mi
envs per GPU, single-step kernel with auto-reset and RecordEpisodeStatistics
fused, actions read from a device-resident array (zero-copy path), EXACT
numeric mode.  State, actions and outputs are resident in HBM before the
timed region starts, and the batch has been stepped out of its post-Reset()
transient (--settle untimed steps: every env starts its first episode at the
same tick, so the share of envs finishing per step -- which the fused
auto-reset's cost follows -- takes tens of steps to reach its stationary rate).

  python bench.py --gpus 1 --steps 200 --warmup 20
  python bench.py --gpus N ...            (starts its own N ranks, one per GPU)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Started plainly with --gpus N > 1 (no WORLD_SIZE in the environment) this process only launches
N fresh children of the same script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays
rank 0's JSON line and exits with the worst exit code; it never imports torch or touches a GPU.
Rank 0 prints one JSON line.  The batch is sharded by env index (RNG streams
are keyed by the global index); there is no collective in the step path, and
episode statistics are all-reduced once over RCCL after the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# RCCL shares buffers between ranks through dmabuf handles; the legacy IPC mode fails on hosts whose driver
# only supports dmabuf (hipIpcGetMemHandle: invalid argument).  Inherited by the ranks spawn_ranks() starts.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def _gpus_from_argv(argv):
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--gpus="):
            return int(a.split("=", 1)[1])
    return 1


def spawn_ranks(world, argv, script=None, grace=None):
    """One child process per rank, each a fresh interpreter that initialises its own GPU.  Returns
    the exit code of the first rank that failed (0 if none did).  Rank 0's standard output is
    relayed by a thread while ALL children are polled, so a rank that dies at start-up is seen at
    once: the others (which would wait for it in the rendezvous or at a barrier until torch's own
    timeout) get `grace` seconds to end by themselves, are then terminated, each by its own PID,
    and killed ten seconds later if they ignore that."""
    import socket
    import subprocess
    import threading

    script = script or os.path.abspath(__file__)
    grace = float(os.environ.get("GYMX_BENCH_GRACE_S", "30")) if grace is None else grace
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-u", script] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))

    def relay():  # rank 0 is the only one that prints; ends when its pipe closes
        for line in iter(procs[0].stdout.readline, b""):
            sys.stdout.write(line.decode("utf-8", "replace"))
            sys.stdout.flush()

    t = threading.Thread(target=relay, daemon=True)
    t.start()
    first_bad, deadline, killed_at = 0, None, None
    pending = dict(enumerate(procs))
    while pending:
        for r, p in list(pending.items()):
            rc = p.poll()
            if rc is None:
                continue
            del pending[r]
            if rc != 0 and deadline is None:
                first_bad = rc if rc > 0 else 128 - rc
                deadline = time.time() + grace
                print("bench.py launcher: rank %d exited with code %d; the other ranks have %.0f s to end"
                      % (r, rc, grace), file=sys.stderr)
        now = time.time()
        if pending and deadline is not None and now > deadline:
            if killed_at is None:
                print("bench.py launcher: terminating ranks %s" % sorted(pending), file=sys.stderr)
                for p in pending.values():
                    p.terminate()
                killed_at = now + 10.0
            elif now > killed_at:
                for p in pending.values():
                    p.kill()
                killed_at = now + 1e9
        time.sleep(0.05)
    t.join(timeout=5.0)
    return first_bad


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ and _gpus_from_argv(sys.argv[1:]) > 1:
    # the launcher: before torch is imported, before anything can touch a GPU
    sys.exit(spawn_ranks(_gpus_from_argv(sys.argv[1:]), sys.argv[1:]))

import torch  # noqa: E402  (device memory, streams, events, torch.distributed only)

from gym_amd import _lib as L  # noqa: E402
from gym_amd.distributed import allreduce_stats, shard  # noqa: E402
from gym_amd.engine import Context, DeviceBatch  # noqa: E402

COLLECTIVE_DEVICE = "cuda"   # tensors handed to torch.distributed live here (cpu for the gloo rehearsal)
ACTION_RING = int(os.environ.get("GYMX_BENCH_ACTION_RING", "8"))  # rows of pre-sampled actions cycled step by step
HBM_PEAK_GBPS = 8000.0       # MI355X HBM3E peak
FP64_VALU_PEAK_TFLOPS = 78.6  # MI355X vector FP64 peak

# algorithmic bytes per env-step, single-step mode with auto-reset and
# statistics fused: each logical array counted once per read and once per
# write at its stored width
B_ALG = {
    # state r+w, action r, steps r+w, reward w, done w.  CartPole, MountainCar and Acrobot pay a
    # constant reward on every non-final step, so with auto-reset the open-episode return is that
    # constant times the step counter and no ep_return row is read or written
    "cartpole": 16 + 16 + 4 + 4 + 4 + 4 + 1,               # 49
    "mountaincar": 8 + 8 + 4 + 4 + 4 + 4 + 1,              # 33
    "acrobot": 24 + 24 + 4 + 4 + 4 + 4 + 1,                # 65
    "frozenlake": 4 + 4 + 4 + 4 + 4 + 4 + 1,               # 25 (every non-terminal move pays 0)
    "taxi": 4 + 4 + 4 + 4 + 4 + 4 + 1,                     # 25 for the reference's stub; --taxi-full adds ep_return r+w
    # the others also read and write ep_return
    "pendulum": 12 + 12 + 4 + 4 + 4 + 8 + 8 + 8 + 1,       # 61: reward and ep_return are binary64
    "mountaincarcontinuous": 8 + 8 + 4 + 4 + 4 + 8 + 8 + 8 + 1,  # 53: likewise
    "pong": 60 + 60 + 4 + 4 + 4 + 4 + 1 + 100800,          # 100937
    "breakout": 72 + 72 + 4 + 4 + 4 + 4 + 1 + 100800,      # 100961
    "spaceinvaders": 140 + 140 + 4 + 4 + 4 + 4 + 1 + 100800,  # 101097
    # observation index r+w, steps r+w, reward, done.  The reference's Step ignores its action
    # (Envs.Blackjack.cs:35-37) and so does the kernel: the action row is never read.  A reward other than 0 ends
    # the episode under either rule set, so the open-episode return is always 0 and its row is never moved.
    # --blackjack-full (upstream's rules) adds the action and the face-down row r+w.
    "blackjack": 4 + 4 + 4 + 4 + 4 + 1,                   # 21
}
# binary64 mode: the state rows double in width (read + write), nothing else changes
B_ALG_FP64_EXTRA = {"cartpole": 32, "mountaincar": 16, "mountaincarcontinuous": 16, "pendulum": 24, "acrobot": 48}
ENV_IDS = {"cartpole": L.ENV_CARTPOLE, "mountaincar": L.ENV_MOUNTAINCAR, "acrobot": L.ENV_ACROBOT,
           "mountaincarcontinuous": L.ENV_MOUNTAINCAR_CONT, "taxi": L.ENV_TAXI,
           "pendulum": L.ENV_PENDULUM, "frozenlake": L.ENV_FROZENLAKE, "pong": L.ENV_PONG, "breakout": L.ENV_BREAKOUT,
           "spaceinvaders": L.ENV_SPACEINVADERS, "blackjack": L.ENV_BLACKJACK}
FRAME_ENVS = ("pong", "breakout", "spaceinvaders")
# FP64 operations per env-step of the RK4 Acrobot (4 derivative evaluations of
# ~32 flops, 52 for the RK4 combination, ~16 for wrap/clamp/termination),
# transcendentals not counted
ACROBOT_RK4_FLOPS = 196
# the same per env-step as the hardware counts it, sin/cos polynomials and reciprocals included:
# (SQ_INSTS_VALU_ADD_F64 + MUL_F64 + TRANS_F64 + 2 FMA_F64) x 64 lanes / (envs x K) of the fused
# 64-step rollout, profiles/r3/fp64_valu_acrobot_rk4_rollout64.json (tools/fp64_profile.sh); 533.1 before the
# table-driven sin/cos, which needs fewer binary64 operations per evaluation
ACROBOT_RK4_FLOPS_COUNTED = 486.9
ACROBOT_RK4_FLOPS_SOURCE = "SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64, profiles/r3/fp64_valu_acrobot_rk4_rollout64.json"


def time_steps(batch, step_fn, steps, warmup, dist):
    """W untimed + K timed steps; returns (wall seconds max over ranks, mean kernel ms from HIP events)"""
    stream = torch.cuda.current_stream()
    for _ in range(warmup):
        step_fn()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(steps):
        step_fn()
    ev1.record(stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / steps
    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64, device=COLLECTIVE_DEVICE)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    return wall, kernel_ms


def join_ranks(rank, world, timeout_s):
    """Roll call before init_process_group, over the rendezvous store itself: every rank writes its
    key, rank 0 waits for all of them for at most `timeout_s` and names the ranks that never came
    (a rank that died at start-up would otherwise only show as a timeout somewhere inside the
    backend).  Returns the store to build the process group on."""
    import datetime

    from torch.distributed import PrefixStore, TCPStore

    host, port = os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"])
    # under torch.distributed.run the launcher's agent already serves a store on this port and every worker,
    # rank 0 included, is a client of it
    agent_store = os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True"
    try:
        store = TCPStore(host, port, world, is_master=(rank == 0 and not agent_store),
                         timeout=datetime.timedelta(seconds=timeout_s), wait_for_workers=False, multi_tenant=True)
    except Exception as e:  # noqa: BLE001  (rank 0's store never came up)
        print("bench.py: rank %d could not reach the rendezvous store at %s:%d within %d s: %s" % (rank, host, port, timeout_s, e),
              file=sys.stderr)
        sys.exit(4)
    store = PrefixStore("gymx_bench", store)
    store.set("joined/%d" % rank, b"1")
    if rank == 0:
        deadline = time.time() + timeout_s
        missing = list(range(1, world))
        while missing and time.time() < deadline:
            missing = [r for r in missing if not store.check(["joined/%d" % r])]
            if missing:
                time.sleep(0.05)
        if missing:
            print("bench.py: ranks %s of %d did not join within %d s" % (missing, world, timeout_s), file=sys.stderr)
            store.set("go", b"0")
            sys.exit(4)
        store.set("go", b"1")
    else:
        try:
            store.wait(["go"], datetime.timedelta(seconds=timeout_s))
        except Exception:  # noqa: BLE001
            print("bench.py: rank %d waited %d s for the roll call of rank 0" % (rank, timeout_s), file=sys.stderr)
            sys.exit(4)
        if store.get("go") != b"1":
            sys.exit(4)
    return store


def cpu_baseline(env_name, target_seconds=12.0):
    """The oracle arranged like the reference's VectorEnv (heap objects, virtual Step,
    one sequential loop, VectorEnv.cs:28-34) timed on one host core, on a bounded sample."""
    from oracle import oracle as O

    eid = {"cartpole": O.CARTPOLE, "acrobot": O.ACROBOT, "mountaincar": O.MOUNTAINCAR, "pendulum": O.PENDULUM,
           "taxi": O.TAXI, "mountaincarcontinuous": O.MOUNTAINCAR_CONT, "frozenlake": O.FROZENLAKE, "pong": O.PONG, "breakout": O.BREAKOUT, "spaceinvaders": O.SPACEINVADERS,
           "blackjack": O.BLACKJACK}[env_name]
    n = 1024 if env_name in FRAME_ENVS else 1 << 16
    secs, steps_done = O.bench_reference_shaped(eid, n, 8 if env_name in FRAME_ENVS else 16)
    rate = steps_done / max(secs, 1e-9)
    k = max(8, int(target_seconds * rate / n))
    secs, steps_done = O.bench_reference_shaped(eid, n, k)
    return {
        "value": steps_done / secs, "unit": "env-steps/s", "cores": 1, "kind": "port",
        "sample": "%s, %d envs x %d batched steps, random actions, reset on done; C restatement shaped like the "
                  "reference VectorEnv loop (no per-step Dictionary/boxing, so it flatters the C# reference)" % (env_name, n, k),
    }


def cpu_baseline_all_cores(env_name, target_seconds=5.0):
    """The same env code on every host core this job may use (at most 16): contiguous env
    structs, no result lists, one thread per core.  FrozenLake shares a map object and is skipped."""
    from oracle import oracle as O

    if env_name in ("frozenlake", "taxi", "blackjack"):
        return None
    eid = {"cartpole": O.CARTPOLE, "acrobot": O.ACROBOT, "mountaincar": O.MOUNTAINCAR, "pendulum": O.PENDULUM,
           "mountaincarcontinuous": O.MOUNTAINCAR_CONT, "pong": O.PONG, "breakout": O.BREAKOUT, "spaceinvaders": O.SPACEINVADERS}[env_name]
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    n = 1 << 18
    secs, steps_done = O.bench_flat_threads(eid, n, 4, cores)
    k = max(4, int(target_seconds * steps_done / max(secs, 1e-9) / n))
    secs, steps_done = O.bench_flat_threads(eid, n, k, cores)
    return {"value": steps_done / secs, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%s game/dynamics step only (no frame rendering), %d envs x %d batched steps, random actions, reset on "
                      "done; contiguous env structs, no result lists, %d threads" % (env_name, n, k, cores)}


def d2d_copy_rate(nbytes=1 << 31, reps=5):
    """Measured device-to-device copy rate (bytes read + bytes written per second), GB/s"""
    src = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(src)
    dst.copy_(src)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def bytes_per_env_step(env, policy, mode="exact", rk4=False, taxi_full=False, blackjack_full=False):
    """Algorithmic bytes one env-step of the single-step kernel moves (B_ALG's rule)"""
    if rk4:
        # f64 state r+w, steps r+w, reward, done, derived f32 observation rows
        return 32 + 32 + 4 + 4 + 4 + 1 + 24 + (4 if policy == "device" else 0)
    b = B_ALG[env] + (8 if taxi_full or blackjack_full else 0)
    if env == "blackjack":
        b += 4 if blackjack_full and policy != "random" else 0  # only upstream's rules look at the action
    elif policy == "random":
        b -= 4
    if mode == "fp64" and env in B_ALG_FP64_EXTRA:
        b += B_ALG_FP64_EXTRA[env]
    return b


def device_action_ring(batch, stream_ptr):
    """A ring of pre-sampled action rows, ActionSpace.Sample() of ticks 0..7, cycled step by step: the policy
    is random in time as well as across envs, and a step still reads one 4-byte action per env.  Returns the
    step function (the ring tensor stays alive in its closure)."""
    adt = torch.float32 if batch.info.action_dtype == L.F32 else torch.int32
    ring = torch.empty((ACTION_RING, int(batch.view().stride)), dtype=adt, device="cuda")
    for t in range(ACTION_RING):
        batch.sample_actions(t, ring[t].data_ptr(), stream_ptr)
    ring_ptrs = [ring[t].data_ptr() for t in range(ACTION_RING)]
    turn = [0]

    def step_fn(_keep=ring):
        batch.step_device(ring_ptrs[turn[0] % ACTION_RING], stream_ptr)
        turn[0] += 1

    return step_fn


def rank_evidence(dist, backend, rank, world, local_rank, kernel_ms):
    """What shows that the job really had `world` ranks on `world` devices: every rank's device and kernel
    time (gathered over the collective backend itself) and the time of one statistics all-reduce."""
    prop = torch.cuda.get_device_properties(local_rank)
    mine = {"rank": rank, "local_rank": local_rank, "device": prop.name,
            "pci_bus_id": getattr(prop, "pci_bus_id", None), "kernel_ms": kernel_ms, "pid": os.getpid()}
    if dist is None:
        return {"world": 1, "backend": None, "per_rank_kernel_ms": [kernel_ms], "per_rank": [mine], "allreduce_us": None}
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    # the collective of the data path: two tiny all-reduces (sums, extrema) per statistics record
    sums = torch.zeros(4, dtype=torch.float64, device=COLLECTIVE_DEVICE)
    ext = torch.zeros(2, dtype=torch.float64, device=COLLECTIVE_DEVICE)
    reps = 20
    for timed in (False, True):
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            dist.all_reduce(sums, op=dist.ReduceOp.SUM)
            dist.all_reduce(ext, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / reps * 1e6
    return {"world": world, "backend": "rccl (torch nccl)" if backend == "nccl" else backend,
            "per_rank_kernel_ms": [e["kernel_ms"] for e in everyone], "per_rank": everyone,
            "allreduce_us": us, "allreduce_what": "one statistics record: all_reduce(sum, 4 x f64) + all_reduce(max, 2 x f64), mean of %d" % reps}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle", type=int, default=100,
                    help="untimed steps after Reset() and before the warm-up, so that the batch is in the stationary regime "
                         "of its policy (single-step workloads only)")
    ap.add_argument("--env", default="cartpole", choices=sorted(ENV_IDS))
    ap.add_argument("--num-envs", type=int, default=0, help="envs per GPU (default 16 Mi; Pong, Breakout 65536)")
    ap.add_argument("--policy", default="device", choices=["device", "random"])
    ap.add_argument("--mode", default="exact", choices=["exact", "fast32", "fp64"])
    ap.add_argument("--map", default=None, help="FrozenLake only: '8x8' or rows joined by '/'")
    ap.add_argument("--slippery", action="store_true", help="FrozenLake only: the is_slippery extension")
    ap.add_argument("--taxi-full", action="store_true", help="Taxi only: upstream Taxi-v3 dynamics (extension)")
    ap.add_argument("--blackjack-full", action="store_true", help="Blackjack only: upstream Blackjack-v1 rules (extension)")
    ap.add_argument("--rk4", action="store_true", help="Acrobot only: the RK4 / binary64 extension (upstream dynamics)")
    ap.add_argument("--rollout", type=int, default=1,
                    help="K > 1: every timed step is one fused K-step rollout launch (state stays in registers, "
                         "actions sampled in-kernel); env-steps are counted as envs x K per launch")
    ap.add_argument("--rollout-out", default="last", choices=["last", "all", "none"],
                    help="with --rollout K: what the fused rollout hands back -- the last step's reward / done rows, every "
                         "step's (written to [K][stride] device rows, 5 or 9 more bytes per env-step), or statistics only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary measurements (the other named workloads)")
    ap.add_argument("--extra-envs-scale", type=float, default=1.0,
                    help="tests only: scales the env counts of the secondary measurements")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl is RCCL on ROCm; gloo only to rehearse several ranks on one GPU")
    ap.add_argument("--share-device", action="store_true", help="rehearsal: every rank uses GPU local_rank %% device_count")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: initialise torch.distributed even for one rank (exercises the RCCL barrier / all-reduce path)")
    ap.add_argument("--join-timeout", type=int, default=120,
                    help="seconds a rank waits for the others at start-up (roll call and process-group initialisation)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    result_line = claim_stdout()
    if os.environ.get("GYMX_BENCH_FAIL_RANK") == str(rank):
        # fault injection for the launcher's test: this rank dies before it joins anything
        print("bench.py: rank %d leaves at start-up (GYMX_BENCH_FAIL_RANK)" % rank, file=sys.stderr)
        sys.exit(3)
    dist = None
    if world > 1 or args.force_dist:
        import datetime

        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.share_device:
            local_rank %= torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        store = join_ranks(rank, world, args.join_timeout)
        # naming the device binds the RCCL communicator to this rank's GPU up front (no guessing at
        # the first barrier)
        kw = {"device_id": torch.device("cuda", local_rank)} if args.dist_backend == "nccl" else {}
        try:
            dist_mod.init_process_group(args.dist_backend, store=store, rank=rank, world_size=world,
                                        timeout=datetime.timedelta(seconds=args.join_timeout), **kw)
        except Exception as e:  # noqa: BLE001
            print("bench.py: rank %d of %d could not initialise the %s process group within %d s: %s"
                  % (rank, world, args.dist_backend, args.join_timeout, e), file=sys.stderr)
            sys.exit(5)
        dist = dist_mod
        global COLLECTIVE_DEVICE
        COLLECTIVE_DEVICE = "cuda" if args.dist_backend == "nccl" else "cpu"
    else:
        torch.cuda.set_device(0)
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)

    n = args.num_envs or (65536 if args.env in FRAME_ENVS else 1 << 24)
    # a non-default torch stream: kernels are launched on it through the C ABI's stream
    # argument and the HIP events that time them are recorded on the same stream
    torch.cuda.set_stream(torch.cuda.Stream())
    ctx = Context(local_rank)
    flags = L.AUTORESET | L.RECORD_STATS
    if args.rk4:
        if args.env != "acrobot":
            ap.error("--rk4 applies to --env acrobot")
        flags |= L.ACROBOT_RK4
    if args.slippery:
        flags |= L.SLIPPERY
    if args.taxi_full:
        flags |= L.TAXI_FULL
    if args.blackjack_full:
        flags |= L.BLACKJACK_FULL
    k_roll = max(1, args.rollout)
    if k_roll > 1:
        args.policy = "random"
    mode = {"exact": L.MODE_EXACT, "fast32": L.MODE_FAST32, "fp64": L.MODE_FP64}[args.mode]
    fp64_state = args.mode == "fp64" and args.env in B_ALG_FP64_EXTRA
    # weak scaling: every rank owns n envs of the n * world batch, keyed by global env index
    base, n_local = shard(n * world, rank, world)
    assert n_local == n
    batch = DeviceBatch(ctx, ENV_IDS[args.env], n, seed=0x9E3779B97F4A7C15, flags=flags, mode=mode,
                        env_index_base=base, map_rows=args.map)
    batch.reset()
    stream_ptr = torch.cuda.current_stream().cuda_stream
    view = batch.view()
    out_rows = None
    if k_roll > 1 and args.rollout_out == "all":
        rdt = torch.float64 if batch.info.reward_dtype == L.F64 else torch.float32
        out_rows = (torch.empty((k_roll, int(view.stride)), dtype=rdt, device="cuda"),
                    torch.empty((k_roll, int(view.stride)), dtype=torch.uint8, device="cuda"))
    if k_roll > 1:
        def step_fn():
            # one fused rollout, then the episode statistics reduced on the device and across ranks
            if args.rollout_out == "all":
                batch.rollout_out(k_roll, L.OUT_ALL, L.POLICY_RANDOM, None, out_rows[0].data_ptr(), out_rows[1].data_ptr(), None, stream_ptr)
            elif args.rollout_out == "none":
                batch.rollout_out(k_roll, L.OUT_NONE, L.POLICY_RANDOM, None, None, None, None, stream_ptr)
            else:
                batch.rollout(k_roll, L.POLICY_RANDOM, None, stream_ptr)
            if dist is not None:
                torch.cuda.current_stream().synchronize()
                allreduce_stats(batch.stats(), dist, device=COLLECTIVE_DEVICE)
    elif args.policy == "device":
        step_fn = device_action_ring(batch, stream_ptr)
    else:
        step_fn = lambda: batch.step_random(stream_ptr)  # noqa: E731
    torch.cuda.synchronize()

    # Part of the set-up, like the resident inputs: the batch is stepped out of the state Reset() left it in.  All
    # envs start their first episode together, so for the first tens of steps the share of envs that finish (and
    # with it the cost of the fused auto-reset) swings before it settles at the policy's stationary rate
    # (tools/step_time_series.py).  The W warm-up and K timed steps that follow see that stationary batch.
    for _ in range(args.settle if k_roll == 1 else 0):
        step_fn()
    torch.cuda.synchronize()

    wall, kernel_ms = time_steps(batch, step_fn, args.steps, args.warmup, dist)
    total_steps = n * world * args.steps * k_roll
    value = total_steps / wall

    b_alg = bytes_per_env_step(args.env, args.policy, args.mode, args.rk4, args.taxi_full, args.blackjack_full)
    if k_roll > 1:
        # per env-step of a K-step rollout: the rows are read and written once per launch; the last step's reward /
        # done rows, or every step's
        rew_b = 8 if batch.info.reward_dtype == L.F64 else 4
        b_alg = (b_alg - rew_b - 1) / k_roll + {"last": (rew_b + 1) / k_roll, "all": (rew_b + 1) * (1 + 1 / k_roll), "none": 0.0}[args.rollout_out]
    achieved = b_alg * n * k_roll / (kernel_ms * 1e-3) / 1e9
    if k_roll > 1:
        kind = "fused %d-step rollout kernel (one launch per timed step; reward / done rows handed back: %s)" % (k_roll, args.rollout_out)
    else:
        kind = "single-step kernel"
    name = args.env + (" RK4/fp64" if args.rk4 else "") + (" " + args.map if args.map else "") + \
        (" slippery" if args.slippery else "") + (" full Taxi-v3" if args.taxi_full else "") + (" upstream rules" if args.blackjack_full else "")
    out = {
        "metric": "env-steps/sec", "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64" if args.rk4 or fp64_state else "u8" if args.env in FRAME_ENVS else ("i32" if args.env in ("frozenlake", "taxi", "blackjack") else "f32"),
        "data": "synthetic",
        "config": {
            "workload": "%s VectorEnv.Step, %d envs/GPU, %s, auto-reset + RecordEpisodeStatistics fused, "
                        "%s, %s mode" % (name, n, kind, "device-resident random actions, a ring of %d pre-sampled rows cycled per step (zero-copy path)" % ACTION_RING
                                         if args.policy == "device" else "actions sampled in-kernel", args.mode),
            "envs_per_gpu": n, "sharding": "independent env shards, no data-path collective",
            "settle_steps_after_reset": args.settle if k_roll == 1 else 0,
        },
        # per GPU and per launch on every line: rank 0's kernel on rank 0's shard (the shards are the same size)
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                     "kernel_ms": kernel_ms, "bytes_per_env_step": b_alg,
                     "algorithmic_bytes_per_launch": b_alg * n * k_roll, "scope": "one GPU (rank 0), one launch"},
    }
    if args.rk4 and k_roll > 1:
        # the fused rollout is bound by the FP64 vector ALU, not by HBM: flops as the SQ counters
        # count them per env-step x env-steps per launch / live launch time
        tf = n * k_roll * ACROBOT_RK4_FLOPS_COUNTED / (kernel_ms * 1e-3) / 1e12
        out["roofline"] = {"bound": "fp64-valu", "achieved": tf, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": tf / FP64_VALU_PEAK_TFLOPS, "traffic": None, "kernel_ms": kernel_ms,
                           "fp64_flops_per_env_step": ACROBOT_RK4_FLOPS_COUNTED,
                           "arithmetic_only_flops_per_env_step": ACROBOT_RK4_FLOPS,
                           "source": ACROBOT_RK4_FLOPS_SOURCE, "scope": "one GPU (rank 0), one launch"}
    # HBM bytes per launch from the rocprofv3 counter passes (FETCH_SIZE doubled as gfx950
    # needs, WRITE_SIZE as read) of this same command on one GPU, taken by tools/collect_profiles.sh and
    # committed under profiles/; counters cannot be read from inside this process.  A rank of a larger job
    # launches the same kernel on a shard of the same size, so the figure holds per GPU at every N.
    traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(traffic_file):
        try:
            from gym_amd.build import csrc_hash

            t = json.load(open(traffic_file)).get("%s:%d:%s" % (args.env, n, args.policy))
            if t and not args.rk4 and k_roll == 1 and not (args.slippery or args.taxi_full or args.blackjack_full or args.map or args.mode != "exact"):
                if t.get("csrc_hash") == csrc_hash():
                    out["roofline"]["traffic"] = t["hbm_bytes_per_launch"]
                    out["roofline"]["traffic_source"] = t["source"] + (
                        "; counters taken on one GPU, the kernel and the shard size are the same on every rank" if world > 1 else "")
                elif rank == 0:
                    # counters taken on other kernel sources say nothing about this build
                    print("note: profiles/traffic.json was measured on kernel sources %s, this build is %s; traffic left null"
                          % (t.get("csrc_hash"), csrc_hash()), file=sys.stderr)
        except (OSError, ValueError, KeyError):
            pass

    # episode statistics: device reduction, then one RCCL all-reduce across ranks
    ep = allreduce_stats(batch.stats(), dist, device=COLLECTIVE_DEVICE)
    out["episode_stats"] = {"episodes": ep["episodes"], "mean_length": ep["sum_length"] / max(ep["episodes"], 1),
                            "mean_return": ep["sum_return"] / max(ep["episodes"], 1),
                            "min_return": ep["min_return"], "max_return": ep["max_return"]}
    out["ranks"] = rank_evidence(dist, args.dist_backend, rank, world, local_rank, kernel_ms)
    batch.close()

    if not args.no_extra and args.env == "cartpole":
        # every rank takes part: the timings are maxima over ranks, the rollout all-reduces its statistics
        extra = extra_measurements(ctx, n, stream_ptr, dist, rank, world, args.extra_envs_scale)
        if rank == 0:
            out["extra"] = extra
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    # The CPU legs come last, at every N, on rank 0 alone: the other ranks have left by now, so the host cores
    # are as free as in the one-GPU run.
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.env)
        if args.slippery or args.taxi_full or args.blackjack_full or args.rk4:
            out["cpu_baseline"]["sample"] += "; the CPU loop runs the reference's own rules, not the extension timed on the GPU"
        allc = cpu_baseline_all_cores(args.env)
        if allc:
            out["cpu_baseline_all_cores"] = allc
    result_line.write(json.dumps(out) + "\n")
    result_line.flush()


def claim_stdout():
    """The contract is one JSON line on standard output.  Libraries write there too (gloo announces its connections,
    RCCL logs there when NCCL_DEBUG is set), so from here on file descriptor 1 is standard error for everybody, native
    code included, and the result line goes to the descriptor returned, a duplicate of the original standard output."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    return os.fdopen(saved, "w")


def measure_single_step(ctx, env, n, stream_ptr, dist, rank, world, policy="device", mode="exact", flags_extra=0,
                        map_rows=None, steps=100, settle=100, seed=1):
    """One named workload under the same clock as the headline: n envs per rank, `settle` untimed steps after
    Reset(), `steps` timed between barriers.  Rates are whole-job (all ranks' env-steps over the slowest
    rank's wall time); the kernel time is rank 0's, from HIP events on the launch stream."""
    flags = L.AUTORESET | L.RECORD_STATS | flags_extra
    b = DeviceBatch(ctx, ENV_IDS[env], n, seed=seed, flags=flags, env_index_base=rank * n, map_rows=map_rows,
                    mode={"exact": L.MODE_EXACT, "fast32": L.MODE_FAST32, "fp64": L.MODE_FP64}[mode])
    b.reset()
    step_fn = device_action_ring(b, stream_ptr) if policy == "device" else (lambda: b.step_random(stream_ptr))
    wall, ms = time_steps(b, step_fn, steps, settle, dist)
    b_alg = bytes_per_env_step(env, policy, mode, bool(flags_extra & L.ACROBOT_RK4), bool(flags_extra & L.TAXI_FULL),
                               bool(flags_extra & L.BLACKJACK_FULL))
    gbps = b_alg * n / (ms * 1e-3) / 1e9
    res = {"env_steps_per_s": n * world * steps / wall, "n_gpus": world, "envs_per_gpu": n, "kernel_ms": ms,
           "bytes_per_env_step": b_alg, "hbm_GBps_per_gpu": gbps, "frac": gbps / HBM_PEAK_GBPS,
           "actions": "device row" if policy == "device" else "sampled in-kernel", "mode": mode,
           "untimed_steps_after_reset": settle, "timed_steps": steps}
    del step_fn
    b.close()
    return res


def extra_measurements(ctx, n, stream_ptr, dist, rank, world, scale=1.0):
    """The other workloads the metric and the benchmark configurations name, each timed like the headline
    (barriers, slowest rank, HIP events): CartPole with in-kernel actions, Acrobot single step (reference
    dynamics), the fused 64-step RK4/binary64 rollout (FP64 vector-ALU bound); MountainCar, MountainCarContinuous
    and Pendulum at 16 Mi envs in binary32 and binary64; FrozenLake 8x8 slippery and Taxi (the reference's stub
    and upstream's dynamics) at 64 Mi envs; Pong at 65 536 envs, step + frame raster."""
    res = {}
    if world == 1:
        res["hbm_copy_ceiling_GBps_measured"] = d2d_copy_rate()

    def envs(count):
        return max(4096, int(count * scale) // 4096 * 4096)

    def m(key, *a, **kw):
        res[key] = measure_single_step(ctx, *a, stream_ptr=stream_ptr, dist=dist, rank=rank, world=world, **kw)

    # the headline workload with the actions sampled in the kernel (Spaces.Discrete.Sample per env and step on the
    # device) instead of read from a device row: 45 B per env-step
    m("cartpole_actions_sampled_in_kernel", "cartpole", n, policy="random", settle=150)
    # The cost of a fused step depends on how many envs finish in it (every wave with a finished env runs a reset
    # round).  Acrobot's Reset() fills all six slots with small uniforms (Envs.Acrobot.cs:24-25), i.e. both angles are
    # uniform on the circle, so the first steps after it end a large share of the episodes at once; the rate
    # settles at 0.2-0.4 % of the envs per step after a few hundred steps (tools/step_time_series.py: 0.29 ms
    # per step over the first 50 steps, 0.23-0.24 ms from step 150 on).  Timed in that stationary regime.
    m("acrobot_euler_exact_single_step", "acrobot", n, policy="random", settle=300)
    m("acrobot_euler_exact_single_step_device_actions", "acrobot", n, policy="device", settle=300)
    k = 64
    b = DeviceBatch(ctx, L.ENV_ACROBOT, n, seed=1, flags=L.AUTORESET | L.RECORD_STATS | L.ACROBOT_RK4, env_index_base=rank * n)
    b.reset()

    def rollout():
        # one fused rollout, then the episode statistics reduced on the device and across ranks
        b.rollout(k, L.POLICY_RANDOM, None, stream_ptr)
        if dist is not None:
            torch.cuda.current_stream().synchronize()
            allreduce_stats(b.stats(), dist, device=COLLECTIVE_DEVICE)

    steps = 3
    wall, ms = time_steps(b, rollout, steps, 1, dist)
    rate = n * k / (ms * 1e-3)  # per GPU, from the kernel time
    res["acrobot_rk4_fp64_rollout64"] = {"env_steps_per_s": n * world * k * steps / wall, "n_gpus": world,
                                         "env_steps_per_s_per_gpu_kernel": rate, "kernel_ms": ms,
                                         "fp64_TFLOPs": rate * ACROBOT_RK4_FLOPS_COUNTED / 1e12,
                                         "frac_of_fp64_valu_peak": rate * ACROBOT_RK4_FLOPS_COUNTED / 1e12 / FP64_VALU_PEAK_TFLOPS,
                                         "fp64_TFLOPs_arith_only": rate * ACROBOT_RK4_FLOPS / 1e12,
                                         "stats_allreduced_after_every_rollout": dist is not None}
    b.close()
    # classic control, 16 Mi envs per GPU, binary32 (EXACT) and binary64 state
    for env in ("mountaincar", "mountaincarcontinuous", "pendulum"):
        for mode in ("exact", "fp64"):
            m("%s_%s" % (env, mode), env, n, mode=mode)
    # table-driven envs, 64 Mi envs per GPU, tables staged in LDS
    big = envs(4 * n)
    for policy, tag in (("device", "device_actions"), ("random", "in_kernel_actions")):
        m("frozenlake_8x8_slippery_64Mi_" + tag, "frozenlake", big, policy=policy, flags_extra=L.SLIPPERY, map_rows="8x8")
        m("taxi_stub_64Mi_" + tag, "taxi", big, policy=policy)
        m("taxi_v3_full_64Mi_" + tag, "taxi", big, policy=policy, flags_extra=L.TAXI_FULL)
    # Pong: game step + 210x160x3 frame raster per env-step
    m("pong_step_and_raster", "pong", envs(65536), settle=20)
    res["cartpole_1024_host_path"] = measure_host_path(ctx, 1024)
    return res


def measure_host_path(ctx, n, steps=2000, warmup=200):
    """The reference's own CPU-runnable configuration (CartPole, N = 1024, random actions from the host, auto-reset)
    through the host path of the boundary: gx_vec_step_async + gx_vec_wait per step with pinned buffers -- actions over
    PCIe in, observation, reward and done out.  Wall clock on this rank, so the rate includes the copies and the
    launch: at this size a step is latency, not bandwidth."""
    import ctypes as C

    b = DeviceBatch(ctx, L.ENV_CARTPOLE, n, seed=1, flags=L.AUTORESET | L.RECORD_STATS)
    b.reset()
    lib = L.load()
    act = torch.randint(0, 2, (n,), dtype=torch.int32).pin_memory()
    rew = torch.empty(n, dtype=torch.float32).pin_memory()
    done = torch.empty(n, dtype=torch.uint8).pin_memory()
    obs = torch.empty((4, n), dtype=torch.float32).pin_memory()
    vp = lambda t: C.c_void_p(t.data_ptr())

    def step():
        L.check(lib.gx_vec_step_async(b._h, vp(act), vp(obs), vp(rew), vp(done), None))
        L.check(lib.gx_vec_wait(b._h))

    for _ in range(warmup):
        step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = (time.perf_counter() - t0) / steps
    b.close()
    return {"envs": n, "steps": steps, "us_per_step": dt * 1e6, "env_steps_per_s": n / dt,
            "includes": "PCIe copies of actions in and observation, reward, done out; launch and wait per step"}


if __name__ == "__main__":
    main()
