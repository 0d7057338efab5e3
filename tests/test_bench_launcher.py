"""bench.py's own N-rank launcher and start-up roll call, without a GPU: a rank that dies at start-up must end the
job quickly and be named, whichever way the ranks were started."""
import os
import subprocess
import sys
import textwrap
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_launcher_relays_rank0_and_ends_when_another_rank_dies(tmp_path, capfd):
    import bench

    script = tmp_path / "ranks.py"
    script.write_text(textwrap.dedent("""
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(3)
        print("line from rank 0", flush=True)
        time.sleep(300)        # stands for a rank stuck in a rendezvous or barrier
    """))
    t0 = time.time()
    rc = bench.spawn_ranks(2, [], script=str(script), grace=1.0)
    took = time.time() - t0
    out, err = capfd.readouterr()
    assert rc == 3 and took < 30.0
    assert "line from rank 0" in out
    assert "rank 1 exited with code 3" in err and "terminating ranks [0]" in err


def test_launcher_kills_a_rank_that_ignores_sigterm(tmp_path, capfd):
    import bench

    script = tmp_path / "ranks.py"
    script.write_text(textwrap.dedent("""
        import os, signal, sys, time
        if os.environ["RANK"] == "0":
            sys.exit(7)
        signal.signal(signal.SIGTERM, signal.SIG_IGN)
        time.sleep(300)
    """))
    t0 = time.time()
    rc = bench.spawn_ranks(3, [], script=str(script), grace=0.5)
    assert rc == 7 and time.time() - t0 < 40.0


def test_launcher_returns_zero_and_all_output_when_every_rank_ends_well(tmp_path, capfd):
    import bench

    script = tmp_path / "ranks.py"
    script.write_text(textwrap.dedent("""
        import os
        assert os.environ["WORLD_SIZE"] == "3" and os.environ["MASTER_ADDR"] == "127.0.0.1"
        print('{"rank": %s}' % os.environ["RANK"])
    """))
    assert bench.spawn_ranks(3, [], script=str(script), grace=1.0) == 0
    out, _ = capfd.readouterr()
    assert out.strip() == '{"rank": 0}'   # only rank 0's output is relayed


ROLL_CALL = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import bench, torch, datetime
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    store = bench.join_ranks(rank, world, int(os.environ.get("JOIN_TIMEOUT", "60")))
    dist.init_process_group("gloo", store=store, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    assert t.item() == world * (world + 1) / 2
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("roll call ok", world)
""" % ROOT)


def test_roll_call_names_the_rank_that_never_came(tmp_path):
    script = tmp_path / "rollcall.py"
    script.write_text(ROLL_CALL)
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), JOIN_TIMEOUT="2")
    t0 = time.time()
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 4, r.stderr[-2000:]
    assert "ranks [1] of 2 did not join within 2 s" in r.stderr
    assert time.time() - t0 < 60.0


def test_roll_call_then_process_group_with_own_launcher(tmp_path, capfd):
    import bench

    script = tmp_path / "rollcall.py"
    script.write_text(ROLL_CALL)
    assert bench.spawn_ranks(2, [], script=str(script), grace=5.0) == 0
    out, _ = capfd.readouterr()
    assert "roll call ok 2" in out


def test_roll_call_under_torch_distributed_run(tmp_path):
    """there the launcher's agent already serves the store on MASTER_PORT and rank 0 is a client like the others"""
    script = tmp_path / "rollcall.py"
    script.write_text(ROLL_CALL)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "roll call ok 2" in r.stdout
