"""bench.py prints one JSON line with the agreed fields (small sizes; the real run uses the defaults)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "1"] + list(args),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    # the result line and nothing else on standard output, whatever the libraries below announce (gloo does)
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_default_workload_line():
    j = _run("--num-envs", "262144", "--no-extra")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["metric"] == "env-steps/sec" and j["n_gpus"] == 1 and j["steps"] == 5 and j["warmup"] == 1
    assert j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"]
    roof = j["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12 and roof["bytes_per_env_step"] == 49
    # value and the kernel time describe the same run
    assert abs(j["value"] - 262144 * 5 / (j["ms_per_step"] * 5e-3)) / j["value"] < 1e-6
    cpu = j["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["value"] > 0 and "sample" in cpu
    assert j["cpu_baseline_all_cores"]["cores"] >= 1


@pytest.mark.gpu
def test_rollout_workload_line():
    j = _run("--env", "acrobot", "--rk4", "--rollout", "8", "--num-envs", "65536", "--no-cpu-baseline")
    assert j["dtype"] == "f64" and j["roofline"]["bound"] == "fp64-valu" and j["roofline"]["unit"] == "TFLOP/s"
    assert abs(j["value"] - 65536 * 8 * 5 / (j["ms_per_step"] * 5e-3)) / j["value"] < 1e-6


@pytest.mark.gpu
def test_plain_launch_with_two_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts its own two ranks (here both on the one GPU of
    the test box, collectives over gloo) and prints one line for the whole job"""
    j = _run("--gpus", "2", "--share-device", "--dist-backend", "gloo", "--num-envs", "262144", "--no-extra")
    assert j["n_gpus"] == 2 and j["scaling"] == "weak"
    # the CPU leg and the per-GPU roofline are on the line at every N, and the line shows its ranks
    assert j["cpu_baseline"]["cores"] == 1 and j["cpu_baseline"]["value"] > 0
    assert "traffic" in j["roofline"] and j["roofline"]["scope"].startswith("one GPU")
    rk = j["ranks"]
    assert rk["world"] == 2 and rk["backend"] == "gloo" and len(rk["per_rank_kernel_ms"]) == 2 and rk["allreduce_us"] > 0
    assert sorted(e["rank"] for e in rk["per_rank"]) == [0, 1] and len({e["pid"] for e in rk["per_rank"]}) == 2
    # whole-job value: both ranks' envs over the slowest rank's time
    assert abs(j["value"] - 2 * 262144 * 5 / (j["ms_per_step"] * 5e-3)) / j["value"] < 1e-6
    assert j["episode_stats"]["episodes"] >= 0  # six steps: no CartPole episode is over yet


@pytest.mark.gpu
def test_rccl_path_with_one_rank():
    """the code a multi-GPU run takes -- process group over RCCL bound to the rank's device, barriers around the timed
    region, max-over-ranks of the time -- with a world of one"""
    j = _run("--force-dist", "--num-envs", "262144", "--no-extra", "--no-cpu-baseline")
    assert j["n_gpus"] == 1 and j["value"] > 0


@pytest.mark.gpu
def test_launched_by_torch_distributed_run_with_two_ranks():
    """the launch line a multi-GPU run uses (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`): ranks come from the environment, bench.py starts
    none of its own, rank 0 prints the one line.  Both ranks share the test box's GPU, collectives over gloo."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "5", "--warmup", "1", "--share-device", "--dist-backend", "gloo",
                        "--num-envs", "262144", "--no-extra"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()  # (torch.distributed.run itself writes its notices to standard error)
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak"
    assert abs(j["value"] - 2 * 262144 * 5 / (j["ms_per_step"] * 5e-3)) / j["value"] < 1e-6


@pytest.mark.gpu
def test_a_rank_that_dies_at_start_up_ends_the_job_quickly():
    """rank 1 leaves before it joins anything: the launcher sees it, gives rank 0 its grace period and ends it; the
    whole call returns rank 1's exit code well within a minute instead of waiting for a collective's timeout"""
    import time

    env = dict(os.environ, GYMX_BENCH_FAIL_RANK="1", GYMX_BENCH_GRACE_S="10")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1",
                        "--share-device", "--dist-backend", "gloo", "--num-envs", "262144", "--no-extra"],
                       capture_output=True, text=True, timeout=300, env=env)
    took = time.time() - t0
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert took < 60.0, took
    assert "rank 1 exited with code 3" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.gpu
def test_roll_call_alone_names_the_missing_rank():
    """the same failure without the launcher's help (as under another launcher that does not watch its children):
    rank 0 names the rank that never came when its own timeout runs out"""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1",
                        "--share-device", "--dist-backend", "gloo", "--num-envs", "262144", "--no-extra", "--join-timeout", "3"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 4 and "ranks [1] of 2 did not join within 3 s" in r.stderr, r.stderr[-2000:]


EXTRA_KEYS = ["cartpole_actions_sampled_in_kernel", "acrobot_euler_exact_single_step", "acrobot_euler_exact_single_step_device_actions",
              "acrobot_rk4_fp64_rollout64",
              "mountaincar_exact", "mountaincar_fp64", "mountaincarcontinuous_exact", "mountaincarcontinuous_fp64",
              "pendulum_exact", "pendulum_fp64",
              "frozenlake_8x8_slippery_64Mi_device_actions", "frozenlake_8x8_slippery_64Mi_in_kernel_actions",
              "taxi_stub_64Mi_device_actions", "taxi_stub_64Mi_in_kernel_actions",
              "taxi_v3_full_64Mi_device_actions", "taxi_v3_full_64Mi_in_kernel_actions", "pong_step_and_raster"]


@pytest.mark.gpu
def test_extra_carries_every_named_configuration():
    """the secondary measurements (every workload the benchmark configurations name) at a hundredth of their size"""
    j = _run("--num-envs", "262144", "--no-cpu-baseline", "--extra-envs-scale", "0.0625")
    ex = j["extra"]
    for key in EXTRA_KEYS:
        assert key in ex, key
        assert ex[key]["kernel_ms"] > 0 and ex[key]["env_steps_per_s"] > 0
        if key != "acrobot_rk4_fp64_rollout64":
            e = ex[key]
            assert abs(e["frac"] - e["bytes_per_env_step"] * e["envs_per_gpu"] / (e["kernel_ms"] * 1e-3) / 8e12) < 1e-9
    assert ex["mountaincar_exact"]["bytes_per_env_step"] == 33 and ex["mountaincar_fp64"]["bytes_per_env_step"] == 49
    assert ex["pendulum_exact"]["bytes_per_env_step"] == 61 and ex["pendulum_fp64"]["bytes_per_env_step"] == 85
    assert ex["mountaincarcontinuous_exact"]["bytes_per_env_step"] == 53
    assert ex["frozenlake_8x8_slippery_64Mi_in_kernel_actions"]["bytes_per_env_step"] == 21
    assert ex["taxi_v3_full_64Mi_device_actions"]["bytes_per_env_step"] == 33
    assert ex["taxi_stub_64Mi_device_actions"]["envs_per_gpu"] == 65536
    assert ex["pong_step_and_raster"]["bytes_per_env_step"] == 100937 and ex["pong_step_and_raster"]["envs_per_gpu"] == 4096
    host = ex["cartpole_1024_host_path"]  # the reference's own CPU-runnable configuration through the host path
    assert host["envs"] == 1024 and host["us_per_step"] > 0 and abs(host["env_steps_per_s"] - 1024 / (host["us_per_step"] * 1e-6)) < 1.0
