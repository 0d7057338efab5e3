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
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
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
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and "cpu_baseline" not in j
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
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak"
    assert abs(j["value"] - 2 * 262144 * 5 / (j["ms_per_step"] * 5e-3)) / j["value"] < 1e-6
