"""include/gymx.hpp, the C++ host facade over the C ABI: tests/cpp/host_facade_test.cpp restates the
reference's unit tests, the VectorEnv contract and the wrappers in C++ and replays a CartPole batch
against the oracle.  Without a GPU the same binary must fail loudly (exit status 77), never fall back."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "cpp"))


def _exe():
    import facade_build as cpp_build  # tests/cpp/facade_build.py

    return cpp_build.build()


def test_facade_builds_and_refuses_to_run_without_a_gpu():
    exe = _exe()
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu test")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 77 and "no GPU" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_facade_on_device():
    r = subprocess.run([_exe()], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout + r.stderr
