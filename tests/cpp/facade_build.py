"""Builds tests/cpp/host_facade_test.cpp against libgymx.so and the oracle (g++, no GPU needed).
The binary lands in tests/cpp/_build/ with run paths relative to itself, so it travels with the tree."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
EXE = os.path.join(HERE, "_build", "host_facade_test")


def build():
    src = os.path.join(HERE, "host_facade_test.cpp")
    deps = [src, os.path.join(ROOT, "include", "gymx.hpp"), os.path.join(ROOT, "include", "gymx.h"),
            os.path.join(ROOT, "gym_amd", "lib", "libgymx.so"), os.path.join(ROOT, "oracle", "_build", "libgym_oracle.so")]
    if os.path.exists(EXE) and os.path.getmtime(EXE) >= max(os.path.getmtime(d) for d in deps):
        return EXE
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    subprocess.check_call([
        "g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"), "-I" + ROOT, src, "-o", EXE,
        "-L" + os.path.join(ROOT, "gym_amd", "lib"), "-lgymx", "-L" + os.path.join(ROOT, "oracle", "_build"), "-lgym_oracle",
        "-L/opt/rocm/lib", "-Wl,-rpath,$ORIGIN/../../../gym_amd/lib", "-Wl,-rpath,$ORIGIN/../../../oracle/_build",
        "-Wl,-rpath,/opt/rocm/lib"])
    return EXE


if __name__ == "__main__":
    print(build())
