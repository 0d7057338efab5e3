"""The C-ABI library loads without a GPU, exports every symbol include/gymx.h declares, and
reports errors through status codes (no compute calls here)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from gym_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "gymx.h")).read()


def declared_functions():
    # `int gx_name(` / `uint32_t gx_name(` / `const char* gx_name(` at the start of a line
    return sorted(set(re.findall(r"^(?:int|uint32_t|const char\*)\s+(gx_\w+)\s*\(", HEADER, re.M)))


def test_library_exports_every_declared_symbol():
    lib = L.load()
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libgymx.so does not export %s" % n
    assert set(L.SIGNATURES) == set(names), set(L.SIGNATURES) ^ set(names)


def test_enum_values_in_sync(oracle):
    def enum(name):
        m = re.search(r"\b%s\s*=\s*(\d+)u?" % name, HEADER)
        assert m, name
        return int(m.group(1))

    for py, c in [("ENV_CARTPOLE", "GX_ENV_CARTPOLE"), ("ENV_MOUNTAINCAR", "GX_ENV_MOUNTAINCAR"),
                  ("ENV_MOUNTAINCAR_CONT", "GX_ENV_MOUNTAINCAR_CONT"), ("ENV_PENDULUM", "GX_ENV_PENDULUM"),
                  ("ENV_ACROBOT", "GX_ENV_ACROBOT"), ("ENV_FROZENLAKE", "GX_ENV_FROZENLAKE"), ("ENV_TAXI", "GX_ENV_TAXI"),
                  ("ENV_CLIFFWALKING", "GX_ENV_CLIFFWALKING"), ("ENV_PONG", "GX_ENV_PONG"), ("ENV_DUMMY", "GX_ENV_DUMMY"),
                  ("ENV_BREAKOUT", "GX_ENV_BREAKOUT"), ("ENV_SPACEINVADERS", "GX_ENV_SPACEINVADERS"),
                  ("ENV_BLACKJACK", "GX_ENV_BLACKJACK"), ("ENV_TABLE", "GX_ENV_TABLE"), ("BLACKJACK_FULL", "GX_BLACKJACK_FULL"),
                  ("AUTORESET", "GX_AUTORESET"), ("RECORD_STATS", "GX_RECORD_STATS"), ("SLIPPERY", "GX_SLIPPERY"),
                  ("TAXI_FULL", "GX_TAXI_FULL"), ("ACROBOT_RK4", "GX_ACROBOT_RK4"), ("GX_ERR_INVALID", "GX_ERR_INVALID"),
                  ("GX_ERR_RCCL", "GX_ERR_RCCL"), ("POLICY_RANDOM", "GX_POLICY_RANDOM"), ("MODE_FAST32", "GX_MODE_FAST32")]:
        assert getattr(L, py) == enum(c), py
    # the oracle numbers envs and flags the same way
    for name in ("CARTPOLE", "MOUNTAINCAR", "MOUNTAINCAR_CONT", "PENDULUM", "ACROBOT", "FROZENLAKE", "TAXI",
                 "CLIFFWALKING", "PONG", "DUMMY", "BREAKOUT", "SPACEINVADERS", "BLACKJACK", "TABLE"):
        assert getattr(oracle, name) == getattr(L, "ENV_" + name)
    for name in ("AUTORESET", "RECORD_STATS", "SLIPPERY", "TAXI_FULL", "ACROBOT_RK4", "BLACKJACK_FULL"):
        assert getattr(oracle, name) == getattr(L, name)


def test_struct_sizes_match_header():
    # natural alignment on x86-64 / the layout a C# [StructLayout(Sequential)] mirror gets
    assert C.sizeof(L.GxDesc) == 96 and L.GxDesc.table.offset == 56 == L.DESC_SIZE_V1
    assert int(re.search(r"#define GX_DESC_SIZE_V1 (\d+)u", HEADER).group(1)) == L.DESC_SIZE_V1
    assert C.sizeof(L.GxTransition) == 12
    assert C.sizeof(L.GxStats) == 48
    assert C.sizeof(L.GxView) == 64
    assert C.sizeof(L.GxInfo) == 72


def test_version_and_error_paths_without_gpu():
    lib = L.load()
    assert lib.gx_version() == 0x00010000
    assert lib.gx_ctx_create(0, None) == L.GX_ERR_INVALID
    assert b"null" in lib.gx_last_error()
    assert lib.gx_vec_reset(None) == L.GX_ERR_INVALID
    assert lib.gx_vec_stats(None, None) == L.GX_ERR_INVALID
    assert lib.gx_vec_destroy(None) == L.GX_OK and lib.gx_ctx_destroy(None) == L.GX_OK
    n = C.c_int(-1)
    rc = lib.gx_device_count(C.byref(n))
    if rc != L.GX_OK:  # no GPU in this container: the product path fails loudly, it does not fall back
        assert rc == L.GX_ERR_HIP and n.value == 0
        h = C.c_void_p()
        assert lib.gx_ctx_create(0, C.byref(h)) in (L.GX_ERR_HIP, L.GX_ERR_INVALID) and not h.value


def test_product_package_never_imports_the_oracle():
    """nothing under gym_amd/ or include/ may reach into oracle/"""
    for d in ("gym_amd", "include"):
        for root, _, files in os.walk(os.path.join(ROOT, d)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                    text = open(os.path.join(root, f)).read()
                    assert "oracle" not in text.lower().replace("# the checker", ""), os.path.join(root, f)


def test_device_math_header_against_libm(tmp_path):
    """gym_amd/csrc/gx_math.h compiled for the host: the reduction-light sin/cos and the
    constant division agree with glibc after rounding to float (sampled; the same program
    with stride 1 checks every float)."""
    exe = str(tmp_path / "host_math_check")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-mfma", "-std=c++17",
                           os.path.join(ROOT, "tests", "host_math_check.cpp"), "-o", exe])
    out = subprocess.run([exe, "257"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "div_by_total_mass" in out.stdout and "mismatches 0" in out.stdout.splitlines()[-1]


def test_header_is_plain_c_and_struct_sizes_hold(tmp_path):
    """include/gymx.h is a C header (what a cgo / P/Invoke / ctypes author reads): it compiles as strict C99 on its
    own, and the sizes and offsets the bindings mirror are what the compiler lays out"""
    src = tmp_path / "abi.c"
    src.write_text("""
#include <stddef.h>
#include "gymx.h"
#define SA(c, n) typedef char static_assert_##n[(c) ? 1 : -1]
SA(sizeof(gx_desc) == 96, desc);
SA(offsetof(gx_desc, table) == GX_DESC_SIZE_V1, desc_v1);
SA(offsetof(gx_desc, map_rows) == 48, desc_map);
SA(sizeof(gx_transition) == 12, transition);
SA(sizeof(gx_info) == 72, info);
SA(sizeof(gx_view) == 64, view);
SA(sizeof(gx_stats) == 48, stats);
SA(sizeof(gx_rows) == %d, rows);
SA(GX_ENV_COUNT == 14 && GX_ENV_TABLE == 13, envs);
int main(void) { gx_desc d = {0}; d.struct_size = (uint32_t)sizeof d; return (int)d.struct_size - 96; }
""" % C.sizeof(L.GxRows))
    exe = str(tmp_path / "abi")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src), "-o", exe])
    assert subprocess.run([exe]).returncode == 0
