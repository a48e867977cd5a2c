"""A plain-C program (examples/c_abi_demo.c) is a consumer of the C ABI: include/qprop.h must be
valid C11 on its own and the library must link without Python / torch.  The program is run on the GPU
by tests/test_00_multirank_gpu.py::test_c_consumer_runs (a child process: it has to start before this
process touches the GPU)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "quantumpropagators.jl_amd", "lib")


def _build(tmp_path):
    exe = str(tmp_path / "c_abi_demo")
    cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "c_abi_demo.c"), "-o", exe, "-L" + LIBDIR, "-lqprop_hip",
           "-Wl,-rpath," + LIBDIR, "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_plain_c_and_library_links(tmp_path):
    if not os.path.exists(os.path.join(LIBDIR, "libqprop_hip.so")):
        pytest.skip("library not built")
    exe = _build(tmp_path)
    out = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libqprop_hip.so" in out and "libtorch" not in out and "libpython" not in out
