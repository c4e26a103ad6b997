"""-m gpu: build tests/abi_client.c with plain gcc against include/mxlo.h + libmxlo.so and run it — the C ABI
used from a foreign language with no Python / torch in the process (what a Julia `ccall` glue would do)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_client(tmp_path):
    exe = str(tmp_path / "abi_client")
    libdir = os.path.join(ROOT, "linearoperators.jl_amd", "csrc")
    cmd = ["gcc", "-std=c99", "-O1", os.path.join(ROOT, "tests", "abi_client.c"), "-I", os.path.join(ROOT, "include"),
           "-L", libdir, "-lmxlo", "-lm", f"-Wl,-rpath,{libdir}", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ABI CLIENT OK" in out.stdout, out.stdout + out.stderr
