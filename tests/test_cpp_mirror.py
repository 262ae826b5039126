"""The C++ host mirror (include/sla_hip.hpp) compiles against the C ABI with plain g++ (CPU check) and
reproduces the reference's solver tests on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "reference_cases")


def build():
    import __graft_entry__ as g
    g.build()
    lib = os.path.join(ROOT, "sparse-linear-algebra_amd", "lib")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", f"-I{ROOT}/include", os.path.join(ROOT, "examples", "reference_cases.cpp"),
           f"-L{lib}", "-lsla_hip", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib", "-o", EXE]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert out.returncode == 0, out.stdout


def test_cpp_mirror_compiles_and_links():
    build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cpp_mirror_reference_cases_on_gpu():
    build()
    out = subprocess.run([EXE], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert out.returncode == 0 and "all passed" in out.stdout, out.stdout
