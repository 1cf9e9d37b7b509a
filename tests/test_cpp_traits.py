"""The C++ mirror of the trait surface (include/forge_ec_gpu.hpp): it must compile against the C
ABI (CPU check), and its reference-style test program must pass on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_traits")


def _build():
    from forge_ec_amd import build
    build.build()
    src = os.path.join(ROOT, "tests", "cpp", "test_traits.cpp")
    deps = [src, os.path.join(ROOT, "include", "forge_ec_gpu.hpp"), os.path.join(ROOT, "include", "fecgpu.h")]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), src,
                               "-L", os.path.join(ROOT, "forge_ec_amd"), "-lfecgpu",
                               "-Wl,-rpath,$ORIGIN/../../forge_ec_amd", "-Wl,-rpath-link,/opt/rocm/lib", "-o", EXE])
    return EXE


def test_cpp_mirror_compiles_and_links():
    assert os.path.exists(_build())


@pytest.mark.gpu
def test_cpp_reference_style_tests_pass_on_gpu():
    exe = _build()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all trait-surface checks passed" in r.stdout
