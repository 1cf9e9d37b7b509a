"""The plain-C consumer of the ABI (examples/c_abi_example.c): it must compile and link with gcc against
the two public headers (CPU check) and run correctly on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "c_abi_example")


def _build():
    from forge_ec_amd import build
    build.build()
    src = os.path.join(ROOT, "examples", "c_abi_example.c")
    deps = [src, os.path.join(ROOT, "include", "fecgpu.h"), os.path.join(ROOT, "include", "fecgpu_canon.h")]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-I", os.path.join(ROOT, "include"), src,
                               "-L", os.path.join(ROOT, "forge_ec_amd"), "-lfecgpu",
                               "-Wl,-rpath,$ORIGIN/../forge_ec_amd", "-Wl,-rpath-link,/opt/rocm/lib", "-o", EXE])
    return EXE


def test_c_example_compiles_as_c11():
    assert os.path.exists(_build())


def test_c_example_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([_build()], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "no usable gfx950 GPU" in r.stdout


@pytest.mark.gpu
def test_c_example_runs_on_the_gpu():
    r = subprocess.run([_build()], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "c abi example ok" in r.stdout
