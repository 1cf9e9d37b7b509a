"""Build libfecgpu.so (the HIP kernels + C ABI) in-tree for gfx950.

    python -m forge_ec_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with gpurun snapshots.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libfecgpu.so")
SOURCES = ["fecgpu.hip"]
DEPS = ["fecgpu.hip", "limbs.hpp", "secp256k1.hpp", "p256.hpp", "ed25519.hpp", "canon_curves.hpp",
        "canon_kernels.hpp", os.path.join("..", "..", "include", "fecgpu.h"),
        os.path.join("..", "..", "include", "fecgpu_canon.h")]


def _stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(os.path.join(SRC_DIR, d)) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not _stale():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-o", SO] + [os.path.join(SRC_DIR, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
