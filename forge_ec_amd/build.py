"""Build libfecgpu.so (the HIP kernels + C ABI) in-tree for gfx950.

    python -m forge_ec_amd.build [--force]

hipcc cross-compiles without a GPU.  The translation units are compiled in parallel and linked into
one shared library.  The .so is git-ignored but travels with gpurun snapshots.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = os.path.join(HERE, "csrc")
OBJ_DIR = os.path.join(SRC_DIR, "_obj")
SO = os.path.join(HERE, "libfecgpu.so")
SOURCES = ["fecgpu.hip", "canon.hip"]
HEADERS = ["limbs.hpp", "staging.hpp", "host_ctx.hpp", "secp256k1.hpp", "p256.hpp", "ed25519.hpp",
           "canon_curves.hpp", "canon_kernels.hpp", os.path.join("..", "..", "include", "fecgpu.h"),
           os.path.join("..", "..", "include", "fecgpu_canon.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def _mtime(rel):
    return os.path.getmtime(os.path.join(SRC_DIR, rel))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(_mtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ_DIR, exist_ok=True)
    jobs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + HEADERS):
            cmd = [hipcc] + FLAGS + ["-c", os.path.join(SRC_DIR, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            jobs.append((cmd, subprocess.Popen(cmd)))
    failed = [(cmd, proc.returncode) for cmd, proc in jobs if proc.wait() != 0]
    if failed:
        if os.path.exists(SO):
            os.remove(SO)  # never leave a library that no longer matches the sources
        raise subprocess.CalledProcessError(failed[0][1], failed[0][0])
    if jobs or not os.path.exists(SO) or any(os.path.getmtime(o) > os.path.getmtime(SO) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
