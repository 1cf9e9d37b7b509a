"""Build libfecgpu.so (the HIP kernels + C ABI) in-tree for gfx950.

    python -m forge_ec_amd.build [--force]

hipcc cross-compiles without a GPU.  The translation units are compiled in parallel and linked into
one shared library.  The .so is git-ignored but travels with gpurun snapshots; whether it is up to
date is decided by a hash of the sources stored beside it (file times do not survive a copy to
another machine).
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = os.path.join(HERE, "csrc")
OBJ_DIR = os.path.join(SRC_DIR, "_obj")
SO = os.path.join(HERE, "libfecgpu.so")
STAMP = SO + ".sha256"
SOURCES = ["fecgpu.hip", "canon.hip", "kernels_p256.hip", "kernels_ed.hip", "kernels_secp.hip", "kernels_codec.hip", "kernels_ecdsa.hip"]
HEADERS = ["coop.hpp", "sched_lf.hpp", "limbs.hpp", "staging.hpp", "host_ctx.hpp", "secp256k1.hpp", "p256.hpp", "ed25519.hpp",
           "canon_curves.hpp", "canon_kernels.hpp", "field_asm.inc", "kernels.hpp", "cu_split.hpp", os.path.join("..", "..", "include", "fecgpu.h"),
           os.path.join("..", "..", "include", "fecgpu_canon.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def source_hash():
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for rel in SOURCES + HEADERS:
        with open(os.path.join(SRC_DIR, rel), "rb") as f:
            h.update(rel.encode() + b"\0" + f.read())
    return h.hexdigest()


def up_to_date():
    try:
        return os.path.exists(SO) and open(STAMP).read().strip() == source_hash()
    except OSError:
        return False


DECLARATION_ONLY = ("kernels.hpp",)  # launcher prototypes: no effect on any kernel's code


def tu_closure_hash(src, kernel_code_only=False):
    """Hash of one translation unit and of exactly the files it #includes from csrc/ (transitively), plus
    the flags: what determines the code of the kernels defined in `src`.  A committed PMC pass stays
    attributable to a kernel for as long as this hash stands, whatever else changes in the library."""
    import re
    seen, todo = [], [src]
    while todo:
        rel = todo.pop()
        if rel in seen or not os.path.exists(os.path.join(SRC_DIR, rel)) or (kernel_code_only and rel in DECLARATION_ONLY):
            continue
        seen.append(rel)
        with open(os.path.join(SRC_DIR, rel), "r") as f:
            todo += re.findall(r'^\s*#\s*include\s+"([^"/]+)"', f.read(), flags=re.M)
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for rel in sorted(seen):
        with open(os.path.join(SRC_DIR, rel), "rb") as f:
            h.update(rel.encode() + b"\0" + f.read())
    return h.hexdigest()


# the translation unit that defines the dominant kernel of each bench workload
WORKLOAD_TU = {"secp256k1-var": "kernels_secp.hip", "secp256k1-fixed": "kernels_secp.hip",
               "p256-var": "kernels_p256.hip", "p256-fixed": "kernels_p256.hip",
               "ed25519-var": "kernels_ed.hip", "ed25519-fixed": "kernels_ed.hip", "secp256k1-double": "kernels_secp.hip"}


def _tu_hash(src):
    """Hash of one translation unit: its own source, the headers it includes (transitively) and the flags."""
    return tu_closure_hash(src)


def build(force=False, verbose=False):
    if not force and up_to_date():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ_DIR, exist_ok=True)
    for stale in (SO, STAMP):  # never leave a library that no longer matches the sources
        if os.path.exists(stale):
            os.remove(stale)
    jobs, objs, stamps = [], [], []
    for src in SOURCES:
        obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        objs.append(obj)
        ostamp, want = obj + ".sha256", _tu_hash(src)
        try:  # an object whose translation unit did not change is reused (the TUs compile in ~2 min)
            if not force and os.path.exists(obj) and open(ostamp).read().strip() == want:
                continue
        except OSError:
            pass
        if os.path.exists(ostamp):
            os.remove(ostamp)
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(SRC_DIR, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        jobs.append((cmd, subprocess.Popen(cmd)))
        stamps.append((ostamp, want))
    failed = [(cmd, proc.returncode) for cmd, proc in jobs if proc.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(failed[0][1], failed[0][0])
    for ostamp, want in stamps:
        with open(ostamp, "w") as f:
            f.write(want + "\n")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(source_hash() + "\n")
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
