"""Where the PCIe-inclusive time of the host-pointer entry points goes: fresh against reused output buffers, chunk sizes.

    python tools/host_path_probe.py
"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import forge_ec_amd as F  # noqa: E402
from forge_ec_amd import synth as V  # noqa: E402
from forge_ec_amd._lib import lib  # noqa: E402

n = 1 << 20
ctx = F.Context(0)
L = lib()


def ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def run(label, fn, out_words, reps=3):
    warm = np.zeros((n, out_words), dtype=np.uint64)
    for mode in ("fresh", "reused"):
        best = 1e9
        for _ in range(reps):
            out = np.empty((n, out_words), dtype=np.uint64) if mode == "fresh" else warm
            t0 = time.perf_counter()
            rc = fn(out)
            dt = time.perf_counter() - t0
            assert rc == 0, rc
            best = min(best, dt)
        print("%-34s out %-6s : %6.2f ms" % (label, mode, best * 1e3), flush=True)


for curve, name in ((0, "secp256k1"), (2, "ed25519")):
    k, p = V.scalars(n, curve, 1), V.points(n, curve, 2)
    pl = p.shape[1]
    ctx.batch_mul(curve, k[:1024], p[:1024])
    for logc in (16, 17, 18, 19, 20):
        ctx.set_chunk(1 << logc)
        run("%s var chunk 2^%d" % (name, logc), lambda out: L.fec_batch_mul(ctx._h, curve, ptr(k), ptr(p), ptr(out), n), pl)
g = ctx.generator(2)
k = V.scalars(n, 2, 3)
for logc in (15, 16, 17, 18, 19, 20):
    ctx.set_chunk(1 << logc)
    run("ed25519 fixed chunk 2^%d" % logc, lambda out: L.fec_batch_mul_fixed(ctx._h, 2, ptr(k), ptr(g), ptr(out), n), 16)
