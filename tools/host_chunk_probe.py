"""Host-pointer fec_batch_mul (variable base, 2^20 elements, reused output array, PCIe included) against the pipeline's
chunk size, powers of two and the sizes that fill the kernels' in-flight capacity exactly: 196 608 = 256 CUs x 3
workgroups x 256 lanes (secp256k1 ladder), 212 992 = 256 CUs x 832 slots (P-256 / Ed25519 schedulers).

    python tools/host_chunk_probe.py
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


for curve, name in ((0, "secp256k1"), (1, "p256"), (2, "ed25519")):
    k, p = V.scalars(n, curve, 1), V.points(n, curve, 2)
    out = np.zeros_like(p)
    ctx.batch_mul(curve, k[:1024], p[:1024])
    for chunk in (1 << 17, 196608, 212992, 1 << 18, 393216, 425984, 1 << 19):
        ctx.set_chunk(chunk)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            rc = L.fec_batch_mul(ctx._h, curve, ptr(k), ptr(p), ptr(out), n)
            best = min(best, time.perf_counter() - t0)
            assert rc == 0, rc
        print("%-10s var  chunk %7d : %6.2f ms" % (name, chunk, best * 1e3), flush=True)
