"""Host-pointer fec_batch_mul at 2^20 elements with the library's default chunking and with chunks of 2^18 (what every
entry point used before the scheduler kernels got a chunk that fills their slots exactly once).

    python tools/host_default_chunk_ab.py
"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import forge_ec_amd as F
from forge_ec_amd import synth as V
from forge_ec_amd._lib import lib
n = 1 << 20
L = lib()
def ptr(a): return ctypes.c_void_p(a.ctypes.data)
for curve, name in ((0, "secp256k1"), (1, "p256"), (2, "ed25519")):
    k, p = V.scalars(n, curve, 1), V.points(n, curve, 2)
    out = np.zeros_like(p)
    for label, setc in (("default chunk", None), ("chunk 2^18", 1 << 18)):
        ctx = F.Context(0)
        if setc: ctx.set_chunk(setc)
        ctx.batch_mul(curve, k[:1024], p[:1024])
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); rc = L.fec_batch_mul(ctx._h, curve, ptr(k), ptr(p), ptr(out), n); best = min(best, time.perf_counter() - t0); assert rc == 0
        print("%-10s var 2^20 host pointers, %-14s: %6.2f ms" % (name, label, best * 1e3), flush=True)
        ctx.close()
