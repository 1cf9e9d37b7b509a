"""Host-pointer entry point (what the Rust shim calls) end-to-end: PCIe-inclusive rate."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import forge_ec_amd as F
from forge_ec_amd import synth as V
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << logn
ctx = F.Context(0)
for curve, name in ((0, "secp256k1"), (2, "ed25519")):
    k = V.scalars(n, curve, 1); p = V.points(n, curve, 2)
    ctx.batch_mul(curve, k[:1024], p[:1024])
    for rep in range(3):
        t0 = time.perf_counter(); out = ctx.batch_mul(curve, k, p); dt = time.perf_counter() - t0
        print("%s host-pointer batch_mul n=2^%d: %.1f ms  %.2f M scalar-mul/s (PCIe-inclusive)" % (name, logn, dt * 1e3, n / dt / 1e6), flush=True)
g = ctx.generator(2); k = V.scalars(n, 2, 3)
for rep in range(3):
    t0 = time.perf_counter(); out = ctx.batch_mul_fixed(2, k, g); dt = time.perf_counter() - t0
    print("ed25519 host-pointer batch_mul_fixed n=2^%d: %.1f ms  %.2f M scalar-mul/s (PCIe-inclusive)" % (logn, dt * 1e3, n / dt / 1e6), flush=True)
