"""MSM timing split: the fold kernel alone (HIP events) vs the whole host-pointer call."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import forge_ec_amd as F
import vectors as V
ctx = F.Context(0)
ctx.set_timing(True)
for n in (1024, 4096):
    k, p = V.scalars(n, 0, 5), V.points(n, 0, 6)
    ctx.multi_scalar_mul(0, k, p)
    for rep in range(3):
        t0 = time.perf_counter()
        ctx.multi_scalar_mul(0, k, p)
        wall = (time.perf_counter() - t0) * 1e3
        ms, name = ctx.last_kernel_ms()
        print("n=%d wall %.2f ms; last kernel %s %.2f ms = %.2f us per addition" % (n, wall, name, ms, ms * 1e3 / n), flush=True)
