"""Fixed-base multiplication by the generator with and without the prefix table, per curve: kernel time of six launches
(the first one builds the table) and a comparison of the outputs with the table-less ones.

    python tools/fixed_prefix_ab.py [bits ...]      (default: 0 24; 28 needs ~50 GB of device memory for secp256k1)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import forge_ec_amd as F  # noqa: E402
from forge_ec_amd import synth  # noqa: E402

n = 1 << 20
widths = [int(a) for a in sys.argv[1:]] or [0, 24]
for curve in (0, 1, 2):
    ref = None
    for bits in widths:
        ctx = F.Context(0)
        ctx.set_fixed_prefix_bits(bits)
        k = torch.from_numpy(synth.scalars(n, curve, 5).view(np.int64)).cuda()
        out = torch.empty((n, F.POINT_LIMBS[curve]), dtype=torch.int64, device="cuda")
        s = torch.cuda.Stream()
        ctx.set_timing(True)
        ts = []
        for i in range(6):
            t0 = time.perf_counter()
            ctx.batch_mul_fixed_dev(curve, k.data_ptr(), ctx.generator_dev(curve), out.data_ptr(), n, s.cuda_stream)
            ms, name = ctx.last_kernel_ms()
            torch.cuda.synchronize()
            ts.append((round(ms, 3), round((time.perf_counter() - t0) * 1e3, 2)))
        same = None
        if ref is None:
            ref = out.clone()
        else:
            same = bool(torch.equal(ref, out))
        print({"curve": curve, "bits_asked": bits, "bits_built": ctx.fixed_prefix_bits(curve), "kernel": name,
               "kernel_ms, wall_ms per launch": ts, "same_as_first_width": same}, flush=True)
        ctx.close()
        del out, k
        torch.cuda.empty_cache()
