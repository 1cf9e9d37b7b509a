"""The four kernel times cu_split.hpp divides the CUs by (ms per 2^20 P-256 multiplications on the whole chip): variable
base with projective / affine (z = 1) base points, fixed base without / with the generator's prefix table.

    python tools/p256_split_constants.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import forge_ec_amd as F  # noqa: E402
import vectors as V  # noqa: E402


def main():
    n = 1 << 20
    ctx = F.Context(0)
    ctx.set_timing(True)
    k = V.scalars(n, 1, 1)
    p = V.points(n, 1, 2)
    pa = p.copy()
    pa[:, 8:] = 0
    pa[:, 8] = 1                       # from_affine(key): z = 1
    dk = torch.from_numpy(k.view(np.int64)).cuda()
    do = torch.empty((n, 12), dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def best(f):
        t = []
        for _ in range(4):
            f()
            t.append(ctx.last_kernel_ms()[0])
        return min(t)

    for name, pts in (("kP256VarMs (projective bases)", p), ("kP256VarAffineMs (z = 1 bases)", pa)):
        dp = torch.from_numpy(pts.view(np.int64)).cuda()
        print("%-36s %.2f ms" % (name, best(lambda: ctx.batch_mul_dev(1, dk.data_ptr(), dp.data_ptr(), do.data_ptr(), n, st))))
    for name, bits in (("kP256FixedMs (no table)", 0), ("kP256FixedPrefixMs (24-bit table)", 24)):
        ctx.set_fixed_prefix_bits(bits)
        ctx.build_fixed_prefix(1)
        print("%-36s %.2f ms" % (name, best(lambda: ctx.batch_mul_fixed_dev(1, dk.data_ptr(), ctx.generator_dev(1), do.data_ptr(), n, st))))
    pr = torch.cuda.get_device_properties(0)
    print({a: getattr(pr, a) for a in dir(pr) if "pci" in a.lower()})


main()
