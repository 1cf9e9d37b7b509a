"""Fixed-base multiplication by a base of the caller's own (random projective point, not the generator) with the tables
off and on: from 2^16 elements on such a launch builds a prefix table for itself (fecgpu.hip: per_call_prefix); the kernel
time of each of five launches includes that build.

    python tools/own_base_prefix_ab.py
"""
import sys, os, numpy as np, torch, time
sys.path.insert(0, os.getcwd())
import forge_ec_amd as F
from forge_ec_amd import synth
n = 1 << 20
for curve in (0, 1, 2):
    for bits in (0, 24):
        ctx = F.Context(0)
        ctx.set_fixed_prefix_bits(bits)
        k = torch.from_numpy(synth.scalars(n, curve, 5).view(np.int64)).cuda()
        base = torch.from_numpy(synth.points(1, curve, 6).view(np.int64)).cuda()
        out = torch.empty((n, F.POINT_LIMBS[curve]), dtype=torch.int64, device="cuda")
        s = torch.cuda.Stream()
        ctx.set_timing(True)
        ts = []
        for i in range(5):
            ctx.batch_mul_fixed_dev(curve, k.data_ptr(), base.data_ptr(), out.data_ptr(), n, s.cuda_stream)
            ms, name = ctx.last_kernel_ms()
            torch.cuda.synchronize()
            ts.append(round(ms, 3))
        print({"curve": curve, "tables": bits != 0, "base": "caller's own (projective)", "ms per launch incl. the per-launch table": ts}, flush=True)
        ctx.close()
