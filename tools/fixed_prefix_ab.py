import sys, os, numpy as np, torch, time
sys.path.insert(0, os.getcwd())
import forge_ec_amd as F
from forge_ec_amd import synth
n = 1 << 20
for curve in (1, 0, 2):
    for bits in (0, 24):
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
        print(curve, bits, ctx.fixed_prefix_bits(curve), name, ts, flush=True)
        ctx.close()
