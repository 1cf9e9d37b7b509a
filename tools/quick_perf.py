"""Quick single-GPU timing probe: kernel ms (HIP events inside the library) per curve/config.
FEC_AB_LIB=<path> times another build of the library (same-box A/B comparisons)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from forge_ec_amd import _lib
if os.environ.get("FEC_AB_LIB"):
    _lib.SO_PATH = os.path.abspath(os.environ["FEC_AB_LIB"])
import forge_ec_amd as F
import vectors as V

def main():
    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    curves = [int(c) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2]
    n = 1 << logn
    ctx = F.Context(0)
    print(ctx.device_info(), flush=True)
    peak = ctx.measure_peak_mad32()
    print("peak MAD32/s: %.3e" % peak, flush=True)
    ctx.set_timing(True)
    need = {0: 693248, 1: 278528, 2: 248832}
    mode = sys.argv[3] if len(sys.argv) > 3 else "var"
    need_fixed = {0: 693248, 1: 278528, 2: 82944}
    for curve in curves:
        k = V.scalars(n, curve, 1); p = V.points(n, curve, 2)
        if mode == "fixed":
            dk = torch.from_numpy(k.view(np.int64)).cuda()
            do = torch.empty((n, V.POINT_LIMBS[curve]), dtype=torch.int64, device="cuda")
            st = torch.cuda.current_stream().cuda_stream
            for rep in range(3):
                ctx.batch_mul_fixed_dev(curve, dk.data_ptr(), ctx.generator_dev(curve), do.data_ptr(), n, st)
                ms, name = ctx.last_kernel_ms()
                rate = n / (ms * 1e-3)
                print("curve %d n=2^%d FIXED %s: %.3f ms  %.3f M scalar-mul/s  alg-MAD32 %.2f T/s (%.1f%% of peak)" % (
                    curve, logn, name, ms, rate / 1e6, rate * need_fixed[curve] / 1e12, 100 * rate * need_fixed[curve] / peak), flush=True)
            continue
        dk = torch.from_numpy(k.view(np.int64)).cuda(); dp = torch.from_numpy(p.view(np.int64)).cuda()
        do = torch.empty_like(dp)
        st = torch.cuda.current_stream().cuda_stream
        for rep in range(3):
            ctx.batch_mul_dev(curve, dk.data_ptr(), dp.data_ptr(), do.data_ptr(), n, st)
            ms, name = ctx.last_kernel_ms()
            rate = n / (ms * 1e-3)
            print("curve %d n=2^%d %s: %.3f ms  %.3f M scalar-mul/s  alg-MAD32 %.2f T/s (%.1f%% of peak)" % (
                curve, logn, name, ms, rate / 1e6, rate * need[curve] / 1e12, 100 * rate * need[curve] / peak), flush=True)
main()
