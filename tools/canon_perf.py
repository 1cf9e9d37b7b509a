"""Canonical-math mode timing probe (NOT the headline bench, NOT reference parity): kernel ms from
HIP events inside the library for secp256k1 key generation (k*G, comb) and ECDH (k*P, windowed),
inputs resident in HBM.  One JSON line per workload.

    python tools/canon_perf.py [log2_n] [reps] [curve,...]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import forge_ec_amd as F  # noqa: E402
from forge_ec_amd import synth  # noqa: E402
from forge_ec_amd.canon import CANON_CURVES  # noqa: E402

# field multiplications per unit (each = 64 MAD32 for the 512-bit product + 8 for the fold)
MULS = {"keygen": 64 * 11 + 274, "ecdh": 7 + 13 * 11 + 64 * (4 * 7 + 16) + 274 + 5}
MAD_PER_MUL = 72


def main():
    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    curves = sys.argv[3].split(",") if len(sys.argv) > 3 else list(CANON_CURVES)
    n = 1 << logn
    ctx = F.Context(0)
    ctx.set_timing(True)
    st = torch.cuda.current_stream().cuda_stream
    k = torch.from_numpy(synth.scalars(n, 0, 41).view(np.int64)).cuda()
    k2 = torch.from_numpy(synth.scalars(n, 0, 42).view(np.int64)).cuda()
    k3 = torch.from_numpy(synth.scalars(n, 0, 43).view(np.int64)).cuda()
    k4 = torch.from_numpy(synth.scalars(n, 0, 44).view(np.int64)).cuda()
    pub = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    out = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    status = torch.empty(n, dtype=torch.uint8, device="cuda")
    for cname in curves:
        c = CANON_CURVES[cname](ctx)
        extra = {"secp256k1": ("ecdsa-verify", "bip340-verify"), "p256": ("ecdsa-verify",), "ed25519": ("eddsa-verify",)}[cname]
        for name in ("keygen", "ecdh", "double-mul") + extra:
            best = None
            for _ in range(reps + 1):
                if name == "keygen":
                    c.mul_base_dev(k.data_ptr(), pub.data_ptr(), status.data_ptr(), n, st)
                    ms, kern = ctx.last_kernel_ms()
                elif name == "ecdh":
                    c.mul_dev(k2.data_ptr(), pub.data_ptr(), out.data_ptr(), status.data_ptr(), n, st)
                    ms, kern = ctx.last_kernel_ms()
                elif name == "ecdsa-verify":  # scalars + comb + accumulate + normalise + compare, wall clock
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    c.ecdsa_verify_dev(k.data_ptr(), k2.data_ptr(), k3.data_ptr(), pub.data_ptr(), status.data_ptr(), n, st)
                    torch.cuda.synchronize()
                    ms, kern = (time.perf_counter() - t0) * 1e3, "k_canon_ecdsa_scalars + comb + accumulate + normalize + finish"
                elif name in ("bip340-verify", "eddsa-verify"):  # prepare (decode / lift_x) + double-mul + final test
                    fn = c.bip340_verify_dev if name == "bip340-verify" else c.eddsa_verify_dev
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    fn(k.data_ptr(), k2.data_ptr(), k3.data_ptr(), k4.data_ptr(), status.data_ptr(), n, st)
                    torch.cuda.synchronize()
                    ms, kern = (time.perf_counter() - t0) * 1e3, "prepare + comb + accumulate + normalize + finish"
                else:  # u1*G + u2*P: three launches on the ctx stream; wall clock around a device sync
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    c.double_mul_dev(k.data_ptr(), k2.data_ptr(), pub.data_ptr(), out.data_ptr(), status.data_ptr(), n, st)
                    torch.cuda.synchronize()
                    ms, kern = (time.perf_counter() - t0) * 1e3, "k_canon_mul_base + k_canon_mul<accum> + k_canon_normalize"
                best = ms if best is None or ms < best else best
            torch.cuda.synchronize()
            assert name.endswith("-verify") or int(status.sum()) == 0
            rate = n / (best * 1e-3)
            print(json.dumps({"workload": "%s-canon-%s" % (cname, name), "mode": "canonical math, NOT reference parity",
                              "n": n, "kernel": kern, "ms": round(best, 4), "M_per_s": round(rate / 1e6, 3)}),
                  flush=True)


main()
