"""Latency of fec_batch_double_mul_dev at modest batch sizes (inputs resident in HBM, host wall clock around a
synchronised call, best of 10).  `FEC_AB_LIB=<path>` times another build of the library for comparison.

    python tools/double_mul_small_perf.py
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from forge_ec_amd import _lib  # noqa: E402

if os.environ.get("FEC_AB_LIB"):
    _lib.SO_PATH = os.path.abspath(os.environ["FEC_AB_LIB"])
import forge_ec_amd as F  # noqa: E402
from forge_ec_amd import synth  # noqa: E402

NAMES = {0: "secp256k1", 1: "p256", 2: "ed25519"}


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()


def main():
    ctx = F.Context(0)
    s = torch.cuda.Stream()
    torch.cuda.set_stream(s)
    for c in (0, 1, 2):
        for logn in [int(v) for v in os.environ.get("FEC_DM_LOG2", "10,12,14,16").split(",")]:
            n = 1 << logn
            u1, u2, q = dev(synth.scalars(n, c, 11)), dev(synth.scalars(n, c, 12)), dev(synth.points(n, c, 13))
            out = torch.empty_like(q)
            best = 1e9
            for _ in range(10):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ctx.batch_double_mul_dev(c, u1.data_ptr(), u2.data_ptr(), q.data_ptr(), out.data_ptr(), n, s.cuda_stream)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            print(json.dumps({"row": "double_mul", "curve": NAMES[c], "n": n, "ms": round(best * 1e3, 3),
                              "lib": os.path.relpath(_lib.SO_PATH, ROOT), "side_stream_max": os.environ.get("FEC_SIDE_STREAM_MAX", "default (98304)"), "checksum": int(out.sum().item()) & 0xFFFFFFFF}), flush=True)


main()
