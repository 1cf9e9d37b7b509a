"""Kernel timings of the parity-mode rows next to the hot path: to_affine, compressed
encoding, ECDSA verify (secp256k1), multi_scalar_multiply, Schnorr batch_verify.  Inputs resident in HBM
for the *_dev entry points; HIP events inside the library.  One JSON line per measurement.

    python tools/next_rows_perf.py [log2_n]
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

NAMES = {0: "secp256k1", 1: "p256", 2: "ed25519"}


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64 if a.dtype == np.uint64 else a.dtype)).cuda()


def emit(**kw):
    print(json.dumps(kw), flush=True)


def main():
    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n = 1 << logn
    ctx = F.Context(0)
    ctx.set_timing(True)
    s = torch.cuda.Stream()
    torch.cuda.set_stream(s)
    st = s.cuda_stream
    for c in (0, 1, 2):
        pts = dev(synth.points(n, c, 51))
        xy = torch.empty((n, 8), dtype=torch.int64, device="cuda")
        inf = torch.empty(n, dtype=torch.uint8, device="cuda")
        out = torch.empty(n * 33 + 3, dtype=torch.uint8, device="cuda")
        best = [1e9, 1e9]
        for _ in range(4):
            ctx.batch_to_affine_dev(c, pts.data_ptr(), xy.data_ptr(), inf.data_ptr(), n, st)
            best[0] = min(best[0], ctx.last_kernel_ms()[0])
            ctx.batch_compress_dev(c, xy.data_ptr(), inf.data_ptr(), out.data_ptr(), n, st)
            best[1] = min(best[1], ctx.last_kernel_ms()[0])
        emit(row="to_affine", curve=NAMES[c], n=n, ms=round(best[0], 4), M_per_s=round(n / best[0] / 1e3, 2))
        emit(row="compress", curve=NAMES[c], n=n, ms=round(best[1], 4), M_per_s=round(n / best[1] / 1e3, 2),
             GBps=round(n * 98 / best[1] / 1e6, 1))
    # ECDSA verify in parity mode: scalar pre-pass + the two multiplication kernels + finishing pass
    for c, fn in ((0, ctx.ecdsa_verify_secp256k1_dev), (1, ctx.ecdsa_verify_p256_dev)):
        dg = dev(np.frombuffer(synth.scalars(n, c, 61).tobytes(), dtype=np.uint8).copy())
        r, sg, pk = dev(synth.scalars(n, c, 62)), dev(synth.scalars(n, c, 63)), dev(synth.field_elements(2 * n, c, 64))
        status = torch.empty(n, dtype=torch.uint8, device="cuda")
        best = 1e9
        for _ in range(3):
            fn(dg.data_ptr(), r.data_ptr(), sg.data_ptr(), pk.data_ptr(), None, status.data_ptr(), n, st)
            best = min(best, ctx.last_kernel_ms()[0])
        emit(row="ecdsa_verify (parity)", curve=NAMES[c], n=n, ms=round(best, 4), M_per_s=round(n / best / 1e3, 2),
             kernels=ctx.last_kernel_ms()[1])
    # ECDH: validation + from_affine, the variable-base multiplication, to_affine + x.to_bytes()
    for c in (0, 1):
        kk, pp = dev(synth.scalars(n, c, 91)), dev(synth.field_elements(2 * n, c, 92))
        sec = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
        status = torch.empty(n, dtype=torch.uint8, device="cuda")
        best = 1e9
        for _ in range(3):
            ctx.batch_ecdh_dev(c, kk.data_ptr(), pp.data_ptr(), None, sec.data_ptr(), status.data_ptr(), n, st)
            best = min(best, ctx.last_kernel_ms()[0])
        emit(row="derive_shared_secret (ECDH, parity)", curve=NAMES[c], n=n, ms=round(best, 4), M_per_s=round(n / best / 1e3, 2),
             kernels=ctx.last_kernel_ms()[1])
    # EdDSA verify from the point computation on: from_affine + fixed-base table kernel + scheduler + finishing pass
    rr, pp = dev(synth.field_elements(2 * n, 2, 65)), dev(synth.field_elements(2 * n, 2, 66))
    sg, kk = dev(synth.scalars(n, 2, 67)), dev(synth.scalars(n, 2, 68))
    status = torch.empty(n, dtype=torch.uint8, device="cuda")
    best = 1e9
    for _ in range(3):
        ctx.eddsa_verify_ed25519_dev(rr.data_ptr(), None, pp.data_ptr(), None, sg.data_ptr(), kk.data_ptr(), status.data_ptr(), n, st)
        best = min(best, ctx.last_kernel_ms()[0])
    emit(row="eddsa_verify (parity)", curve="ed25519", n=n, ms=round(best, 4), M_per_s=round(n / best / 1e3, 2),
         kernels=ctx.last_kernel_ms()[1])
    # host-pointer rows with a sequential fold: modest sizes
    m = 1 << 12
    ctx.set_timing(False)
    k, p = synth.scalars(m, 0, 71), synth.points(m, 0, 72)
    t0 = time.perf_counter()
    ctx.multi_scalar_mul(0, k, p)
    emit(row="multi_scalar_multiply", curve="secp256k1", n=m, ms=round((time.perf_counter() - t0) * 1e3, 3),
         note="products in parallel + ordered fold, each addition on four lanes (secp::padd_coop); host pointers, PCIe included")
    for c in (1, 2):
        kc, pc = synth.scalars(m, c, 71), synth.points(m, c, 72)
        ctx.multi_scalar_mul(c, kc, pc)
        t0 = time.perf_counter()
        ctx.multi_scalar_mul(c, kc, pc)
        emit(row="multi_scalar_multiply", curve=NAMES[c], n=m, ms=round((time.perf_counter() - t0) * 1e3, 3),
             note="products in parallel + ordered fold, each addition on %s; host pointers, PCIe included"
                  % ("five lanes (p256::padd_coop)" if c == 1 else "four lanes (ed::padd_coop)"))
    pkxy, rxy = synth.field_elements(2 * m, 0, 73).reshape(m, 8), synth.field_elements(2 * m, 0, 74).reshape(m, 8)
    s_, a_, e_ = synth.scalars(m, 0, 75), synth.scalars(m, 0, 76), synth.scalars(m, 0, 77)
    t0 = time.perf_counter()
    ctx.schnorr_batch_verify_secp256k1(pkxy, rxy, s_, a_, e_)
    emit(row="schnorr batch_verify", curve="secp256k1", n=m, ms=round((time.perf_counter() - t0) * 1e3, 3),
         note="3 ladders per signature in parallel + two ordered folds (four lanes per addition); host pointers, PCIe included")
    for c in (0, 1):
        dgb = np.frombuffer(synth.scalars(m, c, 81).tobytes(), dtype=np.uint8).reshape(m, 32).copy()
        dgb[:, 0] &= 0x7F
        rb, sb, ab = synth.scalars(m, c, 82), synth.scalars(m, c, 83), synth.scalars(m, c, 84)
        pkb = synth.field_elements(2 * m, c, 85).reshape(m, 8)
        ctx.ecdsa_batch_verify(c, dgb, rb, sb, pkb, None, ab)  # first call sizes the work areas
        t0 = time.perf_counter()
        ctx.ecdsa_batch_verify(c, dgb, rb, sb, pkb, None, ab)
        emit(row="ecdsa batch_verify", curve=NAMES[c], n=m, ms=round((time.perf_counter() - t0) * 1e3, 3),
             note="scalars + 2 multiplications per signature in parallel, then the ordered point fold%s and the ordered "
                  "scalar sum; host pointers, PCIe included" % (" (four lanes per addition)" if c == 0 else " (five lanes per addition)"))


main()
