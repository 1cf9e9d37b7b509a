"""Whole-batch differential soak on the GPU box: every element of large random batches of the verification /
key-exchange / validation pipelines against the C oracle (16 host threads).  Prints one JSON line per pipeline.

  python tests/soak.py [log2-batch, default 19] [seed offset, default 0: another offset is another set of batches]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/ is one of the places allowed to call the oracle
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import forge_ec_amd as F  # noqa: E402
import vectors as V  # noqa: E402
from oracle import c_oracle as O  # noqa: E402


def main():
    n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 19)
    so = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ctx = F.Context(0)
    rng = np.random.default_rng(20261004 + so)

    def report(name, got, want, t_gpu, t_cpu):
        bad = int(np.count_nonzero(np.any(np.asarray(got).reshape(n, -1) != np.asarray(want).reshape(n, -1), axis=1)))
        hist = {int(k): int(v) for k, v in zip(*np.unique(np.asarray(want).reshape(n, -1)[:, -1] if np.asarray(want).ndim > 1 else want, return_counts=True))} if np.asarray(want).ndim == 1 else None
        print(json.dumps({"pipeline": name, "n": n, "seed_offset": so, "mismatches": bad, "status_histogram": hist,
                          "gpu_s": round(t_gpu, 3), "oracle_s": round(t_cpu, 1)}), flush=True)
        return bad

    total = 0
    for curve in (0, 1, 2):   # whole-batch fixed-base and variable-base multiplications
        g = O.generator(curve)
        k = V.scalars(n, curve, 8001 + curve + so)
        k[::1001] = 0
        t0 = time.perf_counter(); got = ctx.batch_mul_fixed(curve, k, g); t1 = time.perf_counter()
        want = O.batch_mul_fixed(curve, k, g, nthreads=16); t2 = time.perf_counter()
        total += report("multiply fixed-base curve %d" % curve, got, want, t1 - t0, t2 - t1)
        p = V.points(n, curve, 8011 + curve + so)
        p[7::997] = O.identity(curve)
        t0 = time.perf_counter(); got = ctx.batch_mul(curve, k, p); t1 = time.perf_counter()
        want = O.batch_mul(curve, k, p, nthreads=16); t2 = time.perf_counter()
        total += report("multiply variable-base curve %d" % curve, got, want, t1 - t0, t2 - t1)
    for curve in (0, 1, 2):   # multiply(G, u1) + multiply(Q, u2) (BASELINE configs[4] is the secp256k1 one)
        u1, u2 = V.scalars(n, curve, 8021 + curve + so), V.scalars(n, curve, 8031 + curve + so)
        u1[::1003] = 0
        q = V.points(n, curve, 8041 + curve + so)
        q[5::991] = O.identity(curve)
        t0 = time.perf_counter(); got = ctx.batch_double_mul(curve, u1, u2, q); t1 = time.perf_counter()
        want = O.batch_double_mul(curve, u1, u2, q, nthreads=16); t2 = time.perf_counter()
        total += report("double multiplication curve %d" % curve, got, want, t1 - t0, t2 - t1)
    for curve, fn, ofn in ((0, ctx.ecdsa_verify_secp256k1, O.batch_secp256k1_ecdsa_verify), (1, ctx.ecdsa_verify_p256, O.batch_p256_ecdsa_verify)):
        dg = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        dg[::3, 0] &= 0x7F
        r, s = V.scalars(n, curve, 9001 + so), V.scalars(n, curve, 9002 + so)
        pk = np.ascontiguousarray(np.concatenate([V.field_elements(n, curve, 9003 + so), V.field_elements(n, curve, 9004 + so)], axis=1))
        inf = (rng.integers(0, 64, size=n) == 0).astype(np.uint8)
        t0 = time.perf_counter(); got = fn(dg, r, s, pk, inf); t1 = time.perf_counter()
        want = ofn(dg, r, s, pk, inf, nthreads=16); t2 = time.perf_counter()
        total += report("ecdsa_verify " + ("secp256k1" if curve == 0 else "p256"), got, want, t1 - t0, t2 - t1)
    s, k = V.scalars(n, 2, 9011 + so), V.scalars(n, 2, 9012 + so)
    pk = np.ascontiguousarray(np.concatenate([V.field_elements(n, 2, 9013 + so), V.field_elements(n, 2, 9014 + so)], axis=1))
    rr = np.ascontiguousarray(np.concatenate([V.field_elements(n, 2, 9015 + so), V.field_elements(n, 2, 9016 + so)], axis=1))
    pinf = (rng.integers(0, 16, size=n) == 0).astype(np.uint8)
    rinf = (rng.integers(0, 64, size=n) == 0).astype(np.uint8)
    t0 = time.perf_counter(); got = ctx.eddsa_verify_ed25519(rr, rinf, pk, pinf, s, k); t1 = time.perf_counter()
    want = O.batch_ed25519_eddsa_verify(rr, rinf, pk, pinf, s, k, nthreads=16); t2 = time.perf_counter()
    total += report("eddsa_verify ed25519", got, want, t1 - t0, t2 - t1)
    for curve in (0, 1, 2):   # Schnorr verify per signature (P-256: a quarter of the keys true curve points, e = 1 for some)
        m4 = n >> 2
        pk = np.ascontiguousarray(np.concatenate([V.field_elements(m4, curve, 9041 + so), V.field_elements(m4, curve, 9042 + so)], axis=1))
        rr = np.ascontiguousarray(np.concatenate([V.field_elements(m4, curve, 9043 + so), V.field_elements(m4, curve, 9044 + so)], axis=1))
        ss, ee = V.scalars(m4, curve, 9045 + so), V.scalars(m4, curve, 9046 + so)
        ee[::5] = [1, 0, 0, 0]
        pinf = (rng.integers(0, 16, size=m4) == 0).astype(np.uint8)
        rinf = (rng.integers(0, 64, size=m4) == 0).astype(np.uint8)
        t0 = time.perf_counter(); got = ctx.schnorr_verify(curve, pk, rr, ss, ee, pk_inf=pinf, r_inf=rinf); t1 = time.perf_counter()
        want = O.batch_schnorr_verify(curve, pk, pinf, rr, rinf, ss, ee, nthreads=16); t2 = time.perf_counter()
        bad = int(np.count_nonzero(got != want))
        print(json.dumps({"pipeline": "schnorr_verify curve %d" % curve, "n": m4, "mismatches": bad, "gpu_s": round(t1 - t0, 3),
                          "oracle_s": round(t2 - t1, 1)}), flush=True)
        total += bad
    for curve in (0, 1):
        sk = V.scalars(n, curve, 9021 + so)
        pk = np.ascontiguousarray(np.concatenate([V.field_elements(n, curve, 9022 + so), V.field_elements(n, curve, 9023 + so)], axis=1))
        if curve == 1:  # true curve points for a quarter of the batch: the reference accepts about half of them
            import random
            pr, pb = V.PRIME[1], 0x5AC635D8AA3A93E7B3EBBD55769886BC651D06B0CC53B0F63BCE3C3E27D2604B
            rnd, rows = random.Random(77), []
            while len(rows) < n // 4:
                x = rnd.randrange(pr)
                rhs = (x * x * x - 3 * x + pb) % pr
                y = pow(rhs, (pr + 1) // 4, pr)
                if y * y % pr == rhs:
                    rows.append(V.limbs_of(x) + V.limbs_of(y))
            pk[: n // 4] = np.array(rows, dtype=np.uint64)
        inf = (rng.integers(0, 64, size=n) == 0).astype(np.uint8)
        t0 = time.perf_counter(); sec, st = ctx.batch_ecdh(curve, sk, pk, inf); t1 = time.perf_counter()
        wsec, wst = O.batch_ecdh(curve, sk, pk, inf, nthreads=16); t2 = time.perf_counter()
        total += report("ecdh " + ("secp256k1" if curve == 0 else "p256") + " (Ok: %d)" % int((wst == 0).sum()),
                        np.concatenate([sec, st[:, None]], axis=1), np.concatenate([wsec, wst[:, None]], axis=1), t1 - t0, t2 - t1)
    m = n >> 2
    xy = np.ascontiguousarray(np.concatenate([V.field_elements(m, 2, 9031 + so), V.field_elements(m, 2, 9032 + so)], axis=1))
    xy[::7, :4] = 0
    inf = (rng.integers(0, 16, size=m) == 0).astype(np.uint8)
    t0 = time.perf_counter(); got = ctx.batch_validate_point(2, xy, inf); t1 = time.perf_counter()
    want = O.batch_validate_point(2, xy, inf, nthreads=16); t2 = time.perf_counter()
    bad = int(np.count_nonzero(got != want))
    print(json.dumps({"pipeline": "validate_point ed25519", "n": m, "mismatches": bad, "valid": int(want.sum()),
                      "gpu_s": round(t1 - t0, 3), "oracle_s": round(t2 - t1, 1)}), flush=True)
    total += bad
    # round 4: device-resident shards of a multi-device ctx gathered onto one device (this box: its GPU listed twice), whole
    # batch against the oracle results computed above for the same scalars / points of curve 1
    import torch
    with F.Context(devices=[0, 0]) as mctx:
        for curve in (0, 1, 2):
            L = F.POINT_LIMBS[curve]
            k = V.scalars(n, curve, 8001 + curve + so)
            p = V.points(n, curve, 8011 + curve + so)
            cut = n // 2 + 12345 % max(n // 4, 1)
            shards = ((0, cut), (cut, n))
            dk = [torch.from_numpy(k[a:b].view(np.int64)).cuda() for a, b in shards]
            dp = [torch.from_numpy(p[a:b].view(np.int64)).cuda() for a, b in shards]
            do = [torch.zeros((b - a, L), dtype=torch.int64, device="cuda") for a, b in shards]
            full = torch.zeros((n, L), dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            mctx.multi_batch_mul_dev(curve, [t.data_ptr() for t in dk], [t.data_ptr() for t in dp], [t.data_ptr() for t in do],
                                     [b - a for a, b in shards], full.data_ptr(), 0)
            t1 = time.perf_counter()
            want = O.batch_mul(curve, k, p, nthreads=16); t2 = time.perf_counter()
            total += report("multi_batch_mul_dev [0, 0] gathered, curve %d" % curve, full.cpu().numpy().view(np.uint64), want, t1 - t0, t2 - t1)
    # round 4: schnorr::batch_verify::<Ed25519, D> (release-profile scalar Mul), batches of 512 signatures
    bad = 0
    t_gpu = t_cpu = 0.0
    nb = max(1, min(16, n >> 12))
    for b in range(nb):
        m5 = 512
        pk = np.ascontiguousarray(np.concatenate([V.field_elements(m5, 2, 9051 + so + 7 * b), V.field_elements(m5, 2, 9052 + so + 7 * b)], axis=1))
        rr = np.ascontiguousarray(np.concatenate([V.field_elements(m5, 2, 9053 + so + 7 * b), V.field_elements(m5, 2, 9054 + so + 7 * b)], axis=1))
        ss, aa, ee = V.scalars(m5, 2, 9055 + so + 7 * b), V.scalars(m5, 2, 9056 + so + 7 * b), V.scalars(m5, 2, 9057 + so + 7 * b)
        t0 = time.perf_counter(); got = ctx.schnorr_batch_verify_ed25519(pk, rr, ss, aa, ee); t1 = time.perf_counter()
        want = O.ed25519_schnorr_batch_verify(pk, None, rr, None, ss, aa, ee); t2 = time.perf_counter()
        t_gpu += t1 - t0
        t_cpu += t2 - t1
        if not (got[0] == want[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2]) and got[3] == bool(want[3])):
            bad += 1
    print(json.dumps({"pipeline": "schnorr batch_verify ed25519 (release profile), %d batches of 512" % nb, "mismatches": bad,
                      "gpu_s": round(t_gpu, 3), "oracle_s": round(t_cpu, 1)}), flush=True)
    total += bad
    print(json.dumps({"total_mismatches": total}))
    sys.exit(1 if total else 0)


main()
