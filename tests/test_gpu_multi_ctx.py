"""
Multi-device ctx behind the C ABI (fec_ctx_create_multi, include/fecgpu.h): contiguous shards, one
host thread + copy/compute pipeline per shard worker, results written straight into the caller's
buffer.  A one-GPU box can only list its GPU -- once ([0]) and twice ([0, 0]: two shard workers
on the same device) -- but that exercises the whole host path: sharding of ragged n, concurrent
workers, per-shard pipelines, error propagation.  Results must be oracle-identical.
"""
import numpy as np
import pytest

import vectors as V

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_multi_ctx_batch_mul_matches_oracle(devices, oracle):
    import forge_ec_amd as F
    with F.Context(devices=devices) as ctx:
        assert ctx.device_count() == len(devices)
        for curve in (0, 1, 2):
            for n in (0, 1, 2, 5, 1000, 4099):   # ragged: shards of unequal length, some empty
                k, p = V.scalars(n, curve, 3100 + n), V.points(n, curve, 3200 + n)
                got = ctx.batch_mul(curve, k, p)
                assert got.shape == p.shape
                if n:
                    assert np.array_equal(got, oracle.batch_mul(curve, k, p, nthreads=16)), (devices, curve, n)


def test_multi_ctx_other_elementwise_calls(oracle):
    import forge_ec_amd as F
    n = 3001
    with F.Context(devices=[0, 0]) as ctx, F.Context(0) as one:
        ctx.set_chunk(700)  # several pipeline chunks per shard
        for curve in (0, 1, 2):
            k, k2 = V.scalars(n, curve, 41), V.scalars(n, curve, 42)
            q = V.points(n, curve, 43)
            g = one.generator(curve)
            assert np.array_equal(ctx.batch_mul_fixed(curve, k, g), oracle.batch_mul_fixed(curve, k, g, nthreads=16))
            assert np.array_equal(ctx.batch_double_mul(curve, k, k2, q),
                                  oracle.batch_double_mul(curve, k, k2, q, nthreads=16))
            xy, inf = ctx.batch_to_affine(curve, q)
            xy1, inf1 = one.batch_to_affine(curve, q)
            assert np.array_equal(xy, xy1) and np.array_equal(inf, inf1)
            assert np.array_equal(ctx.batch_compress(curve, xy, inf), one.batch_compress(curve, xy, inf))
            a, b = V.field_elements(n, curve, 44), V.field_elements(n, curve, 45)
            assert np.array_equal(ctx.field_op(curve, 2, a, b), one.field_op(curve, 2, a, b))
            assert np.array_equal(ctx.point_op(curve, 0, q, V.points(n, curve, 46)),
                                  one.point_op(curve, 0, q, V.points(n, curve, 46)))
        # the verification pipelines shard per signature
        rng = np.random.default_rng(5)
        dg = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        for curve, name in ((0, "ecdsa_verify_secp256k1"), (1, "ecdsa_verify_p256")):
            r, s = V.scalars(n, curve, 51), V.scalars(n, curve, 52)
            pk = np.ascontiguousarray(np.concatenate([V.field_elements(n, curve, 53), V.field_elements(n, curve, 54)], axis=1))
            inf = (rng.integers(0, 8, size=n) == 0).astype(np.uint8)
            assert np.array_equal(getattr(ctx, name)(dg, r, s, pk, inf), getattr(one, name)(dg, r, s, pk, inf))
        s, k = V.scalars(n, 2, 55), V.scalars(n, 2, 56)
        pk = np.ascontiguousarray(np.concatenate([V.field_elements(n, 2, 57), V.field_elements(n, 2, 58)], axis=1))
        r = np.ascontiguousarray(np.concatenate([V.field_elements(n, 2, 59), V.field_elements(n, 2, 60)], axis=1))
        inf = (rng.integers(0, 8, size=n) == 0).astype(np.uint8)
        assert np.array_equal(ctx.eddsa_verify_ed25519(r, None, pk, inf, s, k), one.eddsa_verify_ed25519(r, None, pk, inf, s, k))
        for curve in (0, 1, 2):   # Schnorr verify per signature shards like the other verifications
            pkc = np.ascontiguousarray(np.concatenate([V.field_elements(n, curve, 81), V.field_elements(n, curve, 82)], axis=1))
            rc = np.ascontiguousarray(np.concatenate([V.field_elements(n, curve, 83), V.field_elements(n, curve, 84)], axis=1))
            ss, ee = V.scalars(n, curve, 85), V.scalars(n, curve, 86)
            assert np.array_equal(ctx.schnorr_verify(curve, pkc, rc, ss, ee, pk_inf=inf), one.schnorr_verify(curve, pkc, rc, ss, ee, pk_inf=inf))
        # ECDH, validate_point and the three codecs shard as well (fecgpu.h lists every sharded entry point)
        for curve in (0, 1):
            sk = V.scalars(n, curve, 71 + curve)
            pkc = np.ascontiguousarray(np.concatenate([V.field_elements(n, curve, 73), V.field_elements(n, curve, 74)], axis=1))
            sec, st = ctx.batch_ecdh(curve, sk, pkc, inf)
            sec1, st1 = one.batch_ecdh(curve, sk, pkc, inf)
            assert np.array_equal(sec, sec1) and np.array_equal(st, st1)
        for curve in (0, 1, 2):
            xyc = np.ascontiguousarray(np.concatenate([V.field_elements(n, curve, 75), V.field_elements(n, curve, 76)], axis=1))
            assert np.array_equal(ctx.batch_validate_point(curve, xyc, inf), one.batch_validate_point(curve, xyc, inf))
            comp = one.batch_compress(curve, xyc, inf)
            for a_, b_ in zip(ctx.batch_decompress(curve, comp), one.batch_decompress(curve, comp)):
                assert np.array_equal(a_, b_)
            unc = one.batch_encode_uncompressed(curve, xyc, inf)
            assert np.array_equal(ctx.batch_encode_uncompressed(curve, xyc, inf), unc)
            for a_, b_ in zip(ctx.batch_decode_uncompressed(curve, unc), one.batch_decode_uncompressed(curve, unc)):
                assert np.array_equal(a_, b_)
        # not element-wise: runs on devices[0]
        a = V.scalars(64, 1, 61)
        assert ctx.ecdsa_batch_verify(1, dg[:64], V.scalars(64, 1, 62), V.scalars(64, 1, 63), pk[:64], None, a)[0] == \
            one.ecdsa_batch_verify(1, dg[:64], V.scalars(64, 1, 62), V.scalars(64, 1, 63), pk[:64], None, a)[0]
        assert np.array_equal(ctx.generator(0), one.generator(0))
        k, p = V.scalars(64, 0, 47), V.points(64, 0, 48)
        assert np.array_equal(ctx.multi_scalar_mul(0, k, p), one.multi_scalar_mul(0, k, p))


def test_multi_ctx_rejects_device_pointers_and_bad_lists():
    import ctypes
    import forge_ec_amd as F
    from forge_ec_amd import _lib
    L = _lib.lib()
    with F.Context(devices=[0, 0]) as ctx:
        rc = L.fec_batch_mul_dev(ctx._h, 0, None, None, None, 0, None)
        assert rc == -5, rc  # FEC_E_UNSUPPORTED
    h = ctypes.c_void_p()
    assert L.fec_ctx_create_multi(ctypes.byref(h), None, 0) == -1          # FEC_E_ARG
    bad = (ctypes.c_int * 2)(0, 99)
    assert L.fec_ctx_create_multi(ctypes.byref(h), bad, 2) != 0 and not h.value
    assert L.fec_strerror(-6).decode().startswith("multi-device")
