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


# ---- device-resident shards: fec_multi_batch_*_dev ----
def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()


@pytest.mark.parametrize("devices,counts", [([0, 0], (1000, 1777)), ([0, 0, 0], (513, 0, 2049)), ([0], (300,))])
def test_multi_dev_shards_gathered_bit_exact(devices, counts, oracle):
    """Shards already resident in device memory, results gathered onto the consumer device by peer copies (a one-GPU box
    lists its GPU several times: the whole path runs, the copies are device-to-device on one device).  Every curve;
    variable base, fixed base (the generator and a base of the caller's own) and u1*G + u2*Q; the gathered batch and
    the per-shard outputs against a single-device ctx's *_dev call and the oracle."""
    import torch
    import forge_ec_amd as F
    n = sum(counts)
    edges = np.concatenate([[0], np.cumsum(counts)])
    with F.Context(devices=devices) as ctx, F.Context(0) as one:
        ctx.set_chunk(600)   # several chunks per shard: a chunk's copy runs under the next chunk's kernels
        for curve in (0, 1, 2):
            L = F.POINT_LIMBS[curve]
            k, k2, p = V.scalars(n, curve, 9100), V.scalars(n, curve, 9101), V.points(n, curve, 9102)
            shards = [(int(edges[g]), int(edges[g + 1])) for g in range(len(devices))]
            dk = [_dev(k[a:b]) for a, b in shards]
            dk2 = [_dev(k2[a:b]) for a, b in shards]
            dp = [_dev(p[a:b]) for a, b in shards]
            ptr = lambda ts: [t.data_ptr() if t.numel() else 0 for t in ts]
            want = oracle.batch_mul(curve, k, p, nthreads=16)
            for consumer in sorted({0, len(devices) - 1}):
                do = [torch.zeros((b - a, L), dtype=torch.int64, device="cuda") for a, b in shards]
                full = torch.zeros((n, L), dtype=torch.int64, device="cuda")
                torch.cuda.synchronize()
                ctx.multi_batch_mul_dev(curve, ptr(dk), ptr(dp), ptr(do), counts, full.data_ptr(), consumer)
                assert np.array_equal(full.cpu().numpy().view(np.uint64), want), (devices, curve, consumer)
                for (a, b), t in zip(shards, do):
                    assert np.array_equal(t.cpu().numpy().view(np.uint64), want[a:b])
            # no gather: the shards' own outputs only
            do = [torch.zeros((b - a, L), dtype=torch.int64, device="cuda") for a, b in shards]
            torch.cuda.synchronize()
            ctx.multi_batch_mul_dev(curve, ptr(dk), ptr(dp), ptr(do), counts)
            for (a, b), t in zip(shards, do):
                assert np.array_equal(t.cpu().numpy().view(np.uint64), want[a:b])
            # fixed base: the generator (bases = None), then a projective base of the caller's own on every device
            full = torch.zeros((n, L), dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            ctx.multi_batch_mul_fixed_dev(curve, ptr(dk), None, ptr(do), counts, full.data_ptr(), 0)
            g = one.generator(curve)
            assert np.array_equal(full.cpu().numpy().view(np.uint64), oracle.batch_mul_fixed(curve, k, g, nthreads=16))
            base = V.points(1, curve, 9103)[0]
            dbase = [_dev(base) for _ in devices]
            ctx.multi_batch_mul_fixed_dev(curve, ptr(dk), ptr(dbase), ptr(do), counts, full.data_ptr(), 0)
            assert np.array_equal(full.cpu().numpy().view(np.uint64), oracle.batch_mul_fixed(curve, k, base, nthreads=16))
            # u1*G + u2*Q
            ctx.multi_batch_double_mul_dev(curve, ptr(dk), ptr(dk2), ptr(dp), ptr(do), counts, full.data_ptr(), 0)
            assert np.array_equal(full.cpu().numpy().view(np.uint64), oracle.batch_double_mul(curve, k, k2, p, nthreads=16))


def test_multi_dev_argument_errors_and_faults(oracle):
    import torch
    import forge_ec_amd as F
    from forge_ec_amd._lib import FecError
    k, p = V.scalars(128, 1, 1), V.points(128, 1, 2)
    dk, dp = _dev(k), _dev(p)
    do = torch.zeros((128, 12), dtype=torch.int64, device="cuda")
    with F.Context(0) as one:
        with pytest.raises(FecError) as ei:   # a single-device ctx has fec_batch_mul_dev
            one.multi_batch_mul_dev(1, [dk.data_ptr()], [dp.data_ptr()], [do.data_ptr()], [128])
        assert ei.value.status == -5
    with F.Context(devices=[0, 0]) as ctx:
        ptrs = lambda t: [t.data_ptr(), t.data_ptr()]
        with pytest.raises(FecError) as ei:   # consumer out of range
            ctx.multi_batch_mul_dev(1, ptrs(dk), ptrs(dp), ptrs(do), [64, 64], do.data_ptr(), 2)
        assert ei.value.status == -1
        with pytest.raises(FecError) as ei:   # a shard with elements and no input
            ctx.multi_batch_mul_dev(1, [dk.data_ptr(), 0], ptrs(dp), ptrs(do), [64, 64])
        assert ei.value.status == -1
        with pytest.raises(FecError) as ei:   # misaligned output
            ctx.multi_batch_mul_dev(1, ptrs(dk), ptrs(dp), [do.data_ptr() + 8, do.data_ptr()], [64, 64])
        assert ei.value.status == -1
        # a scheduler fault in one shard is FEC_E_LAUNCH from the call, and the next call is right again
        ctx.debug_force_fault(True)
        do2 = torch.zeros((64, 12), dtype=torch.int64, device="cuda")
        with pytest.raises(FecError) as ei:
            ctx.multi_batch_mul_dev(1, [dk.data_ptr(), dk[64:].data_ptr()], [dp.data_ptr(), dp[64:].data_ptr()],
                                    [do.data_ptr(), do2.data_ptr()], [64, 64])
        assert ei.value.status == -4
        ctx.debug_force_fault(False)
        full = torch.zeros((128, 12), dtype=torch.int64, device="cuda")
        ctx.multi_batch_mul_dev(1, [dk.data_ptr(), dk[64:].data_ptr()], [dp.data_ptr(), dp[64:].data_ptr()],
                                [do.data_ptr(), do2.data_ptr()], [64, 64], full.data_ptr(), 1)
        assert np.array_equal(full.cpu().numpy().view(np.uint64), oracle.batch_mul(1, k, p, nthreads=8))


def test_multi_dev_caller_streams_and_shared_prefix_table(oracle):
    """(1) `streams`: the kernels of shard g are queued behind the caller's producer on the stream it names (here the
    host-to-device copies of the shard's inputs, issued on a torch stream and NOT synchronised before the call);
    (2) the shard workers of a [0, 0, 0] ctx that is asked for prefix tables all attach to ONE table on the device
    (they build concurrently from three host threads) and the fixed-base products do not change."""
    import torch
    import forge_ec_amd as F
    n, curve = 6000, 1
    counts = (2000, 2500, 1500)
    edges = np.concatenate([[0], np.cumsum(counts)])
    k, p = V.scalars(n, curve, 9300), V.points(n, curve, 9301)
    want = oracle.batch_mul(curve, k, p, nthreads=16)
    with F.Context(devices=[0, 0, 0]) as ctx:
        streams = [torch.cuda.Stream() for _ in counts]
        hk = [torch.from_numpy(k[a:b].view(np.int64)).pin_memory() for a, b in zip(edges[:-1], edges[1:])]
        hp = [torch.from_numpy(p[a:b].view(np.int64)).pin_memory() for a, b in zip(edges[:-1], edges[1:])]
        dk = [torch.empty_like(t, device="cuda") for t in hk]
        dp = [torch.empty_like(t, device="cuda") for t in hp]
        do = [torch.zeros((c, 12), dtype=torch.int64, device="cuda") for c in counts]
        full = torch.zeros((n, 12), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        for g, s in enumerate(streams):
            with torch.cuda.stream(s):
                dk[g].copy_(hk[g], non_blocking=True)
                dp[g].copy_(hp[g], non_blocking=True)
        ctx.multi_batch_mul_dev(curve, [t.data_ptr() for t in dk], [t.data_ptr() for t in dp], [t.data_ptr() for t in do],
                                counts, full.data_ptr(), 1, streams=[s.cuda_stream for s in streams])
        assert np.array_equal(full.cpu().numpy().view(np.uint64), want)
        # (2)
        free0 = torch.cuda.mem_get_info()[0]
        ctx.set_fixed_prefix_bits(20)
        g1 = oracle.generator(curve)
        got = ctx.batch_mul_fixed(curve, k, g1)            # three workers, three host threads, one table
        assert np.array_equal(got, oracle.batch_mul_fixed(curve, k, g1, nthreads=16))
        assert ctx.fixed_prefix_bits(curve) == 20
        used = free0 - torch.cuda.mem_get_info()[0]
        assert (96 << 20) <= used < 2 * (96 << 20) + (64 << 20)     # one 96 MiB table (and staging), not three
        ctx.set_side_stream_max(0)                         # the measurement knob: results do not depend on it
        k2 = V.scalars(n, curve, 9302)
        assert np.array_equal(ctx.batch_double_mul(curve, k, k2, p), oracle.batch_double_mul(curve, k, k2, p, nthreads=16))
