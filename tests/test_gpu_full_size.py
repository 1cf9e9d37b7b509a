"""
Full-size parity on the GPU at BASELINE.json's configured sizes.

The reference's point arithmetic is not a group, so algebraic properties
(linearity, round trips) do not hold even for the reference itself.  The size-independent checks
used instead:
  * a seeded random sample of indices plus the first/last 4096 rows, compared bit-exactly with
    the oracle (SURVEY section 8d prescribes exactly this for n = 2^22);
  * batch invariance: rows recomputed in small batches (different workgroup/lane placement, a
    ragged tail) equal their value inside the large batch;
  * run-to-run determinism of the whole output (checksum of the full tensor).
Inputs are generated on the host from the seeded synth streams and live in HBM (torch tensors).
"""
import hashlib

import numpy as np
import pytest

import vectors as V

pytestmark = pytest.mark.gpu


def _sample_idx(n, k, seed):
    rng = np.random.default_rng(seed)
    head = np.arange(min(4096, n))
    tail = np.arange(max(0, n - 4096), n)
    mid = rng.integers(0, n, size=k)
    return np.unique(np.concatenate([head, tail, mid]))


def _run_dev(ctx, kind, curve, arrays, n):
    import torch
    limbs = V.POINT_LIMBS[curve]
    dev = [torch.from_numpy(a.view(np.int64)).cuda() for a in arrays]
    out = torch.empty((n, limbs), dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    if kind == "var":
        ctx.batch_mul_dev(curve, dev[0].data_ptr(), dev[1].data_ptr(), out.data_ptr(), n, st)
    elif kind == "fixed":
        ctx.batch_mul_fixed_dev(curve, dev[0].data_ptr(), ctx.generator_dev(curve), out.data_ptr(), n, st)
    else:
        ctx.batch_double_mul_dev(curve, dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), out.data_ptr(), n, st)
    torch.cuda.synchronize()
    return out.cpu().numpy().view(np.uint64)


def _digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_config2_secp256k1_variable_base_2p20(gpu_ctx, oracle):
    n, curve = 1 << 20, 0
    k, p = V.scalars(n, curve, 2001), V.points(n, curve, 2002)
    out = _run_dev(gpu_ctx, "var", curve, [k, p], n)
    idx = _sample_idx(n, 6000, 1)
    want = oracle.batch_mul(curve, k[idx], p[idx], nthreads=16)
    assert np.array_equal(out[idx], want)
    # batch invariance on a ragged sub-batch taken from the middle
    lo = 123457
    sub = gpu_ctx.batch_mul(curve, k[lo:lo + 777], p[lo:lo + 777])
    assert np.array_equal(sub, out[lo:lo + 777])
    # determinism
    assert _digest(out) == _digest(_run_dev(gpu_ctx, "var", curve, [k, p], n))


def test_ed25519_fixed_base_batch_wide_popcount_sort(gpu_ctx, oracle):
    """From 2^16 elements on the table kernel walks a permutation of the whole batch sorted by popcount
    (kernels_ed.hip: k_ed_pc_hist / _scan / _scatter, k_ed_fixed_sorted).  A ragged batch just above the threshold
    with zero scalars, single-bit scalars, all-ones scalars and long runs of equal popcount, every element
    compared with the oracle; and the same rows through the small-batch kernel (in-kernel quartile sort)."""
    n, curve = (1 << 16) + 37, 2
    k = V.scalars(n, curve, 2301)
    k[5] = 0                                   # multiply's zero-scalar early-out (popcount 0: sorts last)
    k[n - 1] = 0
    k[100:164] = np.uint64(0xFFFFFFFFFFFFFFFF)  # popcount 256, consumed as is (no reduction mod l)
    for i in range(256):
        k[1000 + i] = 0
        k[1000 + i, i // 64] = np.uint64(1) << np.uint64(i % 64)   # a single addend each: results are table entries
    k[3000:9000, 1:] = 0                       # low popcounts
    k[20000:30000] = k[20000]                  # one popcount bin with 10^4 elements
    dev_out = _run_dev(gpu_ctx, "fixed", curve, [k], n)
    g = oracle.generator(curve)
    want = oracle.batch_mul_fixed(curve, k, g, nthreads=16)
    assert np.array_equal(dev_out, want)
    for lo, cnt in ((0, 300), (900, 400), (n - 300, 300)):
        assert np.array_equal(gpu_ctx.batch_mul_fixed(curve, k[lo:lo + cnt], g), want[lo:lo + cnt])
    # run-to-run determinism although positions inside a popcount bin are handed out by atomics
    assert _digest(dev_out) == _digest(_run_dev(gpu_ctx, "fixed", curve, [k], n))


def test_config3_ed25519_fixed_base_2p20(gpu_ctx, oracle):
    n, curve = 1 << 20, 2
    k = V.scalars(n, curve, 2003)
    # a second population below the group order l (SURVEY section 8d asks for both)
    k[n // 2:, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
    out = _run_dev(gpu_ctx, "fixed", curve, [k], n)
    g = oracle.generator(curve)
    idx = _sample_idx(n, 8000, 2)
    want = oracle.batch_mul_fixed(curve, k[idx], g, nthreads=16)
    assert np.array_equal(out[idx], want)
    # the table kernel must agree with the generic (variable-base) kernel on the same inputs
    lo = 500001
    pts = np.tile(g, (1500, 1))
    assert np.array_equal(gpu_ctx.batch_mul(curve, k[lo:lo + 1500], pts), out[lo:lo + 1500])


def test_config4_p256_variable_base_2p22_sharded(gpu_ctx, oracle):
    """2^22 P-256 variable-base, computed as 8 contiguous shards exactly as 8 ranks would."""
    from forge_ec_amd.dist import shard_range
    n, curve, world = 1 << 22, 1, 8
    k, p = V.scalars(n, curve, 2004), V.points(n, curve, 2005)
    out = np.empty((n, 12), dtype=np.uint64)
    for r in range(world):
        lo, hi = shard_range(n, r, world)
        out[lo:hi] = _run_dev(gpu_ctx, "var", curve, [k[lo:hi], p[lo:hi]], hi - lo)
    idx = _sample_idx(n, 6000, 3)
    want = oracle.batch_mul(curve, k[idx], p[idx], nthreads=16)
    assert np.array_equal(out[idx], want)
    # one unsharded launch gives the same bits as the 8 shards
    whole = _run_dev(gpu_ctx, "var", curve, [k, p], n)
    assert _digest(whole) == _digest(out)


def test_config5_secp256k1_double_mul_2p20(gpu_ctx, oracle):
    n, curve = 1 << 20, 0
    u1, u2, q = V.scalars(n, curve, 2006), V.scalars(n, curve, 2007), V.points(n, curve, 2008)
    out = _run_dev(gpu_ctx, "double", curve, [u1, u2, q], n)
    idx = _sample_idx(n, 1500, 4)[::3]
    want = oracle.batch_double_mul(curve, u1[idx], u2[idx], q[idx], nthreads=16)
    assert np.array_equal(out[idx], want)
    # R = multiply(G,u1) + multiply(Q,u2) rebuilt from the single-mul kernel and the point-add kernel
    lo = 77777
    g = oracle.generator(curve)
    a = gpu_ctx.batch_mul_fixed(curve, u1[lo:lo + 600], g)
    b = gpu_ctx.batch_mul(curve, u2[lo:lo + 600], q[lo:lo + 600])
    assert np.array_equal(gpu_ctx.point_op(curve, 0, a, b), out[lo:lo + 600])


@pytest.mark.parametrize("curve", [0, 1, 2])
def test_whole_batch_is_bit_exact(gpu_ctx, oracle, curve):
    """Every one of the 2^20 (secp256k1: 2^19) results of a variable-base batch equals the oracle's -- no
    sampling.  The oracle runs on all granted host cores (about 25 s for secp256k1)."""
    n = 1 << (19 if curve == 0 else 20)
    k, p = V.scalars(n, curve, 2101 + curve), V.points(n, curve, 2201 + curve)
    out = _run_dev(gpu_ctx, "var", curve, [k, p], n)
    want = oracle.batch_mul(curve, k, p, nthreads=16)
    bad = np.nonzero((out != want).any(axis=1))[0]
    assert bad.size == 0, "first mismatching rows: %s" % bad[:5]


@pytest.mark.parametrize("curve", [0, 1])
def test_ecdsa_verify_2p20(gpu_ctx, oracle, curve):
    """Ecdsa::<C, D>::verify over 2^20 signatures (secp256k1, P-256): an oracle-checked sample, batch invariance
    on a ragged sub-batch, and signatures that verify under the reference's arithmetic scattered through it."""
    n = 1 << 20
    rng = np.random.default_rng(3100 + curve)
    dg = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    dg[::5, 0] &= 0x7F                       # a fifth of the digests are certainly below n; others may panic (status 2)
    r, s = V.scalars(n, curve, 3101), V.scalars(n, curve, 3102)
    pk = np.ascontiguousarray(np.concatenate([V.field_elements(n, curve, 3103), V.field_elements(n, curve, 3104)], axis=1))
    inf = np.zeros(n, dtype=np.uint8)
    idx = _sample_idx(n, 1200, 7)[::4]
    # with the key at infinity R = multiply(G, h / s) whatever r is: set r to the x the reference derives there
    ver = idx[::3]
    op = oracle.secp256k1_scalar_op if curve == 0 else oracle.p256_scalar_op
    g = oracle.generator(curve)
    one = np.array([1, 0, 0, 0], dtype=np.uint64)
    for i in ver:
        inf[i] = 1
        dg[i, 0] &= 0x7F
        h = np.array(V.limbs_of(int.from_bytes(dg[i].tobytes(), "big")), dtype=np.uint64)
        u1 = op("mul", h, op("inv", s[i])[0])[0]
        xy, is_inf = oracle.to_affine(curve, oracle.multiply(curve, g, u1))
        x = oracle.field_op(0, "mul", xy[:4], one) if curve == 0 else xy[:4]
        if not is_inf:
            r[i] = x
    fn = gpu_ctx.ecdsa_verify_secp256k1 if curve == 0 else gpu_ctx.ecdsa_verify_p256
    ofn = oracle.batch_secp256k1_ecdsa_verify if curve == 0 else oracle.batch_p256_ecdsa_verify
    got = fn(dg, r, s, pk, inf)
    want = ofn(dg[idx], r[idx], s[idx], pk[idx], inf[idx], nthreads=16)
    assert np.array_equal(got[idx], want)
    assert int((want == 1).sum()) >= len(ver) // 4 and set(int(v) for v in np.unique(got)) <= {0, 1, 2}
    lo = 345679
    assert np.array_equal(fn(dg[lo:lo + 777], r[lo:lo + 777], s[lo:lo + 777], pk[lo:lo + 777], inf[lo:lo + 777]), got[lo:lo + 777])


def test_eddsa_verify_2p20(gpu_ctx, oracle):
    """Eddsa verify (Ed25519, from the point computation on) over 2^20 signatures: oracle-checked sample with
    verifying signatures in it, and batch invariance."""
    n = 1 << 20
    s, k = V.scalars(n, 2, 3201), V.scalars(n, 2, 3202)
    pk = np.ascontiguousarray(np.concatenate([V.field_elements(n, 2, 3203), V.field_elements(n, 2, 3204)], axis=1))
    r = np.ascontiguousarray(np.concatenate([V.field_elements(n, 2, 3205), V.field_elements(n, 2, 3206)], axis=1))
    pinf, rinf = np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
    idx = _sample_idx(n, 1200, 8)[::4]
    ver = idx[::3]
    axy, _ = oracle.batch_to_affine(2, oracle.batch_mul_fixed(2, s[ver], oracle.generator(2), nthreads=16), nthreads=16)
    r[ver] = axy
    pinf[ver] = 1
    rinf[idx[1::9]] = 1
    got = gpu_ctx.eddsa_verify_ed25519(r, rinf, pk, pinf, s, k)
    want = oracle.batch_ed25519_eddsa_verify(r[idx], rinf[idx], pk[idx], pinf[idx], s[idx], k[idx], nthreads=16)
    assert np.array_equal(got[idx], want)
    assert int((want == 1).sum()) >= len(ver) // 2
    lo = 234571
    sl = slice(lo, lo + 777)
    assert np.array_equal(gpu_ctx.eddsa_verify_ed25519(r[sl], rinf[sl], pk[sl], pinf[sl], s[sl], k[sl]), got[sl])
