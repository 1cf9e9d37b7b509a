"""
The boundary's failure paths on the GPU (VERDICT r2 "do this" 1): a fault that a scheduler kernel
reports reaches the caller as FEC_E_LAUNCH -- never FEC_OK with zeroed points --, the popcount sort of the
Ed25519 fixed-base kernel survives heavy scalars in several sort blocks, destroy wipes before it frees.
"""
import ctypes

import numpy as np
import pytest

import vectors as V

pytestmark = pytest.mark.gpu

FEC_E_LAUNCH = -4


def _pk(n, curve, seed):
    return np.ascontiguousarray(np.concatenate([V.field_elements(n, curve, seed), V.field_elements(n, curve, seed + 1)], axis=1))


def test_forced_scheduler_fault_is_an_error_not_zero_points(oracle):
    """fec_ctx_debug_force_fault makes the scheduler kernels take their watchdog exit: every host-pointer entry point
    that runs one returns FEC_E_LAUNCH, a *_dev launch is caught by fec_ctx_check, the state clears on read and the
    next call is correct again."""
    import torch
    import forge_ec_amd as F
    n = 700
    rng = np.random.default_rng(11)
    dg = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    with F.Context(0) as ctx:
        calls = {
            "p256 batch_mul": lambda: ctx.batch_mul(1, V.scalars(n, 1, 1), V.points(n, 1, 2)),
            "p256 batch_mul_fixed": lambda: ctx.batch_mul_fixed(1, V.scalars(n, 1, 3), ctx.generator(1)),
            "ed25519 batch_mul": lambda: ctx.batch_mul(2, V.scalars(n, 2, 4), V.points(n, 2, 5)),
            "p256 double_mul": lambda: ctx.batch_double_mul(1, V.scalars(n, 1, 6), V.scalars(n, 1, 7), V.points(n, 1, 8)),
            "ed25519 double_mul": lambda: ctx.batch_double_mul(2, V.scalars(n, 2, 9), V.scalars(n, 2, 10), V.points(n, 2, 11)),
            "ecdsa_verify_p256": lambda: ctx.ecdsa_verify_p256(dg, V.scalars(n, 1, 12), V.scalars(n, 1, 13), _pk(n, 1, 14), None),
            "eddsa_verify_ed25519": lambda: ctx.eddsa_verify_ed25519(_pk(n, 2, 16), None, _pk(n, 2, 18), None, V.scalars(n, 2, 20), V.scalars(n, 2, 21)),
            "ecdh p256": lambda: ctx.batch_ecdh(1, V.scalars(n, 1, 22), _pk(n, 1, 23), None),
            "schnorr_verify p256": lambda: ctx.schnorr_verify(1, _pk(n, 1, 34), _pk(n, 1, 36), V.scalars(n, 1, 38), V.scalars(n, 1, 39)),
            "schnorr_batch_verify p256": lambda: ctx.schnorr_batch_verify(1, _pk(64, 1, 42), _pk(64, 1, 44), V.scalars(64, 1, 46), V.scalars(64, 1, 47), V.scalars(64, 1, 48)),
            "validate ed25519": lambda: ctx.batch_validate_point(2, _pk(n, 2, 25), None),
            "msm p256": lambda: ctx.multi_scalar_mul(1, V.scalars(64, 1, 27), V.points(64, 1, 28)),
            "ecdsa_batch_verify p256": lambda: ctx.ecdsa_batch_verify(1, dg[:64], V.scalars(64, 1, 29), V.scalars(64, 1, 30), _pk(64, 1, 31), None, V.scalars(64, 1, 33)),
        }
        ctx.debug_force_fault(True)
        for name, call in calls.items():
            with pytest.raises(F.FecError) as ei:
                call()
            assert ei.value.status == FEC_E_LAUNCH, (name, ei.value.status)
        # device-pointer form: the launch itself succeeds (it is only enqueued), fec_ctx_check reports the fault
        dev = torch.device("cuda:0")
        k = torch.from_numpy(V.scalars(n, 1, 40).view(np.int64).copy()).to(dev)
        p = torch.from_numpy(V.points(n, 1, 41).view(np.int64).copy()).to(dev)
        o = torch.full_like(p, 0x55)
        torch.cuda.synchronize()
        ctx.batch_mul_dev(1, k.data_ptr(), p.data_ptr(), o.data_ptr(), n)
        with pytest.raises(F.FecError) as ei:
            ctx.check()
        assert ei.value.status == FEC_E_LAUNCH
        assert int(o.abs().sum().item()) == 0      # the outputs of the faulted launch are zero-filled on top
        ctx.check()                                # reading cleared the state
        ctx.debug_force_fault(False)
        # and the ctx is fully usable again: same calls, oracle-identical
        k1, p1 = V.scalars(n, 1, 1), V.points(n, 1, 2)
        assert np.array_equal(ctx.batch_mul(1, k1, p1), oracle.batch_mul(1, k1, p1, nthreads=8))
        k2, p2 = V.scalars(n, 2, 4), V.points(n, 2, 5)
        assert np.array_equal(ctx.batch_mul(2, k2, p2), oracle.batch_mul(2, k2, p2, nthreads=8))
        ctx.check()
    # a kernel that has no scheduler is not affected by the hook
    with F.Context(0) as ctx:
        ctx.debug_force_fault(True)
        k0, p0 = V.scalars(300, 0, 50), V.points(300, 0, 51)
        assert np.array_equal(ctx.batch_mul(0, k0, p0), oracle.batch_mul(0, k0, p0, nthreads=8))


def test_forced_fault_on_a_multi_ctx_comes_back_from_the_shards():
    import forge_ec_amd as F
    n = 900
    with F.Context(devices=[0, 0]) as ctx:
        ctx.debug_force_fault(True)
        with pytest.raises(F.FecError) as ei:
            ctx.batch_mul(1, V.scalars(n, 1, 1), V.points(n, 1, 2))
        assert ei.value.status == FEC_E_LAUNCH
        ctx.debug_force_fault(False)
        ctx.check()
        got = ctx.batch_mul(1, V.scalars(n, 1, 1), V.points(n, 1, 2))
        assert got.shape == (n, 12)


def test_ed25519_fixed_base_sort_with_heavy_scalars_in_many_blocks(oracle):
    """scalars of popcount 252..256 in SEVERAL 4096-element sort blocks of a batch >= 2^16.  With the
    round-2 header layout the cursors of bins 252..256 were overwritten by perm[0..4] after the first block scattered,
    and later blocks wrote their heavy elements through garbage cursors (missing / out-of-range outputs)."""
    n = (1 << 16) + 4099
    k = V.scalars(n, 2, 7001).copy()
    ones = np.full(4, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
    m252 = np.array([0xFFFFFFFFFFFFFFFF] * 3 + [0x0FFFFFFFFFFFFFFF], dtype=np.uint64)          # 2^252 - 1
    heavy = {}
    for j, i in enumerate(range(17, n, 5000)):       # one or two per sort block, all over the batch
        v = ones.copy()
        if j % 5 == 1:
            v = m252.copy()
        elif j % 5 == 2:
            v[0] &= ~np.uint64(1)                     # popcount 255
        elif j % 5 == 3:
            v[1] &= ~np.uint64(0b101)                 # popcount 254
        elif j % 5 == 4:
            v[2] &= ~np.uint64(0b111)                 # popcount 253
        k[i] = v
        heavy[i] = v
    k[3] = 0                                          # a zero scalar and single-bit scalars as well
    k[4] = np.array([1, 0, 0, 0], dtype=np.uint64)
    k[n - 1] = ones
    import forge_ec_amd as F
    with F.Context(0) as ctx:
        g = ctx.generator(2)
        got = ctx.batch_mul_fixed(2, k, g)
        # the whole batch against the oracle
        want = oracle.batch_mul_fixed(2, k, g, nthreads=16)
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert len(bad) == 0, "first mismatch at element %d (heavy: %s)" % (bad[0], bad[0] in heavy)
        # the sort is per launch: a second, differently laid out batch on the same ctx (work area reused)
        k2 = np.ascontiguousarray(k[::-1])
        got2 = ctx.batch_mul_fixed(2, k2, g)
        assert np.array_equal(got2, want[::-1])


def test_destroy_and_wipe_report_success():
    """fec_ctx_wipe's status is checked on a ctx that has used every kind of buffer; destroy (wipe first, then free)
    leaves a second ctx on the same device intact."""
    import forge_ec_amd as F
    from forge_ec_amd import _lib
    L = _lib.lib()
    n = 2000
    other = F.Context(0)
    k, p = V.scalars(n, 1, 60), V.points(n, 1, 61)
    want = other.batch_mul(1, k, p)
    for _ in range(3):
        h = ctypes.c_void_p()
        assert L.fec_ctx_create(ctypes.byref(h), 0) == 0
        c = F.Context.__new__(F.Context)
        c._lib, c._h, c.device = L, h, 0
        c.batch_double_mul(2, V.scalars(n, 2, 62), V.scalars(n, 2, 63), V.points(n, 2, 64))   # staging + stream scratch
        c.batch_ecdh(1, V.scalars(n, 1, 65), _pk(n, 1, 66), None)
        assert L.fec_ctx_wipe(h) == 0
        assert L.fec_ctx_check(h) == 0
        L.fec_ctx_destroy(h)
        c._h = None
        # the other ctx's buffers were not touched by the destroyed ctx's wipe
        assert np.array_equal(other.batch_mul(1, k, p), want)
    other.close()


def test_prefix_table_policy_and_faults(oracle):
    """A ctx left to its defaults builds a prefix table only once it has multiplied 2^21 scalars by the generator; after
    fec_ctx_set_fixed_prefix_bits the next fixed-base launch builds it.  The build does not run a scheduler kernel, so
    the forced fault hits the multiplication that follows it: the call fails, the table stays, the next call is right."""
    import forge_ec_amd as F
    n = 500
    k = V.scalars(n, 1, 71)
    with F.Context(0) as ctx:
        g = ctx.generator(1)
        want = oracle.batch_mul_fixed(1, k, oracle.generator(1), nthreads=8)
        assert np.array_equal(ctx.batch_mul_fixed(1, k, g), want)
        assert ctx.fixed_prefix_bits(1) == 0          # 500 multiplications: no table
        ctx.set_fixed_prefix_bits(9)
        ctx.debug_force_fault(True)
        with pytest.raises(F.FecError) as ei:
            ctx.batch_mul_fixed(1, k, g)
        assert ei.value.status == FEC_E_LAUNCH
        ctx.debug_force_fault(False)
        assert np.array_equal(ctx.batch_mul_fixed(1, k, g), want)
        assert ctx.fixed_prefix_bits(1) == 9


def test_prefix_table_is_shared_budgeted_and_never_built_by_a_default_dev_call(oracle):
    """The table policy of fecgpu.h: (1) a ctx left to its defaults never builds from a *_dev entry point, it does from a
    host-pointer one; (2) a second ctx on the same device attaches to the SAME allocation (no second table's worth of
    memory); (3) the budget caps the size by the device's free memory -- a 24-bit request shrinks -- and a budget of
    0 % means no table at all; results are identical in every case."""
    import torch
    import forge_ec_amd as F
    n = 4096
    k = V.scalars(n, 1, 881)
    want = oracle.batch_mul_fixed(1, k, oracle.generator(1), nthreads=16)
    dk = torch.from_numpy(k.view(np.int64)).cuda()
    do = torch.zeros((n, 12), dtype=torch.int64, device="cuda")

    def dev_call(ctx):
        ctx.batch_mul_fixed_dev(1, dk.data_ptr(), ctx.generator_dev(1), do.data_ptr(), n)
        ctx.check()
        torch.cuda.synchronize()
        return do.cpu().numpy().view(np.uint64)

    with F.Context(0) as a:
        a.set_fixed_prefix_after(n)                  # defaults otherwise: 24 bits wanted, not asked for explicitly
        for _ in range(3):
            assert np.array_equal(dev_call(a), want)
        assert a.fixed_prefix_bits(1) == 0           # (1) three *_dev launches past the threshold: still no table
        free0 = torch.cuda.mem_get_info()[0]
        assert np.array_equal(a.batch_mul_fixed(1, k, a.generator(1)), want)   # a host-pointer call builds it
        assert a.fixed_prefix_bits(1) == 24
        used = free0 - torch.cuda.mem_get_info()[0]
        assert used >= (96 << 24)                    # the P-256 table: 1.5 GiB
        assert np.array_equal(dev_call(a), want)     # ... and the *_dev launches use it
        with F.Context(0) as b:                      # (2) a second ctx on the device shares it
            free1 = torch.cuda.mem_get_info()[0]
            b.build_fixed_prefix(1)
            assert b.fixed_prefix_bits(1) == 24
            assert free1 - torch.cuda.mem_get_info()[0] < (96 << 24) // 4
            assert np.array_equal(dev_call(b), want)
        assert np.array_equal(dev_call(a), want)     # b is gone, a's reference keeps the table
    with F.Context(0) as c:                          # (3) the budget: a share of what is FREE at that moment
        free = torch.cuda.mem_get_info()[0]
        ballast = None
        for leave in (4 << 30, 5 << 30, 6 << 30):    # leave about 4 GiB free (an allocation of that size may be refused: try less)
            try:
                ballast = torch.empty(free - leave, dtype=torch.uint8, device="cuda")
                break
            except RuntimeError:
                torch.cuda.empty_cache()
        if ballast is not None and torch.cuda.mem_get_info()[0] < (7 << 30):
            c.set_fixed_prefix_budget(25)            # 1 - 1.7 GiB: 2^24 entries (1.5 + 0.75 GiB) do not fit, 2^22 do
            c.set_fixed_prefix_bits(24)
            assert np.array_equal(c.batch_mul_fixed(1, k, c.generator(1)), want)
            assert 16 <= c.fixed_prefix_bits(1) < 24
        else:                                        # no ballast to be had: the same shrink through a budget of 1 % is not possible
            c.set_fixed_prefix_budget(1)             # on a 288 GB device (2.9 GB > 2.25 GiB); the arithmetic is still exercised
            c.set_fixed_prefix_bits(24)
            assert np.array_equal(c.batch_mul_fixed(1, k, c.generator(1)), want)
            assert 16 <= c.fixed_prefix_bits(1) <= 24
        c.set_fixed_prefix_budget(0)
        c.set_fixed_prefix_bits(24)
        assert np.array_equal(c.batch_mul_fixed(1, k, c.generator(1)), want)
        assert c.fixed_prefix_bits(1) == 0           # refused: no table, the same products
        assert np.array_equal(dev_call(c), want)
        del ballast
        torch.cuda.empty_cache()
