"""
CPU-side check of the DEVICE headers' logic: forge_ec_amd/csrc/{secp256k1,p256,ed25519}.hpp are
compiled for the host (tools/host_emul.cpp, FEC_HOST_EMUL: carry chains and selects in portable
C++ instead of gfx950 asm) and diffed against the oracle.  This isolates algorithmic errors in
the 32-bit-limb restatement from code-generation issues, and runs without a GPU.  It is not a
product path (the shipped library contains only the gfx950 build).
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import vectors as V

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "libhost_emul.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
OPS = {"add": 0, "sub": 1, "mul": 2, "sqr": 3, "neg": 4}


@pytest.fixture(scope="module")
def emu():
    if not os.path.exists(CLANG):
        pytest.skip("ROCm clang++ not available")
    src = os.path.join(ROOT, "tools", "host_emul.cpp")
    deps = [src] + [os.path.join(ROOT, "forge_ec_amd", "csrc", f) for f in
                    ("limbs.hpp", "secp256k1.hpp", "p256.hpp", "ed25519.hpp")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.check_call([CLANG, "-O2", "-std=c++17", "-fPIC", "-shared", "-o", SO, src])
    return ctypes.CDLL(SO)


def _p(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else None


@pytest.mark.parametrize("curve", [0, 1, 2])
def test_field_ops(emu, oracle, curve):
    edges = V.edge_field_values(curve)
    a = [V.limbs_of(x) for x in edges for _ in edges]
    b = [V.limbs_of(y) for _ in edges for y in edges]
    a = np.concatenate([np.array(a, dtype=np.uint64), V.field_elements(600, curve, 11),
                        V.splitmix64(2400, V.SEED, 12).reshape(-1, 4)])
    b = np.concatenate([np.array(b, dtype=np.uint64), V.field_elements(600, curve, 13),
                        V.splitmix64(2400, V.SEED, 14).reshape(-1, 4)])
    out = np.zeros(4, dtype=np.uint64)
    for i in range(a.shape[0]):
        ai, bi = np.ascontiguousarray(a[i]), np.ascontiguousarray(b[i])
        for name, op in OPS.items():
            emu.he_field_op(curve, op, _p(ai), _p(bi), _p(out))
            want = oracle.field_op(curve, name, ai, bi)
            assert np.array_equal(out, want), (curve, name, [hex(int(v)) for v in ai], [hex(int(v)) for v in bi])


@pytest.mark.parametrize("curve", [0, 1, 2])
def test_point_ops_and_multiply(emu, oracle, curve):
    pl = V.POINT_LIMBS[curve]
    g = oracle.generator(curve)
    g2 = oracle.point_double(curve, g)
    ident = oracle.identity(curve)
    pts = [g, g2, oracle.point_add(curve, g, g2), ident, oracle.point_negate(curve, g)]
    pts += list(V.points(6, curve, 21))
    out = np.zeros(pl, dtype=np.uint64)
    for p in pts:
        p = np.ascontiguousarray(p)
        for q in pts:
            q = np.ascontiguousarray(q)
            emu.he_point_op(curve, 0, _p(p), _p(q), _p(out))
            assert np.array_equal(out, oracle.point_add(curve, p, q))
        emu.he_point_op(curve, 1, _p(p), None, _p(out))
        assert np.array_equal(out, oracle.point_double(curve, p))
    ks = [V.limbs_of(v) for v in (0, 1, 2, 3, 1 << 255, (1 << 256) - 1, 0x80, 1 << 248)]
    ks += [list(r) for r in V.scalars(12, curve, 22)]
    bases = [g] * 8 + list(V.points(12, curve, 23))
    for k, base in zip(ks, bases):
        k = np.array(k, dtype=np.uint64)
        base = np.ascontiguousarray(base)
        emu.he_multiply(curve, _p(base), _p(k), _p(out))
        assert np.array_equal(out, oracle.multiply(curve, base, k)), (curve, [hex(int(v)) for v in k])


def test_secp256k1_square_fallback_is_exercised(emu, oracle):
    """Operands whose ripples travel (tests/golden/secp256k1_sqr_ripple_operands.json) must reach
    sqr()'s cold continuation blocks and still match the oracle."""
    import json
    ops = json.load(open(os.path.join(ROOT, "tests", "golden", "secp256k1_sqr_ripple_operands.json")))["operands"]
    emu.he_rare_sqr_count.restype = ctypes.c_ulong
    out = np.zeros(4, dtype=np.uint64)
    taken = 0
    for limbs in ops:
        a = np.array(limbs, dtype=np.uint64)
        before = emu.he_rare_sqr_count()
        emu.he_field_op(0, OPS["sqr"], _p(a), None, _p(out))
        assert emu.he_rare_sqr_count() > before, "operand does not reach a cold block any more"
        taken += 1
        assert np.array_equal(out, oracle.field_op(0, "sqr", a))
    assert taken == len(ops)


def test_ed25519_fixed_base_table_walk(emu, oracle):
    """multiply_fixed (LDS addend table, set bits only) == the reference's 256-step loop."""
    g = oracle.generator(2)
    ks = [V.limbs_of(v) for v in (0, 1, 2, 3, 1 << 255, (1 << 256) - 1, 0x80, 1 << 31, 1 << 32, 0x8000000100000000)]
    ks += [list(r) for r in V.scalars(10, 2, 31)]
    bases = [g] * len(ks)
    bases[-3:] = list(V.points(3, 2, 32))
    bases[4] = oracle.identity(2)
    out = np.zeros(16, dtype=np.uint64)
    for k, base in zip(ks, bases):
        k = np.array(k, dtype=np.uint64)
        base = np.ascontiguousarray(base)
        emu.he_ed_multiply_fixed(_p(base), _p(k), _p(out))
        assert np.array_equal(out, oracle.multiply(2, base, k)), [hex(int(v)) for v in k]


@pytest.mark.parametrize("curve", [0, 1, 2])
def test_to_affine(emu, oracle, curve):
    g = oracle.generator(curve)
    pts = [g, oracle.point_double(curve, g), oracle.identity(curve)] + list(V.points(4, curve, 41))
    pts.append(oracle.multiply(curve, g, V.limbs_of(0x123456789ABCDEF)))
    out = np.zeros(8, dtype=np.uint64)
    for p in pts:
        p = np.ascontiguousarray(p)
        inf = emu.he_to_affine(curve, _p(p), _p(out))
        want, winf = oracle.to_affine(curve, p)
        assert bool(inf) == winf and np.array_equal(out, want)


def test_secp256k1_scalar_field(emu, oracle):
    """Scalar Mul (low 256 bits of the product, then reduce) and invert, vs the oracle."""
    n = V.ORDER[0]
    nref = 0xFFFFFFFFFFFFFFFEFFFFFFFFFFFFFFFFBAAEDCE6AF48A03BBFD25E8CD0364141  # the reference's N (limbs swapped)
    vals = [1, 2, n - 1, n, n + 1, nref - 1, nref, nref + 1, (1 << 256) - 1, 1 << 128, (1 << 128) - 1, 1 << 255]
    a = np.concatenate([np.array([V.limbs_of(v) for v in vals], dtype=np.uint64), V.scalars(60, 0, 51),
                        V.splitmix64(120, V.SEED, 52).reshape(-1, 4)])
    b = np.concatenate([np.array([V.limbs_of(v) for v in reversed(vals)], dtype=np.uint64), V.scalars(60, 0, 53),
                        V.splitmix64(120, V.SEED, 54).reshape(-1, 4)])
    out = np.zeros(4, dtype=np.uint64)
    for i in range(a.shape[0]):
        ai, bi = np.ascontiguousarray(a[i]), np.ascontiguousarray(b[i])
        emu.he_secp_scalar_op(0, _p(ai), _p(bi), _p(out))
        assert np.array_equal(out, oracle.secp256k1_scalar_op("mul", ai, bi)[0])
    for i in range(0, 20):
        ai = np.ascontiguousarray(a[i])
        want, ok = oracle.secp256k1_scalar_op("inv", ai)
        if ok:
            emu.he_secp_scalar_op(1, _p(ai), None, _p(out))
            assert np.array_equal(out, want)


def test_p256_scalar_field(emu, oracle):
    """P-256 Scalar Mul (exact product, then the reference's lossy reduce_wide), invert, the default ct_lt
    and compare_with_n as the device headers compute them, vs the C oracle and the Python model."""
    from oracle import py_model as M
    S = M.P256Scalar
    n = S.N
    vals = [0, 1, 2, 3, (1 << 64) - 1, 1 << 64, (1 << 128) - 1, 1 << 128, 1 << 192, n - 2, n - 1, n, n + 1,
            (1 << 256) - n, (1 << 256) - 1, 1 << 255, (1 << 224) - 1, 0xFF << 248, 0xFE << 248]
    a = np.concatenate([np.array([V.limbs_of(v) for v in vals], dtype=np.uint64), V.scalars(60, 1, 55),
                        V.splitmix64(400, V.SEED, 56).reshape(-1, 4)])
    b = np.concatenate([np.array([V.limbs_of(v) for v in reversed(vals)], dtype=np.uint64), V.scalars(60, 1, 57),
                        V.splitmix64(400, V.SEED, 58).reshape(-1, 4)])
    out = np.zeros(4, dtype=np.uint64)
    second_round = carry_round = 0
    for i in range(a.shape[0]):
        ai, bi = np.ascontiguousarray(a[i]), np.ascontiguousarray(b[i])
        emu.he_p256_scalar_op(0, _p(ai), _p(bi), _p(out))
        assert np.array_equal(out, oracle.p256_scalar_op("mul", ai, bi)[0]), i
        la, lb = [int(v) for v in ai], [int(v) for v in bi]
        assert [int(v) for v in out] == S.mul(la, lb)
        w = S.val(la) * S.val(lb)
        first = (w & S.M256) + (w >> 256) * S.val(S.C)
        second_round += (first >> 256) != 0
        carry_round += (first >> 256) != 0 and ((first & S.M256) + (((first >> 256) * S.val(S.C)) & S.M256)) >> 256 != 0
        assert emu.he_p256_scalar_op(2, _p(ai), _p(bi), None) == int(S.ct_lt_default(la, lb))
        assert emu.he_p256_scalar_op(3, _p(ai), None, None) == int(S.val(la) >= n)
    # every branch of reduce_wide is taken by these operands: no second round, second round, its carry round
    assert 0 < second_round < a.shape[0] and carry_round > 50
    for i in range(0, 24):
        ai = np.ascontiguousarray(a[i])
        want, ok = oracle.p256_scalar_op("inv", ai)
        if ok:
            emu.he_p256_scalar_op(1, _p(ai), None, _p(out))
            assert np.array_equal(out, want)


def test_scalar_add_both_curves(emu, oracle):
    """Scalar Add as the device headers compute it (secp256k1: carry dropped, one reduce; P-256: reduce() sees
    only the low 256 bits after a carry out) vs the oracle."""
    out = np.zeros(4, dtype=np.uint64)
    for curve, fn, op in ((0, emu.he_secp_scalar_op, oracle.secp256k1_scalar_op), (1, emu.he_p256_scalar_op, oracle.p256_scalar_op)):
        n = V.ORDER[curve]
        nref = 0xFFFFFFFFFFFFFFFEFFFFFFFFFFFFFFFFBAAEDCE6AF48A03BBFD25E8CD0364141 if curve == 0 else n
        vals = [0, 1, n - 1, n, nref - 1, nref, nref + 1, (1 << 256) - 1, 1 << 255, (1 << 256) - nref, (1 << 256) - nref - 1]
        pairs = [(V.limbs_of(x), V.limbs_of(y)) for x in vals for y in vals]
        w = V.splitmix64(800, V.SEED, 650 + curve).reshape(-1, 4)
        pairs += [([int(v) for v in w[i]], [int(v) for v in w[i + 1]]) for i in range(0, 200, 2)]
        for x, y in pairs:
            ax, ay = np.array(x, dtype=np.uint64), np.array(y, dtype=np.uint64)
            fn(4, _p(ax), _p(ay), _p(out))
            assert np.array_equal(out, op("add", ax, ay)[0]), (curve, x, y)
