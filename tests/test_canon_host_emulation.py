"""
Canonical-math mode (NOT reference parity): the device header canon_curves.hpp compiled for the host
(tools/host_emul.cpp) against the big-integer model oracle/canon_model.py, which is itself pinned by
published points (multiples of G for secp256k1, the RFC 6979 A.2.5 key pair for P-256).  No GPU
needed; the GPU build of the same header is checked in tests/test_gpu_canon.py.
"""
import ctypes
import os
import random
import subprocess

import numpy as np
import pytest

from oracle import canon_model as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "libhost_emul.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
OPS = {"add": 0, "sub": 1, "mul": 2, "sqr": 3, "neg": 4, "inv": 5}
CURVE_IDS = {"secp256k1": 0, "p256": 1}


def edge_values(C):
    P = C.P
    e = [0, 1, 2, 3, 977, 2**32, 2**32 + 977, 2**32 + 976, 2**64 - 1, 2**96, 2**96 - 1, 2**128, 2**192, 2**224,
         2**255, 2**255 + 19, P - 1, P - 2, P - 977, P - 2**32, (P - 1) // 2, (P + 1) // 2, 2**256 - 2**33,
         P - 2**224, 0xFFFFFFFF * (2**224), 2**224 - 1, P - 2**96, 2**256 - P, 2**256 - P - 1, P - 2**192]
    return sorted({v % P for v in e})


@pytest.fixture(scope="module")
def emu():
    if not os.path.exists(CLANG):
        pytest.skip("ROCm clang++ not available")
    src = os.path.join(ROOT, "tools", "host_emul.cpp")
    deps = [src] + [os.path.join(ROOT, "forge_ec_amd", "csrc", f) for f in
                    ("limbs.hpp", "secp256k1.hpp", "p256.hpp", "ed25519.hpp", "canon_curves.hpp")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.check_call([CLANG, "-O2", "-std=c++17", "-fPIC", "-shared", "-o", SO, src])
    return ctypes.CDLL(SO)


@pytest.fixture(params=["secp256k1", "p256"])
def curve(request):
    return M.CURVES[request.param], CURVE_IDS[request.param]


def _arr(v):
    return np.array(M.limbs(v), dtype=np.uint64)


class _Ptr(ctypes.c_void_p):
    """c_void_p that keeps its numpy array alive for the duration of the call."""


def _p(a):
    if a is None:
        return None
    ptr = _Ptr(a.ctypes.data)
    ptr.keep = a
    return ptr


def test_model_is_pinned_by_published_points():
    for C in M.CURVES.values():
        assert C.on_curve(C.G)
        for k, pt in C.KNOWN_MULTIPLES.items():
            assert C.mul(k, C.G) == pt
        assert C.mul(C.N, C.G) is M.INF
        assert C.mul(C.N - 1, C.G) == C.neg(C.G)


def test_field_ops(emu, curve):
    C, cid = curve
    rng = random.Random(0xC0FFEE)
    edge = edge_values(C)
    pairs = [(a, b) for a in edge for b in edge] + [(rng.randrange(C.P), rng.randrange(C.P)) for _ in range(3000)]
    out = np.zeros(4, dtype=np.uint64)
    for a, b in pairs:
        aa, bb = _arr(a), _arr(b)
        for name, op in OPS.items():
            if name == "inv" and rng.random() > 0.03 and a not in edge[:6]:
                continue
            emu.he_canon_field_op(cid, op, _p(aa), _p(bb), _p(out))
            assert M.unlimbs(out) == C.field_op(name, a, b), (C.name, name, hex(a), hex(b))


def test_mul_reduction_extremes(emu, curve):
    """products whose high half drives every fold of the 512-bit reduction to its limit"""
    C, cid = curve
    out = np.zeros(4, dtype=np.uint64)
    P = C.P
    vals = [P - 1, P - 2, P - 3, 2**255, 2**256 - 2**33, P - 2**32, P - 977, P - 2**96, P - 2**224, 2**224 - 1,
            0xFFFFFFFF00000000FFFFFFFF00000000FFFFFFFF00000000FFFFFFFF00000000, 2**256 - 2**224 - 1]
    vals = [v % P for v in vals]
    for a in vals:
        for b in vals:
            emu.he_canon_field_op(cid, 2, _p(_arr(a)), _p(_arr(b)), _p(out))
            assert M.unlimbs(out) == a * b % P, (C.name, hex(a), hex(b))


def _jac(C, pt, z):
    """Jacobian limbs of affine pt scaled by z (None -> infinity with arbitrary X, Y)."""
    if pt is M.INF:
        return np.array(M.limbs(5) + M.limbs(7) + M.limbs(0), dtype=np.uint64)
    x, y = pt
    return np.array(M.limbs(x * z * z % C.P) + M.limbs(y * z * z * z % C.P) + M.limbs(z % C.P), dtype=np.uint64)


def _pt_of(out, inf):
    return M.INF if inf else (M.unlimbs(out[:4]), M.unlimbs(out[4:]))


def test_point_ops_with_exceptional_cases(emu, curve):
    C, cid = curve
    rng = random.Random(7)
    out = np.zeros(8, dtype=np.uint64)
    pts = [C.mul(rng.randrange(1, C.N), C.G) for _ in range(12)]
    for p1 in pts[:6]:
        z1 = rng.randrange(1, C.P)
        inf = emu.he_canon_point_op(cid, 0, _p(_jac(C, p1, z1)), None, _p(out))
        assert _pt_of(out, inf) == C.add(p1, p1)
        cases = [rng.choice(pts), p1, C.neg(p1), M.INF]
        for p2 in cases:
            z2 = rng.randrange(1, C.P)
            for op in (1, 3):  # general add, windowed add
                if op == 3 and p2 is M.INF:
                    continue  # table entries are never infinite
                inf = emu.he_canon_point_op(cid, op, _p(_jac(C, p1, z1)), _p(_jac(C, p2, z2)), _p(out))
                assert _pt_of(out, inf) == C.add(p1, p2), (op, p1, p2)
                inf = emu.he_canon_point_op(cid, op, _p(_jac(C, M.INF, 1)), _p(_jac(C, p1, z2)), _p(out))
                assert _pt_of(out, inf) == p1
            if p2 is not M.INF:  # mixed add: q affine (z = 1)
                inf = emu.he_canon_point_op(cid, 2, _p(_jac(C, p1, z1)), _p(_jac(C, p2, 1)), _p(out))
                assert _pt_of(out, inf) == C.add(p1, p2)
                inf = emu.he_canon_point_op(cid, 2, _p(_jac(C, M.INF, 1)), _p(_jac(C, p2, 1)), _p(out))
                assert _pt_of(out, inf) == p2
    # doubling infinity stays infinity
    assert emu.he_canon_point_op(cid, 0, _p(_jac(C, M.INF, 1)), None, _p(out)) == 1


def test_comb_table_entries(emu, curve):
    C, cid = curve
    emu.he_canon_comb_table.restype = ctypes.POINTER(ctypes.c_uint32)
    tab = np.ctypeslib.as_array(emu.he_canon_comb_table(cid), shape=(64 * 15 * 17,)).reshape(64 * 15, 17)
    base = C.G
    for i in range(64):
        for j in (1, 2, 3, 15):
            e = tab[i * 15 + j - 1]
            x = sum(int(e[w]) << (32 * w) for w in range(8))
            y = sum(int(e[8 + w]) << (32 * w) for w in range(8))
            assert (x, y) == C.mul(j, base), (i, j)
        base = C.mul(16, base)


def scalars_of(C):
    return [0, 1, 2, 3, 15, 16, 17, 2**32 - 1, 2**64, 2**128 - 1, 2**255, C.N - 1, C.N, C.N + 1, C.N + 16,
            2**256 - 1, 2 * C.N % 2**256, 0x1111111111111111111111111111111111111111111111111111111111111111,
            0xF0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0,
            0xC9AFA9D845BA75166B5C215767B1D6934E50C3DB36E89B127B8A622B120F6721]


def test_mul_base_comb(emu, curve):
    C, cid = curve
    rng = random.Random(99)
    out = np.zeros(8, dtype=np.uint64)
    for k in scalars_of(C) + [rng.randrange(2**256) for _ in range(150)]:
        st = emu.he_canon_mul_base(cid, _p(_arr(k)), _p(out))
        want = C.mul(k % C.N, C.G)
        assert _pt_of(out, st == 1) == want, hex(k)
        if st == 1:
            assert not out.any()
    # published vectors straight through the device code
    for k, pt in C.KNOWN_MULTIPLES.items():
        emu.he_canon_mul_base(cid, _p(_arr(k)), _p(out))
        assert _pt_of(out, False) == pt


def test_mul_window_variable_base(emu, curve):
    C, cid = curve
    rng = random.Random(5)
    out = np.zeros(8, dtype=np.uint64)
    S = scalars_of(C)
    for t in range(60):
        pt = C.mul(rng.randrange(1, C.N), C.G)
        k = S[t % len(S)] if t < 25 else rng.randrange(2**256)
        pin = np.array(M.xy_limbs(pt), dtype=np.uint64)
        st = emu.he_canon_mul(cid, _p(_arr(k)), _p(pin), _p(out))
        want = C.mul(k % C.N, pt)
        assert st in (0, 1)
        assert _pt_of(out, st == 1) == want, (hex(k), pt)
    # ECDH symmetry: a*(b*G) == b*(a*G)
    a, b = rng.randrange(1, C.N), rng.randrange(1, C.N)
    A, Bp = np.zeros(8, dtype=np.uint64), np.zeros(8, dtype=np.uint64)
    emu.he_canon_mul_base(cid, _p(_arr(a)), _p(A))
    emu.he_canon_mul_base(cid, _p(_arr(b)), _p(Bp))
    s1, s2 = np.zeros(8, dtype=np.uint64), np.zeros(8, dtype=np.uint64)
    emu.he_canon_mul(cid, _p(_arr(a)), _p(Bp), _p(s1))
    emu.he_canon_mul(cid, _p(_arr(b)), _p(A), _p(s2))
    assert np.array_equal(s1, s2) and s1.any()


def test_bad_points_are_rejected(emu, curve):
    C, cid = curve
    out = np.ones(8, dtype=np.uint64)
    x, y = C.G
    bad = [(x, (y + 1) % C.P), (0, 0), (C.P, 0), (x, C.P), (C.P, y), (2**256 - 1, 2**256 - 1)]
    if x + C.P < 2**256:
        bad.append((x + C.P, y))  # the same residue, not canonical
    for b in bad:
        pin = np.array(M.limbs(b[0]) + M.limbs(b[1]), dtype=np.uint64)
        st = emu.he_canon_mul(cid, _p(_arr(5)), _p(pin), _p(out))
        assert st == 2 and not out.any(), b


@pytest.mark.parametrize("n", [1, 7, 8, 9, 64, 515, 1100])
def test_batched_normalisation(emu, curve, n):
    """Montgomery-trick Jacobian -> affine over ragged group sizes, with infinities and rejected inputs."""
    C, cid = curve
    rng = random.Random(n)
    base = [C.mul(rng.randrange(1, C.N), C.G) for _ in range(6)]
    xy = np.zeros((n, 16), dtype=np.uint32)
    zb = np.zeros((n, 8), dtype=np.uint32)
    st = np.zeros(n, dtype=np.uint8)
    want = []
    for i in range(n):
        kind = rng.random()
        pt = base[i % 6]
        z = rng.randrange(1, C.P)
        if kind < 0.1:      # infinity: Z = 0, X and Y arbitrary
            X, Y, z, w = rng.randrange(C.P), rng.randrange(C.P), 0, (M.INF, 1)
        elif kind < 0.2:    # rejected input: status preset, contents arbitrary
            X, Y, z, w = rng.randrange(C.P), rng.randrange(C.P), rng.randrange(C.P), (M.INF, 2)
            st[i] = 2
        else:
            X, Y, w = pt[0] * z * z % C.P, pt[1] * z * z * z % C.P, (pt, 0)
        for wd in range(8):
            xy[i, wd] = (X >> (32 * wd)) & 0xFFFFFFFF
            xy[i, 8 + wd] = (Y >> (32 * wd)) & 0xFFFFFFFF
            zb[i, wd] = (z >> (32 * wd)) & 0xFFFFFFFF
        want.append(w)
    emu.he_canon_normalize(cid, _p(xy), _p(zb), _p(st), ctypes.c_size_t(n))
    for i in range(n):
        pt, status = want[i]
        assert st[i] == status, i
        x = sum(int(xy[i, w]) << (32 * w) for w in range(8))
        y = sum(int(xy[i, 8 + w]) << (32 * w) for w in range(8))
        assert (x, y) == ((0, 0) if pt is M.INF else pt), i


# ---------------------------------------------------------------------------------------------
# Ed25519 (curve id 2): extended coordinates, signed comb, signed windows
# ---------------------------------------------------------------------------------------------
E = M.ED25519


def _ext(pt, z):
    x, y = pt
    P = E.P
    return np.array(M.limbs(x * z % P) + M.limbs(y * z % P) + M.limbs(z % P) + M.limbs(x * y % P * z % P), dtype=np.uint64)


def test_ed25519_model_is_pinned_by_rfc8032():
    assert E.on_curve(E.G) and E.mul(E.N, E.G) == E.IDENTITY
    for seed, pk in (M.ED25519_RFC8032_TEST1, M.ED25519_RFC8032_TEST2):
        assert E.encode(E.mul(E.secret_scalar(seed), E.G)) == pk


def test_ed25519_field_ops(emu):
    rng = random.Random(0xED)
    P = E.P
    edge = sorted({v % P for v in [0, 1, 2, 18, 19, 20, 37, 38, 39, 2**32, 2**64 - 1, 2**128, 2**254, 2**255 - 20,
                                   P - 1, P - 2, P - 19, P - 38, (P - 1) // 2, (P + 1) // 2, 2**255 - 2**224,
                                   2**224 - 1, P - 2**32]})
    pairs = [(a, b) for a in edge for b in edge] + [(rng.randrange(P), rng.randrange(P)) for _ in range(3000)]
    out = np.zeros(4, dtype=np.uint64)
    for a, b in pairs:
        aa, bb = _arr(a), _arr(b)
        for name, op in OPS.items():
            if name == "inv" and rng.random() > 0.03 and a not in edge[:6]:
                continue
            emu.he_canon_field_op(2, op, _p(aa), _p(bb), _p(out))
            assert M.unlimbs(out) == E.field_op(name, a, b), (name, hex(a), hex(b))


def test_ed25519_point_ops(emu):
    """complete formulas: doubling, +- through both Niels forms, with the identity, P + P, P - P,
    and points of small order; T stays consistent (T Z == X Y)"""
    rng = random.Random(11)
    out = np.zeros(8, dtype=np.uint64)
    order4 = None
    # a point of order 4: (sqrt(-1), 0)
    i = pow(2, (E.P - 1) // 4, E.P)
    if E.on_curve((i, 0)):
        order4 = (i, 0)
    special = [E.IDENTITY, (0, E.P - 1)] + ([order4] if order4 else [])
    pts = [E.mul(rng.randrange(1, E.N), E.G) for _ in range(8)] + special
    for p1 in pts:
        z1 = rng.randrange(1, E.P)
        assert emu.he_ced_point_op(0, _p(_ext(p1, z1)), None, _p(out)) == 0
        assert _pt_of(out, False) == E.add(p1, p1)
        for p2 in [rng.choice(pts), p1, E.neg(p1), E.IDENTITY] + special:
            z2 = rng.randrange(1, E.P)
            for op, want in ((1, E.add(p1, p2)), (2, E.add(p1, E.neg(p2))), (3, E.add(p1, p2)), (4, E.add(p1, E.neg(p2)))):
                assert emu.he_ced_point_op(op, _p(_ext(p1, z1)), _p(_ext(p2, z2)), _p(out)) == 0
                assert _pt_of(out, False) == want, (op, p1, p2)


ED_SCALARS = [0, 1, 2, 7, 8, 9, 15, 16, 17, 0x88, 0x89, 0x8888888888888888, 2**32 - 1, 2**64, 2**128 - 1, 2**252, 2**255 - 1,
              2**255, 2**256 - 1, E.N - 1, E.N, E.N + 1, 8 * E.N % 2**256,
              0x8888888888888888888888888888888888888888888888888888888888888888,
              0x9999999999999999999999999999999999999999999999999999999999999999,
              0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFF8,
              0x7777777777777777777777777777777777777777777777777777777777777778]


def test_ed25519_mul_base_signed_comb(emu):
    rng = random.Random(12)
    out = np.zeros(8, dtype=np.uint64)
    for k in ED_SCALARS + [rng.randrange(2**256) for _ in range(150)]:
        emu.he_ced_mul_base(_p(_arr(k)), _p(out))
        assert _pt_of(out, False) == E.mul(k, E.G), hex(k)
    for seed, pk in (M.ED25519_RFC8032_TEST1, M.ED25519_RFC8032_TEST2):   # RFC 8032 key pairs through the device code
        emu.he_ced_mul_base(_p(_arr(E.secret_scalar(seed))), _p(out))
        assert E.encode(_pt_of(out, False)) == pk


def test_ed25519_mul_window_variable_base(emu):
    rng = random.Random(13)
    out = np.zeros(8, dtype=np.uint64)
    for t in range(60):
        pt = E.mul(rng.randrange(1, E.N), E.G)
        k = ED_SCALARS[t % len(ED_SCALARS)] if t < 30 else rng.randrange(2**256)
        pin = np.array(M.limbs(pt[0]) + M.limbs(pt[1]), dtype=np.uint64)
        assert emu.he_ced_mul(_p(_arr(k)), _p(pin), _p(out)) == 0
        assert _pt_of(out, False) == E.mul(k, pt), (hex(k), pt)
    # rejected inputs
    x, y = E.G
    for bad in [(x, (y + 1) % E.P), (E.P, 1), (0, E.P + 1), (x + E.P, y), (2**256 - 1, 2**256 - 1)]:
        pin = np.array(M.limbs(bad[0]) + M.limbs(bad[1]), dtype=np.uint64)
        out[:] = 1
        assert emu.he_ced_mul(_p(_arr(5)), _p(pin), _p(out)) == 2 and not out.any()


@pytest.mark.parametrize("n", [1, 9, 515])
def test_ed25519_batched_normalisation(emu, n):
    rng = random.Random(n)
    xy = np.zeros((n, 16), dtype=np.uint32)
    zb = np.zeros((n, 8), dtype=np.uint32)
    st = np.zeros(n, dtype=np.uint8)
    want = []
    for i in range(n):
        pt = E.mul(rng.randrange(1, 1000), E.G)
        z = rng.randrange(1, E.P)
        X, Y = pt[0] * z % E.P, pt[1] * z % E.P
        for wd in range(8):
            xy[i, wd] = (X >> (32 * wd)) & 0xFFFFFFFF
            xy[i, 8 + wd] = (Y >> (32 * wd)) & 0xFFFFFFFF
            zb[i, wd] = (z >> (32 * wd)) & 0xFFFFFFFF
        want.append(pt)
    emu.he_ced_normalize(_p(xy), _p(zb), _p(st), ctypes.c_size_t(n))
    for i in range(n):
        x = sum(int(xy[i, w]) << (32 * w) for w in range(8))
        y = sum(int(xy[i, 8 + w]) << (32 * w) for w in range(8))
        assert (x, y) == want[i] and st[i] == 0


def test_standard_signature_vectors_verify_in_the_model():
    """RFC 6979 A.2.5 ECDSA (P-256), BIP-340 vector 0, RFC 8032 test 1 -- the last two are the signatures
    the reference's test_standard_vectors.rs quotes -- reduced to u1*G + u2*P and checked with big integers."""
    for name, vec in M.SIGNATURE_VECTORS.items():
        C, u1, u2, P, check = vec()
        R = C.add(C.mul(u1, C.G), C.mul(u2, P))
        assert check(R), name
        assert not check(C.add(C.mul(u1 + 1, C.G), C.mul(u2, P))), name


def test_scalar_field_and_ecdsa_scalars(emu, curve):
    """Montgomery arithmetic modulo the group order and the scalar half of ECDSA verification"""
    C, cid = curve
    n, R = C.N, 2**256
    rng = random.Random(77)
    out = np.zeros(4, dtype=np.uint64)
    Rinv = pow(R, -1, n)
    edge = [0, 1, 2, n - 1, n - 2, 2**255, 2**256 - 1, n, n + 1, (n - 1) // 2]
    for a, b in [(x, y) for x in edge for y in edge] + [(rng.randrange(R), rng.randrange(n)) for _ in range(500)]:
        if a * b >= n * R:
            continue  # REDC's contract: the product is below n * 2^256
        emu.he_canon_scalar_op(cid, 0, _p(_arr(a)), _p(_arr(b)), _p(out))
        assert M.unlimbs(out) == a * b * Rinv % n, (hex(a), hex(b))
    for a in [1, 2, n - 1, n - 2] + [rng.randrange(1, n) for _ in range(12)]:
        emu.he_canon_scalar_op(cid, 1, _p(_arr(a)), None, _p(out))
        assert M.unlimbs(out) == pow(a, -1, n) * R % n
    u1, u2 = np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
    for _ in range(12):
        z, r, s = rng.randrange(R), rng.randrange(1, n), rng.randrange(1, n)
        assert emu.he_canon_ecdsa_scalars(cid, _p(_arr(z)), _p(_arr(r)), _p(_arr(s)), _p(u1), _p(u2)) == 1
        w = pow(s, -1, n)
        assert M.unlimbs(u1) == z * w % n and M.unlimbs(u2) == r * w % n
    for r, s in [(0, 5), (5, 0), (n, 5), (5, n), (2**256 - 1, 5), (5, n + 3)]:
        assert emu.he_canon_ecdsa_scalars(cid, _p(_arr(7)), _p(_arr(r)), _p(_arr(s)), _p(u1), _p(u2)) == 0
    for x, r, want in [(5, 5, 1), (n + 5, 5, 1), (n - 1, n - 1, 1), (n, 0, 1), (5, 6, 0), (C.P - 1, (C.P - 1) % n, 1)]:
        if x < C.P:
            assert emu.he_canon_ecdsa_x_matches(cid, _p(_arr(x)), _p(_arr(r))) == want


@pytest.mark.parametrize("n", [1, 8, 9, 700])
def test_ecdsa_scalars_grouped_inversion(emu, curve, n):
    """one inversion per 8 signatures per lane (Montgomery's trick on s), with out-of-range r / s mixed in"""
    C, cid = curve
    rng = random.Random(1000 + n)
    N = C.N
    Z, Rr, S = [], [], []
    for i in range(n):
        z, r, s = rng.randrange(2**256), rng.randrange(1, N), rng.randrange(1, N)
        k = rng.random()
        if k < 0.08:
            s = rng.choice([0, N, N + 7, 2**256 - 1])
        elif k < 0.16:
            r = rng.choice([0, N, 2**256 - 1])
        Z.append(z); Rr.append(r); S.append(s)
    def w32(vals):
        return np.array([[(v >> (32 * k)) & 0xFFFFFFFF for k in range(8)] for v in vals], dtype=np.uint32)
    z, r, s = w32(Z), w32(Rr), w32(S)
    u1, u2 = np.zeros((n, 8), dtype=np.uint32), np.zeros((n, 8), dtype=np.uint32)
    ok = np.zeros(n, dtype=np.uint8)
    emu.he_canon_ecdsa_scalars_batch(cid, _p(z), _p(r), _p(s), _p(u1), _p(u2), _p(ok), ctypes.c_size_t(n))
    for i in range(n):
        good = 1 <= Rr[i] < N and 1 <= S[i] < N
        assert ok[i] == (1 if good else 0), i
        if good:
            w = pow(S[i], -1, N)
            assert sum(int(u1[i, k]) << (32 * k) for k in range(8)) == Z[i] * w % N
            assert sum(int(u2[i, k]) << (32 * k) for k in range(8)) == Rr[i] * w % N


# ---------------------------------------------------------------------------------------------
# GLV (secp256k1)
# ---------------------------------------------------------------------------------------------
GLV_LAMBDA = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72
GLV_BETA = 0x7AE96A2B657C07106E64479EAC3434E99CF0497512F58995C1396C28719501EE


def test_glv_constants_and_decomposition(emu):
    S = M.SECP256K1
    n = S.N
    assert pow(GLV_LAMBDA, 3, n) == 1 and pow(GLV_BETA, 3, S.P) == 1
    assert S.mul(GLV_LAMBDA, S.G) == (GLV_BETA * S.G[0] % S.P, S.G[1])     # phi(G) = lambda G
    rng = random.Random(55)
    k1, k2 = np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
    ks = [0, 1, 2, n - 1, n - 2, n, n + 1, 2**256 - 1, GLV_LAMBDA, n - GLV_LAMBDA, 2**128, 2**128 - 1, 2**255,
          (n - 1) // 2, (n + 1) // 2] + [rng.randrange(2**256) for _ in range(3000)]
    for k in ks:
        signs = emu.he_glv_decompose(_p(_arr(k)), _p(k1), _p(k2))
        a, b = M.unlimbs(k1), M.unlimbs(k2)
        assert a < 2**129 and b < 2**129, hex(k)
        sa = -a if signs & 1 else a
        sb = -b if signs & 2 else b
        assert (sa + sb * GLV_LAMBDA - k) % n == 0, hex(k)


def test_glv_ladder_matches_the_model(emu):
    S = M.SECP256K1
    rng = random.Random(56)
    out = np.zeros(8, dtype=np.uint64)
    ks = scalars_of(S) + [GLV_LAMBDA, S.N - GLV_LAMBDA, GLV_LAMBDA + 1] + [rng.randrange(2**256) for _ in range(60)]
    for i, k in enumerate(ks):
        pt = S.mul(rng.randrange(1, S.N), S.G) if i % 3 else S.G
        pin = np.array(M.xy_limbs(pt), dtype=np.uint64)
        st = emu.he_glv_mul(_p(_arr(k)), _p(pin), _p(out))
        assert _pt_of(out, st == 1) == S.mul(k % S.N, pt), hex(k)


def test_mul_base_comb8(emu, curve):
    """the 8-bit comb (32 windows x 255 entries): zero bytes, 0xFF bytes, the usual edge scalars"""
    C, cid = curve
    rng = random.Random(98)
    out = np.zeros(8, dtype=np.uint64)
    extra = [0xFF, 0x100, 0xFF00, 0x01000000000000000000000000000000000000000000000000000000000000FF,
             int.from_bytes(bytes([0xFF, 0] * 16), "big"), int.from_bytes(bytes([0, 0xFF] * 16), "big"),
             int.from_bytes(bytes(range(1, 33)), "big")]
    for k in scalars_of(C) + extra + [rng.randrange(2**256) for _ in range(60)]:
        st = emu.he_canon_mul_base8(cid, _p(_arr(k)), _p(out))
        assert _pt_of(out, st == 1) == C.mul(k % C.N, C.G), hex(k)
    for k, pt in C.KNOWN_MULTIPLES.items():
        emu.he_canon_mul_base8(cid, _p(_arr(k)), _p(out))
        assert _pt_of(out, False) == pt


def test_ed25519_mul_base_signed_comb8(emu):
    rng = random.Random(14)
    out = np.zeros(8, dtype=np.uint64)
    extra = [0x80, 0x81, 0x7F, 0x8080, 0x807F, 0xFF, 0x100, int.from_bytes(bytes([0x80] * 32), "big"),
             int.from_bytes(bytes([0x81] * 32), "big"), int.from_bytes(bytes([0xFF, 0x7F] * 16), "big"),
             int.from_bytes(bytes([0x7F, 0x80] * 16), "big")]
    for k in ED_SCALARS + extra + [rng.randrange(2**256) for _ in range(100)]:
        emu.he_ced_mul_base8(_p(_arr(k)), _p(out))
        assert _pt_of(out, False) == E.mul(k, E.G), hex(k)
    for seed, pk in (M.ED25519_RFC8032_TEST1, M.ED25519_RFC8032_TEST2):
        emu.he_ced_mul_base8(_p(_arr(E.secret_scalar(seed))), _p(out))
        assert E.encode(_pt_of(out, False)) == pk


def test_bip340_prepare(emu):
    """lift_x, range checks and u2 = n - e for BIP-340 verification"""
    S = M.SECP256K1
    rng = random.Random(340)
    pxy, u2 = np.zeros(8, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
    n_lift = 0
    xs = [S.G[0], 0, 1, 2, 3, S.P - 1, S.P, S.P + 1, 2**256 - 1] + [rng.randrange(S.P) for _ in range(40)]
    for x in xs:
        e = rng.choice([0, 1, S.N - 1, S.N, S.N + 5, rng.randrange(2**256)])
        ok = emu.he_bip340_prepare(_p(_arr(x)), _p(_arr(5)), _p(_arr(7)), _p(_arr(e)), _p(pxy), _p(u2))
        c = (pow(x, 3, S.P) + 7) % S.P
        y = pow(c, (S.P + 1) // 4, S.P)
        liftable = x < S.P and y * y % S.P == c
        assert ok == (1 if liftable else 0), hex(x)
        assert M.unlimbs(u2) == (-e) % S.N
        if liftable:
            n_lift += 1
            y = y if y % 2 == 0 else S.P - y
            assert (M.unlimbs(pxy[:4]), M.unlimbs(pxy[4:])) == (x, y) and S.on_curve((x, y))
    assert n_lift > 10
    g = _arr(S.G[0])
    assert emu.he_bip340_prepare(_p(g), _p(_arr(S.P)), _p(_arr(7)), _p(_arr(1)), _p(pxy), _p(u2)) == 0      # r >= p
    assert emu.he_bip340_prepare(_p(g), _p(_arr(5)), _p(_arr(S.N)), _p(_arr(1)), _p(pxy), _p(u2)) == 0      # s >= n
    assert emu.he_bip340_prepare(_p(g), _p(_arr(S.P - 1)), _p(_arr(S.N - 1)), _p(_arr(1)), _p(pxy), _p(u2)) == 1


def test_ed25519_decode_and_eddsa_prepare(emu):
    rng = random.Random(8032)
    xy = np.zeros(8, dtype=np.uint64)
    # every encodable point decodes to itself; sign bit selects x
    pts = [E.G, E.IDENTITY, (0, E.P - 1)] + [E.mul(rng.randrange(1, E.N), E.G) for _ in range(25)]
    i = pow(2, (E.P - 1) // 4, E.P)
    pts += [(i, 0), (E.P - i, 0)]           # order 4: exercises the sqrt(-1) branch? (y = 0 -> u = -1)
    for pt in pts:
        enc = int.from_bytes(E.encode(pt), "little")
        assert emu.he_ed_decode(_p(_arr(enc)), _p(xy)) == 1
        assert (M.unlimbs(xy[:4]), M.unlimbs(xy[4:])) == pt
    # rejected encodings: y >= p; x = 0 with the sign bit set; y with no x
    bad = [E.P, E.P + 1, 2**255 - 1, (1 << 255) | 1, (1 << 255) | (E.P - 1)]
    y = 2
    while True:
        u, v = (y * y - 1) % E.P, (E.D * y * y + 1) % E.P
        x2 = u * pow(v, -1, E.P) % E.P
        if pow(x2, (E.P - 1) // 2, E.P) == E.P - 1:
            bad.append(y)
            break
        y += 1
    for enc in bad:
        assert emu.he_ed_decode(_p(_arr(enc)), _p(xy)) == 0, hex(enc)
    # random 256-bit strings: decode result must agree with the model's decoder
    for _ in range(60):
        enc = rng.randrange(2**256)
        try:
            want = M.ed25519_decode(enc.to_bytes(32, "little"))
            sign, yy = enc >> 255, enc & (2**255 - 1)
            if want[0] == 0 and sign:
                want = None
        except AssertionError:
            want = None
        ok = emu.he_ed_decode(_p(_arr(enc)), _p(xy))
        assert ok == (1 if want else 0), hex(enc)
        if want:
            assert (M.unlimbs(xy[:4]), M.unlimbs(xy[4:])) == want
    # prepare: S < l, h < l, u2 = l - h
    axy, rxy, u2 = np.zeros(8, dtype=np.uint64), np.zeros(8, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
    a_enc = int.from_bytes(E.encode(pts[3]), "little")
    r_enc = int.from_bytes(E.encode(pts[4]), "little")
    for s, h, want in [(5, 7, 1), (E.N - 1, E.N - 1, 1), (E.N, 7, 0), (5, E.N, 0), (0, 0, 1), (2**256 - 1, 1, 0)]:
        ok = emu.he_eddsa_prepare(_p(_arr(a_enc)), _p(_arr(r_enc)), _p(_arr(s)), _p(_arr(h)), _p(axy), _p(rxy), _p(u2))
        assert ok == want, (s, h)
        if want:
            assert M.unlimbs(u2) == (-h) % E.N
            assert (M.unlimbs(axy[:4]), M.unlimbs(axy[4:])) == pts[3] and (M.unlimbs(rxy[:4]), M.unlimbs(rxy[4:])) == pts[4]


@pytest.mark.parametrize("name", ["secp256k1", "p256", "ed25519"])
def test_general_scalar_ops(emu, name):
    """a*b+c and a^-1 modulo the group order for ANY 256-bit inputs (the signing-side scalar arithmetic)"""
    cid = {"secp256k1": 0, "p256": 1, "ed25519": 2}[name]
    n = E.N if name == "ed25519" else M.CURVES[name].N
    rng = random.Random(4242)
    out = np.zeros(4, dtype=np.uint64)
    edge = [0, 1, 2, n - 1, n, n + 1, 2 * n % 2**256, 2**255, 2**256 - 1, 2**252, 15 * n if 15 * n < 2**256 else n - 2]
    cases = [(a, b, c) for a in edge for b in edge[:6] for c in (0, 1, n - 1, 2**256 - 1)]
    cases += [(rng.randrange(2**256), rng.randrange(2**256), rng.randrange(2**256)) for _ in range(400)]
    for a, b, c in cases:
        emu.he_canon_scalar_general(cid, 0, _p(_arr(a)), _p(_arr(b)), _p(_arr(c)), _p(out))
        assert M.unlimbs(out) == (a * b + c) % n, (hex(a), hex(b), hex(c))
    for a in edge + [rng.randrange(2**256) for _ in range(10)]:
        emu.he_canon_scalar_general(cid, 1, _p(_arr(a)), None, None, _p(out))
        want = pow(a % n, -1, n) if a % n else 0
        assert M.unlimbs(out) == want, hex(a)
