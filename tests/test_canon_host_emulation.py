"""
Canonical-math mode (NOT reference parity): the device header canon_secp256k1.hpp compiled for the
host (tools/host_emul.cpp) against the big-integer model oracle/canon_model.py, which is itself
pinned by the published multiples of G.  No GPU needed; the GPU build of the same header is
checked in tests/test_gpu_canon.py.
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

EDGE = [0, 1, 2, 3, 977, 2**32, 2**32 + 977, 2**32 + 976, 2**64 - 1, 2**128, 2**255, 2**255 + 19,
        M.P - 1, M.P - 2, M.P - 977, M.P - 2**32, (M.P - 1) // 2, (M.P + 1) // 2, 2**256 - 2**33,
        M.P - 2**224, 0xFFFFFFFF * (2**224), 2**224 - 1]
EDGE = sorted({v % M.P for v in EDGE})


@pytest.fixture(scope="module")
def emu():
    if not os.path.exists(CLANG):
        pytest.skip("ROCm clang++ not available")
    src = os.path.join(ROOT, "tools", "host_emul.cpp")
    deps = [src] + [os.path.join(ROOT, "forge_ec_amd", "csrc", f) for f in
                    ("limbs.hpp", "secp256k1.hpp", "p256.hpp", "ed25519.hpp", "canon_secp256k1.hpp")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.check_call([CLANG, "-O2", "-std=c++17", "-fPIC", "-shared", "-o", SO, src])
    return ctypes.CDLL(SO)


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


def test_model_is_pinned_by_published_multiples():
    assert M.on_curve(M.G)
    for k, pt in M.KNOWN_MULTIPLES.items():
        assert M.mul(k, M.G) == pt
    assert M.mul(M.N, M.G) is M.INF
    assert M.mul(M.N - 1, M.G) == M.neg(M.G)


def test_field_ops(emu):
    rng = random.Random(0xC0FFEE)
    pairs = [(a, b) for a in EDGE for b in EDGE] + [(rng.randrange(M.P), rng.randrange(M.P)) for _ in range(3000)]
    out = np.zeros(4, dtype=np.uint64)
    for a, b in pairs:
        aa, bb = _arr(a), _arr(b)
        for name, op in OPS.items():
            if name == "inv" and rng.random() > 0.05 and a not in EDGE[:6]:
                continue
            emu.he_canon_field_op(op, _p(aa), _p(bb), _p(out))
            assert M.unlimbs(out) == M.field_op(name, a, b), (name, hex(a), hex(b))


def test_mul_reduction_extremes(emu):
    # products whose high half maximises every fold of reduce512
    out = np.zeros(4, dtype=np.uint64)
    vals = [M.P - 1, M.P - 2, 2**256 - 2**32 - 978, 2**255, 2**256 - 2**33, M.P - 2**32, M.P - 977]
    for a in vals:
        for b in vals:
            a %= M.P
            b %= M.P
            emu.he_canon_field_op(2, _p(_arr(a)), _p(_arr(b)), _p(out))
            assert M.unlimbs(out) == a * b % M.P


def _jac(pt, z):
    """Jacobian limbs of affine pt scaled by z (None -> infinity with arbitrary X, Y)."""
    if pt is M.INF:
        return np.array(M.limbs(5) + M.limbs(7) + M.limbs(0), dtype=np.uint64)
    x, y = pt
    return np.array(M.limbs(x * z * z % M.P) + M.limbs(y * z * z * z % M.P) + M.limbs(z % M.P), dtype=np.uint64)


def _pt_of(out, inf):
    return M.INF if inf else (M.unlimbs(out[:4]), M.unlimbs(out[4:]))


def test_point_ops_with_exceptional_cases(emu):
    rng = random.Random(7)
    out = np.zeros(8, dtype=np.uint64)
    pts = [M.mul(rng.randrange(1, M.N), M.G) for _ in range(12)]
    for p1 in pts[:6]:
        z1 = rng.randrange(1, M.P)
        inf = emu.he_canon_point_op(0, _p(_jac(p1, z1)), None, _p(out))
        assert _pt_of(out, inf) == M.add(p1, p1)
        cases = [rng.choice(pts), p1, M.neg(p1), M.INF]
        for p2 in cases:
            z2 = rng.randrange(1, M.P)
            for op in (1, 3):  # general add, windowed add
                if op == 3 and p2 is M.INF:
                    continue  # table entries are never infinite
                inf = emu.he_canon_point_op(op, _p(_jac(p1, z1)), _p(_jac(p2, z2)), _p(out))
                assert _pt_of(out, inf) == M.add(p1, p2), (op, p1, p2)
                inf = emu.he_canon_point_op(op, _p(_jac(M.INF, 1)), _p(_jac(p1, z2)), _p(out))
                assert _pt_of(out, inf) == p1
            if p2 is not M.INF:  # mixed add: q affine (z = 1)
                inf = emu.he_canon_point_op(2, _p(_jac(p1, z1)), _p(_jac(p2, 1)), _p(out))
                assert _pt_of(out, inf) == M.add(p1, p2)
                inf = emu.he_canon_point_op(2, _p(_jac(M.INF, 1)), _p(_jac(p2, 1)), _p(out))
                assert _pt_of(out, inf) == p2
    # doubling infinity stays infinity
    assert emu.he_canon_point_op(0, _p(_jac(M.INF, 1)), None, _p(out)) == 1


def test_comb_table_entries(emu):
    emu.he_canon_comb_table.restype = ctypes.POINTER(ctypes.c_uint32)
    tab = np.ctypeslib.as_array(emu.he_canon_comb_table(), shape=(64 * 15 * 17,)).reshape(64 * 15, 17)
    base = M.G
    for i in range(64):
        for j in (1, 2, 3, 15):
            e = tab[i * 15 + j - 1]
            x = sum(int(e[w]) << (32 * w) for w in range(8))
            y = sum(int(e[8 + w]) << (32 * w) for w in range(8))
            assert (x, y) == M.mul(j, base), (i, j)
        base = M.mul(16, base)


SCALARS = [0, 1, 2, 3, 15, 16, 17, 2**32 - 1, 2**64, 2**128 - 1, 2**255, M.N - 1, M.N, M.N + 1, M.N + 16,
           2**256 - 1, 2 * M.N % 2**256, 0x1111111111111111111111111111111111111111111111111111111111111111,
           0xF0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0, 0xC9AFA9D845BA75166B5C215767B1D6934E50C3DB36E89B127B8A622B120F6721]


def test_mul_base_comb(emu):
    rng = random.Random(99)
    out = np.zeros(8, dtype=np.uint64)
    for k in SCALARS + [rng.randrange(2**256) for _ in range(150)]:
        st = emu.he_canon_mul_base(_p(_arr(k)), _p(out))
        want = M.mul(k % M.N, M.G)
        assert _pt_of(out, st == 1) == want, hex(k)
        if st == 1:
            assert not out.any()
    # published vectors straight through the device code
    for k, pt in M.KNOWN_MULTIPLES.items():
        emu.he_canon_mul_base(_p(_arr(k)), _p(out))
        assert _pt_of(out, False) == pt


def test_mul_window_variable_base(emu):
    rng = random.Random(5)
    out = np.zeros(8, dtype=np.uint64)
    for t in range(60):
        pt = M.mul(rng.randrange(1, M.N), M.G)
        k = SCALARS[t % len(SCALARS)] if t < 25 else rng.randrange(2**256)
        pin = np.array(M.xy_limbs(pt), dtype=np.uint64)
        st = emu.he_canon_mul(_p(_arr(k)), _p(pin), _p(out))
        want = M.mul(k % M.N, pt)
        assert st in (0, 1)
        assert _pt_of(out, st == 1) == want, (hex(k), pt)
    # ECDH symmetry: a*(b*G) == b*(a*G)
    a, b = rng.randrange(1, M.N), rng.randrange(1, M.N)
    A, Bp = np.zeros(8, dtype=np.uint64), np.zeros(8, dtype=np.uint64)
    emu.he_canon_mul_base(_p(_arr(a)), _p(A))
    emu.he_canon_mul_base(_p(_arr(b)), _p(Bp))
    s1, s2 = np.zeros(8, dtype=np.uint64), np.zeros(8, dtype=np.uint64)
    emu.he_canon_mul(_p(_arr(a)), _p(Bp), _p(s1))
    emu.he_canon_mul(_p(_arr(b)), _p(A), _p(s2))
    assert np.array_equal(s1, s2) and s1.any()


def test_bad_points_are_rejected(emu):
    out = np.ones(8, dtype=np.uint64)
    x, y = M.G
    for bad in [(x, (y + 1) % M.P), (x + M.P, y) if x + M.P < 2**256 else (M.P, y), (0, 0), (M.P, 0), (x, M.P + y) if M.P + y < 2**256 else (x, M.P)]:
        pin = np.array(M.limbs(bad[0]) + M.limbs(bad[1]), dtype=np.uint64)
        st = emu.he_canon_mul(_p(_arr(5)), _p(pin), _p(out))
        assert st == 2 and not out.any(), bad


@pytest.mark.parametrize("n", [1, 7, 8, 9, 64, 515, 1100])
def test_batched_normalisation(emu, n):
    """Montgomery-trick Jacobian -> affine over ragged group sizes, with infinities and rejected inputs."""
    rng = random.Random(n)
    base = [M.mul(rng.randrange(1, M.N), M.G) for _ in range(6)]
    xy = np.zeros((n, 16), dtype=np.uint32)
    zb = np.zeros((n, 8), dtype=np.uint32)
    st = np.zeros(n, dtype=np.uint8)
    want = []
    for i in range(n):
        kind = rng.random()
        pt = base[i % 6]
        z = rng.randrange(1, M.P)
        if kind < 0.1:      # infinity: Z = 0, X and Y arbitrary
            X, Y, z, w = rng.randrange(M.P), rng.randrange(M.P), 0, (M.INF, 1)
        elif kind < 0.2:    # rejected input: status preset, contents arbitrary
            X, Y, z, w = rng.randrange(M.P), rng.randrange(M.P), rng.randrange(M.P), (M.INF, 2)
            st[i] = 2
        else:
            X, Y, w = pt[0] * z * z % M.P, pt[1] * z * z * z % M.P, (pt, 0)
        for wd in range(8):
            xy[i, wd] = (X >> (32 * wd)) & 0xFFFFFFFF
            xy[i, 8 + wd] = (Y >> (32 * wd)) & 0xFFFFFFFF
            zb[i, wd] = (z >> (32 * wd)) & 0xFFFFFFFF
        want.append(w)
    emu.he_canon_normalize(_p(xy), _p(zb), _p(st), ctypes.c_size_t(n))
    for i in range(n):
        pt, status = want[i]
        assert st[i] == status, i
        x = sum(int(xy[i, w]) << (32 * w) for w in range(8))
        y = sum(int(xy[i, 8 + w]) << (32 * w) for w in range(8))
        assert (x, y) == ((0, 0) if pt is M.INF else pt), i
