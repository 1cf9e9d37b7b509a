"""
Canonical-math mode on the GPU (include/fecgpu_canon.h) -- NOT reference parity.

Checked against the big-integer model oracle/canon_model.py (pinned by published multiples of G,
tests/test_canon_host_emulation.py) and through size-independent properties of the real group at
2^20 elements: the fixed-base comb and the variable-base windowed ladder are two independent
algorithms that must agree on k*G, and ECDH must be symmetric.
"""
import random

import numpy as np
import pytest

import vectors as V
from oracle import canon_model as M

pytestmark = pytest.mark.gpu



def scalars_of(C):
    return [0, 1, 2, 3, 15, 16, 17, 2**32 - 1, 2**64, 2**128 - 1, 2**255, C.N - 1, C.N, C.N + 1, C.N + 16,
            2**256 - 1, 2 * C.N % 2**256, 0x1111111111111111111111111111111111111111111111111111111111111111,
            0xF0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0,
            0xC9AFA9D845BA75166B5C215767B1D6934E50C3DB36E89B127B8A622B120F6721,
            # around the GLV eigenvalue of secp256k1 (k2 = +-1, k1 = 0 / small) and the half-order boundary
            0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72,
            0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD73,
            C.N - 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72, (C.N - 1) // 2, (C.N + 1) // 2,
            2**128, 2**128 - 1, 2**129]


@pytest.fixture(scope="module", params=["secp256k1", "p256"])
def canon(request, gpu_ctx):
    """(device wrapper, big-integer model) of one curve"""
    from forge_ec_amd.canon import CANON_CURVES
    return CANON_CURVES[request.param](gpu_ctx), M.CURVES[request.param]


def _arr(vals):
    return np.array([M.limbs(v) for v in vals], dtype=np.uint64)


def _pts(xy, st):
    return [M.INF if st[i] == 1 else (M.unlimbs(xy[i, :4]), M.unlimbs(xy[i, 4:])) for i in range(xy.shape[0])]


def test_field_ops_match_the_model(canon):
    canon, C = canon
    from forge_ec_amd import _lib as L
    rng = random.Random(1)
    edge = sorted({v % C.P for v in [0, 1, 2, 977, 2**32 + 977, 2**255, C.P - 1, C.P - 2, C.P - 977, C.P - 2**32, 2**224 - 1, C.P - 2**224, 2**96, C.P - 2**96, 2**192, 2**256 - C.P]})
    a = [x for x in edge for _ in edge] + [rng.randrange(C.P) for _ in range(4000)]
    b = [y for _ in edge for y in edge] + [rng.randrange(C.P) for _ in range(4000)]
    A, B = _arr(a), _arr(b)
    for name, op in (("add", L.F_ADD), ("sub", L.F_SUB), ("mul", L.F_MUL), ("sqr", L.F_SQR), ("neg", L.F_NEG),
                     ("inv", L.F_INV)):
        out = canon.field_op(op, A, B if name in ("add", "sub", "mul") else None)
        for i in range(len(a)):
            assert M.unlimbs(out[i]) == C.field_op(name, a[i], b[i]), (name, hex(a[i]), hex(b[i]))


def test_published_points(canon):
    canon, C = canon
    ks = sorted(C.KNOWN_MULTIPLES)
    xy, st = canon.mul_base(_arr(ks))
    assert not st.any()
    assert _pts(xy, st) == [C.KNOWN_MULTIPLES[k] for k in ks]
    g = np.array([M.xy_limbs(C.G)] * len(ks), dtype=np.uint64)
    xy2, st2 = canon.mul(_arr(ks), g)
    assert np.array_equal(xy, xy2) and not st2.any()


def test_mul_base_matches_the_model(canon):
    canon, C = canon
    SCALARS = scalars_of(C)
    rng = random.Random(2)
    ks = SCALARS + [rng.randrange(2**256) for _ in range(700)]  # ragged: 720 = 2 workgroups + tail
    xy, st = canon.mul_base(_arr(ks))
    got = _pts(xy, st)
    for i, k in enumerate(ks):
        assert got[i] == C.mul(k % C.N, C.G), hex(k)
        if st[i] == 1:
            assert not xy[i].any()
    assert st[0] == 1 and st[SCALARS.index(C.N)] == 1


def test_mul_variable_base_matches_the_model(canon):
    canon, C = canon
    SCALARS = scalars_of(C)
    rng = random.Random(3)
    ks = SCALARS + [rng.randrange(2**256) for _ in range(280)]
    base = [C.mul(rng.randrange(1, C.N), C.G) for _ in range(20)]
    pts = [base[i % 20] for i in range(len(ks))]
    xy, st = canon.mul(_arr(ks), np.array([M.xy_limbs(p) for p in pts], dtype=np.uint64))
    got = _pts(xy, st)
    for i, k in enumerate(ks):
        assert st[i] in (0, 1)
        assert got[i] == C.mul(k % C.N, pts[i]), (hex(k), pts[i])


def test_bad_points_are_rejected(canon):
    canon, C = canon
    x, y = C.G
    bad = [(x, (y + 1) % C.P), (C.P, y), (0, 0), (x, C.P), (2**256 - 1, 2**256 - 1)]
    good = C.mul(5, C.G)
    pts = np.array([M.limbs(p[0]) + M.limbs(p[1]) for p in bad] + [M.xy_limbs(good)], dtype=np.uint64)
    xy, st = canon.mul(_arr([7] * len(pts)), pts)
    assert list(st[:-1]) == [2] * len(bad) and not xy[:-1].any()
    assert st[-1] == 0 and _pts(xy[-1:], st[-1:])[0] == C.mul(35, C.G)


def test_empty_and_argument_errors(canon):
    canon, C = canon
    import forge_ec_amd as F
    xy, st = canon.mul_base(np.zeros((0, 4), dtype=np.uint64))
    assert xy.shape == (0, 8) and st.shape == (0,)
    with pytest.raises(ValueError):
        canon.mul(np.zeros((2, 4), dtype=np.uint64), np.zeros((3, 8), dtype=np.uint64))
    with pytest.raises(F.FecError):
        canon.field_op(9, np.zeros((1, 4), dtype=np.uint64))


def test_full_size_comb_equals_window_and_ecdh_is_symmetric(canon):
    canon, C = canon
    """2^20 elements, device-resident: k*G by comb == k*G by the windowed ladder on G; a*(b*G) == b*(a*G);
    a seeded sample against the model."""
    import torch
    n = 1 << 20
    a = V.scalars(n, 0, 3001)
    b = V.scalars(n, 0, 3002)
    st_ = torch.cuda.current_stream().cuda_stream
    da = torch.from_numpy(a.view(np.int64)).cuda()
    db = torch.from_numpy(b.view(np.int64)).cuda()
    A = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    B = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    sA = torch.empty(n, dtype=torch.uint8, device="cuda")
    sB = torch.empty(n, dtype=torch.uint8, device="cuda")
    canon.mul_base_dev(da.data_ptr(), A.data_ptr(), sA.data_ptr(), n, st_)
    canon.mul_base_dev(db.data_ptr(), B.data_ptr(), sB.data_ptr(), n, st_)
    torch.cuda.synchronize()
    assert int(sA.sum()) == 0 and int(sB.sum()) == 0
    # comb vs window on the generator
    g = torch.from_numpy(np.tile(np.array(M.xy_limbs(C.G), dtype=np.uint64), (n, 1)).view(np.int64)).cuda()
    A2 = torch.empty_like(A)
    s2 = torch.empty_like(sA)
    canon.mul_dev(da.data_ptr(), g.data_ptr(), A2.data_ptr(), s2.data_ptr(), n, st_)
    torch.cuda.synchronize()
    assert torch.equal(A, A2) and int(s2.sum()) == 0
    # ECDH symmetry
    S1 = torch.empty_like(A)
    S2 = torch.empty_like(A)
    canon.mul_dev(da.data_ptr(), B.data_ptr(), S1.data_ptr(), s2.data_ptr(), n, st_)
    canon.mul_dev(db.data_ptr(), A.data_ptr(), S2.data_ptr(), sB.data_ptr(), n, st_)
    torch.cuda.synchronize()
    assert torch.equal(S1, S2) and int(s2.sum()) == 0 and int(sB.sum()) == 0
    # sample against the model
    rng = np.random.default_rng(5)
    idx = np.unique(np.concatenate([np.arange(4), np.arange(n - 4, n), rng.integers(0, n, size=120)]))
    Ah = A.cpu().numpy().view(np.uint64)
    Sh = S1.cpu().numpy().view(np.uint64)
    for i in idx:
        ka, kb = M.unlimbs(a[i]), M.unlimbs(b[i])
        pa = C.mul(ka % C.N, C.G)
        assert (M.unlimbs(Ah[i, :4]), M.unlimbs(Ah[i, 4:])) == pa
        assert (M.unlimbs(Sh[i, :4]), M.unlimbs(Sh[i, 4:])) == C.mul(kb % C.N, pa)


# ---------------------------------------------------------------------------------------------
# Ed25519
# ---------------------------------------------------------------------------------------------
E = M.ED25519


@pytest.fixture(scope="module")
def ced(gpu_ctx):
    from forge_ec_amd.canon import CanonEd25519
    return CanonEd25519(gpu_ctx)


def _epts(xy):
    return [(M.unlimbs(xy[i, :4]), M.unlimbs(xy[i, 4:])) for i in range(xy.shape[0])]


def test_ed25519_field_ops(ced):
    from forge_ec_amd import _lib as L
    rng = random.Random(21)
    P = E.P
    edge = sorted({v % P for v in [0, 1, 2, 18, 19, 20, 38, 2**254, 2**255 - 20, P - 1, P - 2, P - 19, (P - 1) // 2, 2**224 - 1]})
    a = [x for x in edge for _ in edge] + [rng.randrange(P) for _ in range(3000)]
    b = [y for _ in edge for y in edge] + [rng.randrange(P) for _ in range(3000)]
    A, B = _arr(a), _arr(b)
    for name, op in (("add", L.F_ADD), ("sub", L.F_SUB), ("mul", L.F_MUL), ("sqr", L.F_SQR), ("neg", L.F_NEG),
                     ("inv", L.F_INV)):
        out = ced.field_op(op, A, B if name in ("add", "sub", "mul") else None)
        for i in range(len(a)):
            assert M.unlimbs(out[i]) == E.field_op(name, a[i], b[i]), (name, hex(a[i]), hex(b[i]))


def test_ed25519_rfc8032_key_pairs(ced):
    seeds = [M.ED25519_RFC8032_TEST1, M.ED25519_RFC8032_TEST2]
    xy, st = ced.mul_base(_arr([E.secret_scalar(s) for s, _ in seeds]))
    assert not st.any()
    assert [E.encode(p) for p in _epts(xy)] == [pk for _, pk in seeds]


def test_ed25519_mul_base_and_variable_base_match_the_model(ced):
    rng = random.Random(22)
    special = [0, 1, 2, 7, 8, 9, 16, 0x88, 0x89, 2**252, 2**255 - 1, 2**255, 2**256 - 1, E.N - 1, E.N, E.N + 1,
               0x8888888888888888888888888888888888888888888888888888888888888888,
               0x9999999999999999999999999999999999999999999999999999999999999999,
               0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFF8]
    ks = special + [rng.randrange(2**256) for _ in range(500)]
    xy, st = ced.mul_base(_arr(ks))
    assert not st.any()
    got = _epts(xy)
    for i, k in enumerate(ks):
        assert got[i] == E.mul(k, E.G), hex(k)
    ks = special + [rng.randrange(2**256) for _ in range(150)]
    base = [E.mul(rng.randrange(1, E.N), E.G) for _ in range(10)] + [E.IDENTITY, (0, E.P - 1)]
    pts = [base[i % len(base)] for i in range(len(ks))]
    xy, st = ced.mul(_arr(ks), np.array([M.limbs(p[0]) + M.limbs(p[1]) for p in pts], dtype=np.uint64))
    assert not st.any()
    got = _epts(xy)
    for i, k in enumerate(ks):
        assert got[i] == E.mul(k, pts[i]), (hex(k), pts[i])


def test_ed25519_bad_points(ced):
    x, y = E.G
    bad = [(x, (y + 1) % E.P), (E.P, 1), (0, E.P + 1), (2**256 - 1, 2**256 - 1)]
    pts = np.array([M.limbs(p[0]) + M.limbs(p[1]) for p in bad] + [M.limbs(x) + M.limbs(y)], dtype=np.uint64)
    xy, st = ced.mul(_arr([7] * len(pts)), pts)
    assert list(st) == [2] * len(bad) + [0] and not xy[:-1].any()
    assert _epts(xy[-1:])[0] == E.mul(7, E.G)


def test_ed25519_full_size_comb_equals_window_and_dh_is_symmetric(ced):
    import torch
    n = 1 << 20
    a = V.scalars(n, 2, 3101)
    b = V.scalars(n, 2, 3102)
    st_ = torch.cuda.current_stream().cuda_stream
    da = torch.from_numpy(a.view(np.int64)).cuda()
    db = torch.from_numpy(b.view(np.int64)).cuda()
    A = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    B = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    s1 = torch.empty(n, dtype=torch.uint8, device="cuda")
    s2 = torch.empty(n, dtype=torch.uint8, device="cuda")
    ced.mul_base_dev(da.data_ptr(), A.data_ptr(), s1.data_ptr(), n, st_)
    ced.mul_base_dev(db.data_ptr(), B.data_ptr(), s2.data_ptr(), n, st_)
    g = torch.from_numpy(np.tile(np.array(M.limbs(E.G[0]) + M.limbs(E.G[1]), dtype=np.uint64), (n, 1)).view(np.int64)).cuda()
    A2 = torch.empty_like(A)
    ced.mul_dev(da.data_ptr(), g.data_ptr(), A2.data_ptr(), s1.data_ptr(), n, st_)
    torch.cuda.synchronize()
    assert torch.equal(A, A2) and int(s1.sum()) == 0 and int(s2.sum()) == 0
    S1 = torch.empty_like(A)
    S2 = torch.empty_like(A)
    ced.mul_dev(da.data_ptr(), B.data_ptr(), S1.data_ptr(), s1.data_ptr(), n, st_)
    ced.mul_dev(db.data_ptr(), A.data_ptr(), S2.data_ptr(), s2.data_ptr(), n, st_)
    torch.cuda.synchronize()
    assert torch.equal(S1, S2) and int(s1.sum()) == 0 and int(s2.sum()) == 0
    rng = np.random.default_rng(6)
    idx = np.unique(np.concatenate([np.arange(4), np.arange(n - 4, n), rng.integers(0, n, size=100)]))
    Ah = A.cpu().numpy().view(np.uint64)
    Sh = S1.cpu().numpy().view(np.uint64)
    for i in idx:
        pa = E.mul(M.unlimbs(a[i]), E.G)
        assert (M.unlimbs(Ah[i, :4]), M.unlimbs(Ah[i, 4:])) == pa
        assert (M.unlimbs(Sh[i, :4]), M.unlimbs(Sh[i, 4:])) == E.mul(M.unlimbs(b[i]), pa)


# ---------------------------------------------------------------------------------------------
# u1*G + u2*P (the verification point), all three curves
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["secp256k1", "p256", "ed25519"])
def test_double_mul_matches_the_model(gpu_ctx, name):
    from forge_ec_amd.canon import CANON_CURVES
    dev = CANON_CURVES[name](gpu_ctx)
    C = E if name == "ed25519" else M.CURVES[name]
    rng = random.Random(31)
    q = rng.randrange(2, C.N)
    Q = C.mul(q, C.G)
    inv_q = pow(q, -1, C.N)
    cases = []          # (u1, u2, P)
    for _ in range(280):
        cases.append((rng.randrange(2**256), rng.randrange(2**256), C.mul(rng.randrange(1, C.N), C.G) if rng.random() < 0.1 else Q))
    u = rng.randrange(1, C.N)
    cases += [(0, 0, Q), (0, 5, Q), (5, 0, Q), (1, 1, C.G), (u, C.N - u, C.G),             # ..., G + G, u*G - u*G
              (u * q % C.N, u, Q), (u * q % C.N, C.N - u, Q),                             # equal halves (doubling), opposite halves
              (C.N, C.N, Q), (2**256 - 1, 2**256 - 1, Q), (7, inv_q, Q)]
    u1 = _arr([c[0] for c in cases])
    u2 = _arr([c[1] for c in cases])
    pts = np.array([M.limbs(c[2][0]) + M.limbs(c[2][1]) for c in cases], dtype=np.uint64)
    xy, st = dev.double_mul(u1, u2, pts)
    for i, (a, b, P) in enumerate(cases):
        want = C.add(C.mul(a, C.G), C.mul(b, P)) if name == "ed25519" else C.add(C.mul(a % C.N, C.G), C.mul(b % C.N, P))
        if name != "ed25519" and want is M.INF:
            assert st[i] == 1 and not xy[i].any(), i
        else:
            assert st[i] == 0, i
            assert (M.unlimbs(xy[i, :4]), M.unlimbs(xy[i, 4:])) == want, (i, hex(a), hex(b))
    # a rejected point keeps status 2 through the accumulate pass
    bad = pts.copy()
    bad[3, 4] ^= np.uint64(1)
    xy, st = dev.double_mul(u1, u2, bad)
    assert st[3] == 2 and not xy[3].any() and st[4] == 0


@pytest.mark.parametrize("name", ["secp256k1", "p256", "ed25519"])
def test_standard_signature_vectors_verify_on_the_gpu(gpu_ctx, name):
    """The verification point of a published signature, computed by fec_canon_double_mul: BIP-340 vector 0
    (secp256k1), RFC 6979 A.2.5 ECDSA (P-256), RFC 8032 test 1 (Ed25519).  Scalar preparation (hash, s^-1
    mod n) is done here with Python integers; the GPU evaluates u1*G + u2*P."""
    from forge_ec_amd.canon import CANON_CURVES
    dev = CANON_CURVES[name](gpu_ctx)
    C, u1, u2, P, check = M.SIGNATURE_VECTORS[name]()
    pts = np.array([M.limbs(P[0]) + M.limbs(P[1])] * 2, dtype=np.uint64)
    xy, st = dev.double_mul(_arr([u1, (u1 + 1) % C.N]), _arr([u2, u2]), pts)
    assert list(st) == [0, 0]
    good = (M.unlimbs(xy[0, :4]), M.unlimbs(xy[0, 4:]))
    forged = (M.unlimbs(xy[1, :4]), M.unlimbs(xy[1, 4:]))
    assert check(good) and not check(forged)


# ---------------------------------------------------------------------------------------------
# standard ECDSA verification end to end (everything after the hash on the GPU)
# ---------------------------------------------------------------------------------------------
def _ecdsa_sign(C, d, z, k):
    R = C.mul(k, C.G)
    r = R[0] % C.N
    s = pow(k, -1, C.N) * (z + r * d) % C.N
    return r, s


@pytest.mark.parametrize("name", ["secp256k1", "p256"])
def test_ecdsa_verify_end_to_end(gpu_ctx, name):
    import hashlib
    from forge_ec_amd.canon import CANON_CURVES
    dev = CANON_CURVES[name](gpu_ctx)
    C = M.CURVES[name]
    rng = random.Random(41)
    rows, want = [], []
    keys = [(d, C.mul(d, C.G)) for d in (rng.randrange(1, C.N) for _ in range(12))]
    for i in range(300):
        d, Q = keys[i % len(keys)]
        z = int.from_bytes(hashlib.sha256(b"message %d" % i).digest(), "big")
        while True:
            r, s = _ecdsa_sign(C, d, z, rng.randrange(1, C.N))
            if r and s:
                break
        kind = i % 10
        ok = 1
        if kind == 1:
            z ^= 1 << rng.randrange(256); ok = 0             # another message
        elif kind == 2:
            s = (s + 1) % C.N or 1; ok = 0                    # tampered s
        elif kind == 3:
            r = (r + 1) % C.N or 1; ok = 0                    # tampered r
        elif kind == 4:
            Q = keys[(i + 1) % len(keys)][1]; ok = 0          # another key
        elif kind == 5:
            s = C.N - s                                       # the other valid s (ECDSA malleability)
        elif kind == 6:
            r, ok = [0, C.N, 2**256 - 1][i % 3], 0            # r out of range
        elif kind == 7:
            s, ok = [0, C.N, C.N + 5][i % 3], 0               # s out of range
        elif kind == 8:
            Q, ok = (Q[0], (Q[1] + 1) % C.P), 0               # key not on the curve
        rows.append((z, r, s, Q))
        want.append(ok)
    # the published signature
    if name == "p256":
        d = 0xC9AFA9D845BA75166B5C215767B1D6934E50C3DB36E89B127B8A622B120F6721
        rows.append((int.from_bytes(hashlib.sha256(b"sample").digest(), "big"),
                     0xEFD48B2AACB6A8FD1140DD9CD45E81D69D2C877B56AAF991C34D0EA84EAF3716,
                     0xF7CB1C942D657C41D436C7A1B6E29F65F3E900DBB9AFF4064DC4AB2F843ACDA8, C.KNOWN_MULTIPLES[d]))
        want.append(1)
    got = dev.ecdsa_verify(_arr([t[0] for t in rows]), _arr([t[1] for t in rows]), _arr([t[2] for t in rows]),
                           np.array([M.limbs(t[3][0]) + M.limbs(t[3][1]) for t in rows], dtype=np.uint64))
    assert list(got) == want
    assert dev.ecdsa_verify(np.zeros((0, 4), np.uint64), np.zeros((0, 4), np.uint64), np.zeros((0, 4), np.uint64),
                            np.zeros((0, 8), np.uint64)).shape == (0,)


def test_both_fixed_base_tables_agree(monkeypatch):
    """key generation through the 4-bit LDS comb (FEC_CANON_COMB4=1) equals the default 8-bit L2 comb"""
    import forge_ec_amd as F
    from forge_ec_amd.canon import CANON_CURVES
    k = V.scalars(3000, 0, 4001)
    k[:4] = 0
    k[4, :] = np.uint64(0xFFFFFFFFFFFFFFFF)
    monkeypatch.setenv("FEC_CANON_COMB4", "1")
    with F.Context(0) as ctx4:
        monkeypatch.delenv("FEC_CANON_COMB4")
        with F.Context(0) as ctx8:
            for name in ("secp256k1", "p256", "ed25519"):
                a, sa = CANON_CURVES[name](ctx4).mul_base(k)
                b, sb = CANON_CURVES[name](ctx8).mul_base(k)
                assert np.array_equal(a, b) and np.array_equal(sa, sb) and (name == "ed25519" or sa[:4].all())


# ---------------------------------------------------------------------------------------------
# BIP-340 and EdDSA verification end to end (hashing here, everything else on the GPU)
# ---------------------------------------------------------------------------------------------
def _tagged(tag, msg):
    import hashlib
    t = hashlib.sha256(tag).digest()
    return hashlib.sha256(t + t + msg).digest()


def test_bip340_verify_end_to_end(gpu_ctx):
    from forge_ec_amd.canon import CanonSecp256k1
    dev = CanonSecp256k1(gpu_ctx)
    S = M.SECP256K1
    rng = random.Random(340)
    rows, want = [], []
    for i in range(200):
        d0 = rng.randrange(1, S.N)
        P = S.mul(d0, S.G)
        d = d0 if P[1] % 2 == 0 else S.N - d0
        msg = b"bip340 message %d" % i
        k0 = rng.randrange(1, S.N)
        R = S.mul(k0, S.G)
        k = k0 if R[1] % 2 == 0 else S.N - k0
        pkx, rx = P[0], R[0]
        e = int.from_bytes(_tagged(b"BIP0340/challenge", rx.to_bytes(32, "big") + pkx.to_bytes(32, "big") + msg), "big")
        s = (k + e * d) % S.N
        ok = 1
        kind = i % 8
        if kind == 1:
            e ^= 1 << rng.randrange(255); ok = 0                  # another message
        elif kind == 2:
            s = (s + 1) % S.N; ok = 0
        elif kind == 3:
            rx = S.mul(k0 + 1, S.G)[0]; ok = 0                     # another R
        elif kind == 4:
            pkx = S.mul(d0 + 1, S.G)[0]; ok = 0                    # another key
        elif kind == 5:
            s, ok = S.N + (s % 1000), 0                            # s >= n
        elif kind == 6:
            rx, ok = S.P + (rx % 1000), 0                          # r >= p
        elif kind == 7:
            x = pkx
            while True:                                            # an x with no point on the curve
                x = (x + 1) % S.P
                c = (pow(x, 3, S.P) + 7) % S.P
                if pow(c, (S.P - 1) // 2, S.P) != 1:
                    break
            pkx, ok = x, 0
        rows.append((pkx, rx, s, e))
        want.append(ok)
    # BIP-340 test vector 0 (the signature test_standard_vectors.rs quotes)
    sig = bytes.fromhex("E907831F80848D1069A5371B402410364BDF1C5F8307B0084C55F1CE2DCA8215"
                        "25F66A4A85EA8B71E482A74F382D2CE5EBEEE8FDB2172F477DF4900D310536C0")
    pkx = 0xF9308A019258C31049344F85F89D5229B531C845836F99B08601F113BCE036F9
    e = int.from_bytes(_tagged(b"BIP0340/challenge", sig[:32] + pkx.to_bytes(32, "big") + bytes(32)), "big")
    rows.append((pkx, int.from_bytes(sig[:32], "big"), int.from_bytes(sig[32:], "big"), e))
    want.append(1)
    got = dev.bip340_verify(_arr([t[0] for t in rows]), _arr([t[1] for t in rows]), _arr([t[2] for t in rows]),
                            _arr([t[3] for t in rows]))
    assert list(got) == want


def test_eddsa_verify_end_to_end(gpu_ctx):
    import hashlib
    from forge_ec_amd.canon import CanonEd25519
    dev = CanonEd25519(gpu_ctx)
    L_ = E.N
    rng = random.Random(8032)
    rows, want = [], []

    def sign(seed, msg):
        hh = hashlib.sha512(seed).digest()
        a = E.secret_scalar(seed)
        A = E.encode(E.mul(a, E.G))
        r = int.from_bytes(hashlib.sha512(hh[32:] + msg).digest(), "little") % L_
        Renc = E.encode(E.mul(r, E.G))
        h = int.from_bytes(hashlib.sha512(Renc + A + msg).digest(), "little") % L_
        return A, Renc, (r + h * a) % L_, h

    for i in range(160):
        seed = bytes(rng.randrange(256) for _ in range(32))
        msg = b"eddsa message %d" % i
        A, Renc, S_, h = sign(seed, msg)
        ok = 1
        kind = i % 8
        if kind == 1:
            h = (h + 1) % L_; ok = 0                                # another message
        elif kind == 2:
            S_ = (S_ + 1) % L_; ok = 0
        elif kind == 3:
            A = sign(bytes(32 - len(b"x")) + b"x", msg)[0]; ok = 0  # another key
        elif kind == 4:
            S_, ok = S_ + L_, 0                                     # S >= l (malleability check)
        elif kind == 5:
            Renc = E.encode(E.mul(rng.randrange(1, L_), E.G)); ok = 0
        elif kind == 6:
            Renc, ok = (E.P + 1).to_bytes(32, "little"), 0          # y >= p: undecodable R
        elif kind == 7:
            A = bytes([A[0]]) + A[1:31] + bytes([A[31] ^ 0x80])     # flipped sign bit: another (valid) point
            ok = 0
        rows.append((int.from_bytes(A, "little"), int.from_bytes(Renc, "little"), S_, h))
        want.append(ok)
    # RFC 8032 TEST 1 (the signature test_standard_vectors.rs quotes) and TEST 2
    for (seed, pk), msg, sighex in (
            (M.ED25519_RFC8032_TEST1, b"", "e5564300c360ac729086e2cc806e828a84877f1eb8e5d974d873e06522490155"
                                           "5fb8821590a33bacc61e39701cf9b46bd25bf5f0595bbe24655141438e7a100b"),
            (M.ED25519_RFC8032_TEST2, bytes([0x72]), "92a009a9f0d4cab8720e820b5f642540a2b27b5416503f8fb3762223ebdb69da"
                                                     "085ac1e43e15996e458f3613d0f11d8c387b2eaeb4302aeeb00d291612bb0c00")):
        sig = bytes.fromhex(sighex)
        h = int.from_bytes(hashlib.sha512(sig[:32] + pk + msg).digest(), "little") % L_
        rows.append((int.from_bytes(pk, "little"), int.from_bytes(sig[:32], "little"), int.from_bytes(sig[32:], "little"), h))
        want.append(1)
    got = dev.eddsa_verify(_arr([t[0] for t in rows]), _arr([t[1] for t in rows]), _arr([t[2] for t in rows]),
                           _arr([t[3] for t in rows]))
    assert list(got) == want


def test_host_pointer_calls_are_chunk_invariant(gpu_ctx):
    """the host-pointer entry points stream the batch in chunks of fec_ctx_set_chunk elements; a 100-element
    chunk (ragged last chunk, scratch bounded by one chunk) gives the same bytes as one big chunk"""
    from forge_ec_amd.canon import CANON_CURVES
    n = 1037
    k1, k2, k3 = V.scalars(n, 0, 5001), V.scalars(n, 0, 5002), V.scalars(n, 0, 5003)
    res = {}
    for chunk in (1 << 18, 100):
        gpu_ctx.set_chunk(chunk)
        out = []
        for name in ("secp256k1", "p256", "ed25519"):
            c = CANON_CURVES[name](gpu_ctx)
            pub, st = c.mul_base(k1)
            out += [pub, st, *c.mul(k2, pub), *c.double_mul(k1, k2, pub)]
            if name != "ed25519":
                out.append(c.ecdsa_verify(k1, k2, k3, pub))
            else:
                out.append(c.eddsa_verify(k1, k2, k3, k3))
        out.append(CANON_CURVES["secp256k1"](gpu_ctx).bip340_verify(k1, k2, k3, k1))
        xy = V.field_elements(2 * n, 1, 5004).reshape(n, 8)
        out.append(gpu_ctx.batch_compress(1, xy))
        res[chunk] = out
    gpu_ctx.set_chunk(1 << 18)
    for a, b in zip(res[1 << 18], res[100]):
        assert np.array_equal(a, b)


# ---------------------------------------------------------------------------------------------
# signing side: scalar-field arithmetic on the GPU, RFC vectors
# ---------------------------------------------------------------------------------------------
def test_scalar_ops_and_ecdsa_sign(gpu_ctx):
    import hashlib
    from forge_ec_amd.canon import CANON_CURVES
    rng = random.Random(6979)
    for name in ("secp256k1", "p256", "ed25519"):
        dev = CANON_CURVES[name](gpu_ctx)
        n = E.N if name == "ed25519" else M.CURVES[name].N
        vals = [0, 1, n - 1, n, n + 1, 2**256 - 1] + [rng.randrange(2**256) for _ in range(300)]
        a, b, c = vals, vals[::-1], vals[3:] + vals[:3]
        got = dev.scalar_muladd(_arr(a), _arr(b), _arr(c))
        assert [M.unlimbs(g) for g in got] == [(x * y + z) % n for x, y, z in zip(a, b, c)]
        got = dev.scalar_inv(_arr(a))
        assert [M.unlimbs(g) for g in got] == [pow(x % n, -1, n) if x % n else 0 for x in a]
    # RFC 6979 A.2.5 (P-256, SHA-256, "sample"): the deterministic nonce k and the signature it yields
    dev = CANON_CURVES["p256"](gpu_ctx)
    C = M.P256
    d = 0xC9AFA9D845BA75166B5C215767B1D6934E50C3DB36E89B127B8A622B120F6721
    k = 0xA6E3C57DD01ABE90086538398355DD4C3B17AA873382B0F24D6129493D8AAD60
    z = int.from_bytes(hashlib.sha256(b"sample").digest(), "big")
    r, s, ok = dev.ecdsa_sign(_arr([z]), _arr([d]), _arr([k]))
    assert ok[0] == 1
    assert M.unlimbs(r[0]) == 0xEFD48B2AACB6A8FD1140DD9CD45E81D69D2C877B56AAF991C34D0EA84EAF3716
    assert M.unlimbs(s[0]) == 0xF7CB1C942D657C41D436C7A1B6E29F65F3E900DBB9AFF4064DC4AB2F843ACDA8
    # sign -> verify round trip on the GPU, both curves
    for name in ("secp256k1", "p256"):
        dev = CANON_CURVES[name](gpu_ctx)
        C = M.CURVES[name]
        m = 500
        ds = [rng.randrange(1, C.N) for _ in range(m)]
        ks = [rng.randrange(1, C.N) for _ in range(m)]
        zs = [rng.randrange(2**256) for _ in range(m)]
        pub, st = dev.mul_base(_arr(ds))
        r, s, ok = dev.ecdsa_sign(_arr(zs), _arr(ds), _arr(ks))
        assert ok.all() and not st.any()
        assert dev.ecdsa_verify(_arr(zs), r, s, pub).all()
        zs2 = _arr(zs)
        zs2[:, 0] ^= np.uint64(1)
        assert not dev.ecdsa_verify(zs2, r, s, pub).any()


def test_eddsa_sign_rfc8032(gpu_ctx):
    """RFC 8032 TEST 1 and TEST 2 signatures reproduced: R = r B and S = h a + r mod l on the GPU, the two
    SHA-512 hashes here"""
    import hashlib
    from forge_ec_amd.canon import CanonEd25519
    dev = CanonEd25519(gpu_ctx)
    for (seed, pk), msg, sighex in (
            (M.ED25519_RFC8032_TEST1, b"", "e5564300c360ac729086e2cc806e828a84877f1eb8e5d974d873e06522490155"
                                           "5fb8821590a33bacc61e39701cf9b46bd25bf5f0595bbe24655141438e7a100b"),
            (M.ED25519_RFC8032_TEST2, bytes([0x72]), "92a009a9f0d4cab8720e820b5f642540a2b27b5416503f8fb3762223ebdb69da"
                                                     "085ac1e43e15996e458f3613d0f11d8c387b2eaeb4302aeeb00d291612bb0c00")):
        hh = hashlib.sha512(seed).digest()
        a = E.secret_scalar(seed)
        r = int.from_bytes(hashlib.sha512(hh[32:] + msg).digest(), "little")       # 512 bits: reduce on the GPU
        r_lo, r_hi = r & (2**256 - 1), r >> 256
        # r mod l = r_hi * (2^256 mod l) + r_lo: two scalar ops on the device
        two256 = _arr([2**256 % E.N])
        rl = dev.scalar_muladd(_arr([r_hi]), two256, _arr([r_lo]))
        xy, st = dev.mul_base(rl)
        Renc = E.encode((M.unlimbs(xy[0, :4]), M.unlimbs(xy[0, 4:])))
        h = int.from_bytes(hashlib.sha512(Renc + pk + msg).digest(), "little")
        hl = dev.scalar_muladd(_arr([h >> 256]), two256, _arr([h & (2**256 - 1)]))
        S = dev.eddsa_sign_finish(hl, _arr([a]), rl)
        sig = Renc + M.unlimbs(S[0]).to_bytes(32, "little")
        assert sig.hex() == sighex, "r mod l = %x (want %x), R status %s, xy %s" % (
            M.unlimbs(rl[0]), r % E.N, st, [hex(int(v)) for v in xy[0]])


def test_wipe_is_ordered_before_the_next_call(gpu_ctx):
    """fec_ctx_wipe zeroes the ctx's device work areas; it must be complete when it returns.  (It once used
    hipMemset on the NULL stream -- asynchronous to the host, not ordered with the ctx's non-blocking stream -- and
    a call that followed a signing helper intermittently had its Z buffer zeroed between two of its kernels:
    "point at infinity" out of a plain k*B.)"""
    from forge_ec_amd.canon import CanonEd25519, CanonSecp256k1
    for cls in (CanonEd25519, CanonSecp256k1):
        dev = cls(gpu_ctx)
        k = _arr([0x8332edb9e5d3ac8522a07b5a1857169a67e6509ab59af23ccdb039970ffbf8f, 5, 2**200 + 12345])
        want_xy, want_st = dev.mul_base(k)
        assert not want_st.any()
        big = V.scalars(1 << 16, 0, 9100)          # make the work areas large enough for the wipe to take a while
        dev.mul_base(big)
        for _ in range(25):
            gpu_ctx.wipe()
            xy, st = dev.mul_base(k)
            assert np.array_equal(st, want_st) and np.array_equal(xy, want_xy)
