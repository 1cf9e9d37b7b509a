"""
c_oracle.py -- ctypes loader for oracle/libforge_ec_oracle.so (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
All arrays are numpy uint64, little-endian limbs (see forge_ec_oracle.h).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libforge_ec_oracle.so")

SECP256K1, P256, ED25519 = 0, 1, 2
POINT_LIMBS = {SECP256K1: 12, P256: 12, ED25519: 16}


def build(force=False):
    """Compile the oracle with gcc (seconds).  Building the checker is not using it."""
    src = os.path.join(_HERE, "forge_ec_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libforge_ec_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        p = ctypes.c_void_p
        L.fo_field_op.argtypes = [ctypes.c_int, ctypes.c_char_p, p, p, p]
        L.fo_field_op.restype = ctypes.c_int
        for name in ("fo_identity", "fo_generator"):
            getattr(L, name).argtypes = [ctypes.c_int, p]
            getattr(L, name).restype = None
        L.fo_is_identity.argtypes = [ctypes.c_int, p]
        L.fo_is_identity.restype = ctypes.c_int
        L.fo_point_add.argtypes = [ctypes.c_int, p, p, p]
        L.fo_point_add.restype = None
        L.fo_point_double.argtypes = [ctypes.c_int, p, p]
        L.fo_point_double.restype = None
        L.fo_secp256k1_point_double_trait.argtypes = [p, p]
        L.fo_secp256k1_point_double_trait.restype = None
        L.fo_point_negate.argtypes = [ctypes.c_int, p, p]
        L.fo_point_negate.restype = None
        L.fo_to_affine.argtypes = [ctypes.c_int, p, p]
        L.fo_to_affine.restype = ctypes.c_int
        L.fo_multiply.argtypes = [ctypes.c_int, p, p, p]
        L.fo_multiply.restype = None
        L.fo_batch_mul.argtypes = [ctypes.c_int, p, p, p, ctypes.c_size_t, ctypes.c_int]
        L.fo_batch_mul.restype = None
        L.fo_batch_mul_fixed.argtypes = [ctypes.c_int, p, p, p, ctypes.c_size_t, ctypes.c_int]
        L.fo_batch_mul_fixed.restype = None
        L.fo_batch_double_mul.argtypes = [ctypes.c_int, p, p, p, p, ctypes.c_size_t, ctypes.c_int]
        L.fo_batch_double_mul.restype = None
        L.fo_batch_to_affine.argtypes = [ctypes.c_int, p, p, p, ctypes.c_size_t, ctypes.c_int]
        L.fo_batch_to_affine.restype = None
        L.fo_secp256k1_scalar_op.argtypes = [ctypes.c_char_p, p, p, p]
        L.fo_secp256k1_scalar_op.restype = ctypes.c_int
        L.fo_secp256k1_ecdsa_verify.argtypes = [p, p, p, p, ctypes.c_int]
        L.fo_secp256k1_ecdsa_verify.restype = ctypes.c_int
        L.fo_batch_secp256k1_ecdsa_verify.argtypes = [p, p, p, p, p, p, ctypes.c_size_t, ctypes.c_int]
        L.fo_batch_secp256k1_ecdsa_verify.restype = None
        L.fo_p256_scalar_op.argtypes = [ctypes.c_char_p, p, p, p]
        L.fo_p256_scalar_op.restype = ctypes.c_int
        L.fo_p256_ecdsa_verify.argtypes = [p, p, p, p, ctypes.c_int]
        L.fo_p256_ecdsa_verify.restype = ctypes.c_int
        L.fo_batch_p256_ecdsa_verify.argtypes = [p, p, p, p, p, p, ctypes.c_size_t, ctypes.c_int]
        L.fo_batch_p256_ecdsa_verify.restype = None
        L.fo_ecdsa_batch_verify.argtypes = [ctypes.c_int, p, p, p, p, p, p, ctypes.c_size_t, p]
        L.fo_ecdsa_batch_verify.restype = ctypes.c_int
        L.fo_batch_validate_point.argtypes = [ctypes.c_int, p, p, p, ctypes.c_size_t, ctypes.c_int]
        L.fo_batch_validate_point.restype = None
        L.fo_batch_ecdh.argtypes = [ctypes.c_int, p, p, p, p, p, ctypes.c_size_t, ctypes.c_int]
        L.fo_batch_ecdh.restype = None
        L.fo_batch_ed25519_eddsa_verify.argtypes = [p, p, p, p, p, p, p, ctypes.c_size_t, ctypes.c_int]
        L.fo_batch_ed25519_eddsa_verify.restype = None
        L.fo_secp256k1_schnorr_batch_verify.argtypes = [p, p, p, p, p, p, p, ctypes.c_size_t, p, p]
        L.fo_secp256k1_schnorr_batch_verify.restype = ctypes.c_int
        L.fo_p256_schnorr_batch_verify.argtypes = [p, p, p, p, p, p, p, ctypes.c_size_t, p, p]
        L.fo_p256_schnorr_batch_verify.restype = ctypes.c_int
        L.fo_ed25519_schnorr_batch_verify.argtypes = [p, p, p, p, p, p, p, ctypes.c_size_t, p, p, p]
        L.fo_ed25519_schnorr_batch_verify.restype = ctypes.c_int
        L.fo_ed25519_scalar_mul_release.argtypes = [p, p, p]
        L.fo_ed25519_scalar_mul_release.restype = ctypes.c_int
        L.fo_batch_schnorr_verify.argtypes = [ctypes.c_int, p, p, p, p, p, p, p, ctypes.c_size_t, ctypes.c_int]
        L.fo_batch_schnorr_verify.restype = None
        L.fo_batch_compress.argtypes = [ctypes.c_int, p, p, p, ctypes.c_size_t]
        L.fo_batch_compress.restype = None
        for name in ("fo_batch_decompress", "fo_batch_decode_uncompressed"):
            getattr(L, name).argtypes = [ctypes.c_int, p, p, p, p, ctypes.c_size_t]
            getattr(L, name).restype = None
        L.fo_batch_encode_uncompressed.argtypes = [ctypes.c_int, p, p, p, ctypes.c_size_t]
        L.fo_batch_encode_uncompressed.restype = None
        _lib = L
    return _lib


def cpu_model():
    """Model string of the host CPU (bench.py prints it beside the CPU baseline)."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


_native = None


def native_lib():
    """The SAME source built for the machine this runs on (gcc -O3 -march=native), for bench.py's cpu_baseline leg:
    the shipped libforge_ec_oracle.so is built with -march=x86-64-v2 (no mulx / adx) so that it runs on any box the
    snapshot travels to, which handicaps the CPU number printed beside the GPU one.  Built into oracle/_native/
    (git- and gpurun-ignored; the file name carries a hash of this CPU's flags, so a build never runs on another
    machine).  Returns (library, description); falls back to (lib(), "shipped ...") when gcc is missing or fails."""
    global _native
    if _native is not None:
        return _native
    import hashlib
    import shutil
    shipped = (lib(), "shipped build, gcc -O3 -march=x86-64-v2 (no gcc on this box, or the native build failed)")
    try:
        flags = ""
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                flags = line
                break
        tag = hashlib.sha256((cpu_model() + flags).encode()).hexdigest()[:12]
        out_dir = os.path.join(_HERE, "_native")
        so = os.path.join(out_dir, "libforge_ec_oracle_%s.so" % tag)
        src = os.path.join(_HERE, "forge_ec_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            gcc = shutil.which("gcc")
            if not gcc:
                _native = shipped
                return _native
            os.makedirs(out_dir, exist_ok=True)
            subprocess.check_call([gcc, "-O3", "-march=native", "-fPIC", "-std=c11", "-shared", "-o", so, src, "-lpthread"],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        L = ctypes.CDLL(so)
        p = ctypes.c_void_p
        L.fo_batch_mul.argtypes = [ctypes.c_int, p, p, p, ctypes.c_size_t, ctypes.c_int]
        L.fo_batch_mul.restype = None
        L.fo_batch_mul_fixed.argtypes = [ctypes.c_int, p, p, p, ctypes.c_size_t, ctypes.c_int]
        L.fo_batch_mul_fixed.restype = None
        L.fo_batch_double_mul.argtypes = [ctypes.c_int, p, p, p, p, ctypes.c_size_t, ctypes.c_int]
        L.fo_batch_double_mul.restype = None
        _native = (L, "built on this box: gcc -O3 -march=native")
    except Exception:
        _native = shipped
    return _native


def _u64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint64))


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def field_op(curve, op, a, b=None):
    a = _u64(a)
    b = _u64(b if b is not None else [0, 0, 0, 0])
    r = np.zeros(4, dtype=np.uint64)
    rc = lib().fo_field_op(curve, op.encode(), _ptr(a), _ptr(b), _ptr(r))
    if rc != 0:
        raise ValueError("fo_field_op rc=%d" % rc)
    return r


def identity(curve):
    r = np.zeros(POINT_LIMBS[curve], dtype=np.uint64)
    lib().fo_identity(curve, _ptr(r))
    return r


def generator(curve):
    r = np.zeros(POINT_LIMBS[curve], dtype=np.uint64)
    lib().fo_generator(curve, _ptr(r))
    return r


def is_identity(curve, p):
    p = _u64(p)
    return bool(lib().fo_is_identity(curve, _ptr(p)))


def point_add(curve, p, q):
    p, q = _u64(p), _u64(q)
    r = np.zeros(POINT_LIMBS[curve], dtype=np.uint64)
    lib().fo_point_add(curve, _ptr(p), _ptr(q), _ptr(r))
    return r


def point_double(curve, p):
    p = _u64(p)
    r = np.zeros(POINT_LIMBS[curve], dtype=np.uint64)
    lib().fo_point_double(curve, _ptr(p), _ptr(r))
    return r


def secp256k1_point_double_trait(p):
    p = _u64(p)
    r = np.zeros(12, dtype=np.uint64)
    lib().fo_secp256k1_point_double_trait(_ptr(p), _ptr(r))
    return r


def point_negate(curve, p):
    p = _u64(p)
    r = np.zeros(POINT_LIMBS[curve], dtype=np.uint64)
    lib().fo_point_negate(curve, _ptr(p), _ptr(r))
    return r


def to_affine(curve, p):
    p = _u64(p)
    xy = np.zeros(8, dtype=np.uint64)
    inf = lib().fo_to_affine(curve, _ptr(p), _ptr(xy))
    return xy, bool(inf)


def multiply(curve, point, scalar):
    point, scalar = _u64(point), _u64(scalar)
    r = np.zeros(POINT_LIMBS[curve], dtype=np.uint64)
    lib().fo_multiply(curve, _ptr(point), _ptr(scalar), _ptr(r))
    return r


def batch_mul(curve, scalars, points, nthreads=1, L=None):
    scalars, points = _u64(scalars), _u64(points)
    n = scalars.size // 4
    out = np.zeros((n, POINT_LIMBS[curve]), dtype=np.uint64)
    (L or lib()).fo_batch_mul(curve, _ptr(scalars), _ptr(points), _ptr(out), n, nthreads)
    return out


def batch_mul_fixed(curve, scalars, base, nthreads=1, L=None):
    scalars, base = _u64(scalars), _u64(base)
    n = scalars.size // 4
    out = np.zeros((n, POINT_LIMBS[curve]), dtype=np.uint64)
    (L or lib()).fo_batch_mul_fixed(curve, _ptr(scalars), _ptr(base), _ptr(out), n, nthreads)
    return out


def batch_double_mul(curve, u1, u2, q, nthreads=1, L=None):
    u1, u2, q = _u64(u1), _u64(u2), _u64(q)
    n = u1.size // 4
    out = np.zeros((n, POINT_LIMBS[curve]), dtype=np.uint64)
    (L or lib()).fo_batch_double_mul(curve, _ptr(u1), _ptr(u2), _ptr(q), _ptr(out), n, nthreads)
    return out


def batch_to_affine(curve, points, nthreads=1):
    points = _u64(points)
    n = points.size // POINT_LIMBS[curve]
    xy = np.zeros((n, 8), dtype=np.uint64)
    inf = np.zeros(n, dtype=np.uint8)
    lib().fo_batch_to_affine(curve, _ptr(points), _ptr(xy), _ptr(inf), n, nthreads)
    return xy, inf


def secp256k1_scalar_op(op, a, b=None):
    a = _u64(a)
    b = _u64(b if b is not None else [0, 0, 0, 0])
    r = np.zeros(4, dtype=np.uint64)
    rc = lib().fo_secp256k1_scalar_op(op.encode(), _ptr(a), _ptr(b), _ptr(r))
    if rc < 0:
        raise ValueError("fo_secp256k1_scalar_op rc=%d" % rc)
    return r, rc == 0


def batch_secp256k1_ecdsa_verify(digests, r, s, pk_xy, pk_inf=None, nthreads=1):
    """digests (n,32) uint8 big-endian; r, s (n,4); pk_xy (n,8) raw field limbs; -> (n,) uint8 status."""
    digests = np.ascontiguousarray(np.asarray(digests, dtype=np.uint8)).reshape(-1, 32)
    r, s, pk_xy = _u64(r), _u64(s), _u64(pk_xy)
    n = digests.shape[0]
    inf = np.ascontiguousarray(np.asarray(pk_inf, dtype=np.uint8)) if pk_inf is not None else None
    out = np.zeros(n, dtype=np.uint8)
    lib().fo_batch_secp256k1_ecdsa_verify(_ptr(digests), _ptr(r), _ptr(s), _ptr(pk_xy),
                                          _ptr(inf) if inf is not None else None, _ptr(out), n, nthreads)
    return out


def p256_scalar_op(op, a, b=None):
    """P-256 Scalar Mul ('mul', p256.rs:1409-1432) or invert ('inv', 1057-1080): -> (limbs, ok)."""
    a = _u64(a)
    b = _u64(b if b is not None else [0, 0, 0, 0])
    r = np.zeros(4, dtype=np.uint64)
    rc = lib().fo_p256_scalar_op(op.encode(), _ptr(a), _ptr(b), _ptr(r))
    if rc < 0:
        raise ValueError("fo_p256_scalar_op rc=%d" % rc)
    return r, rc == 0


def batch_p256_ecdsa_verify(digests, r, s, pk_xy, pk_inf=None, nthreads=1):
    """As batch_secp256k1_ecdsa_verify, for Ecdsa::<P256, D>::verify."""
    digests = np.ascontiguousarray(np.asarray(digests, dtype=np.uint8)).reshape(-1, 32)
    r, s, pk_xy = _u64(r), _u64(s), _u64(pk_xy)
    n = digests.shape[0]
    inf = np.ascontiguousarray(np.asarray(pk_inf, dtype=np.uint8)) if pk_inf is not None else None
    out = np.zeros(n, dtype=np.uint8)
    lib().fo_batch_p256_ecdsa_verify(_ptr(digests), _ptr(r), _ptr(s), _ptr(pk_xy),
                                     _ptr(inf) if inf is not None else None, _ptr(out), n, nthreads)
    return out


def ecdsa_batch_verify(curve, digests, r, s, pk_xy, pk_inf, a):
    """Ecdsa::<C, D>::batch_verify (ecdsa.rs:287-391), curve 0 / 1: -> (status, detail (16,) uint64 = r_sum,
    r_scalar_sum)."""
    digests = np.ascontiguousarray(np.asarray(digests, dtype=np.uint8)).reshape(-1, 32)
    r, s, pk_xy, a = _u64(r), _u64(s), _u64(pk_xy), _u64(a)
    n = digests.shape[0]
    inf = np.ascontiguousarray(np.asarray(pk_inf, dtype=np.uint8)) if pk_inf is not None else None
    detail = np.zeros(16, dtype=np.uint64)
    rc = lib().fo_ecdsa_batch_verify(curve, _ptr(digests), _ptr(r), _ptr(s), _ptr(pk_xy),
                                     _ptr(inf) if inf is not None else None, _ptr(a), n, _ptr(detail))
    if rc < 0:
        raise ValueError("fo_ecdsa_batch_verify rc=%d" % rc)
    return rc, detail


def batch_validate_point(curve, xy, inf=None, nthreads=1):
    """Curve::validate_point per affine point: (n,) uint8, 1 valid / 0 not."""
    xy = _u64(xy)
    n = xy.size // 8
    fl = np.ascontiguousarray(np.asarray(inf, dtype=np.uint8)) if inf is not None else None
    ok = np.zeros(n, dtype=np.uint8)
    lib().fo_batch_validate_point(curve, _ptr(xy), _ptr(fl) if fl is not None else None, _ptr(ok), n, nthreads)
    return ok


def batch_ecdh(curve, sk, pk_xy, pk_inf=None, nthreads=1):
    """KeyExchange::derive_shared_secret per element (curve 0 / 1): -> (secrets (n,32) uint8, status (n,) uint8)."""
    sk, pk_xy = _u64(sk), _u64(pk_xy)
    n = sk.size // 4
    inf = np.ascontiguousarray(np.asarray(pk_inf, dtype=np.uint8)) if pk_inf is not None else None
    out = np.zeros((n, 32), dtype=np.uint8)
    st = np.zeros(n, dtype=np.uint8)
    lib().fo_batch_ecdh(curve, _ptr(sk), _ptr(pk_xy), _ptr(inf) if inf is not None else None, _ptr(out), _ptr(st), n, nthreads)
    return out, st


def batch_ed25519_eddsa_verify(r_xy, r_inf, pk_xy, pk_inf, s, k, nthreads=1):
    """Eddsa verify from the point computation on: r_xy, pk_xy (n,8) raw limbs, flags (n,) or None, s, k (n,4)
    -> (n,) uint8 status (1 true, 0 false, 2 the reference panics)."""
    r_xy, pk_xy, s, k = _u64(r_xy), _u64(pk_xy), _u64(s), _u64(k)
    n = s.size // 4
    ri = np.ascontiguousarray(np.asarray(r_inf, dtype=np.uint8)) if r_inf is not None else None
    pi = np.ascontiguousarray(np.asarray(pk_inf, dtype=np.uint8)) if pk_inf is not None else None
    out = np.zeros(n, dtype=np.uint8)
    lib().fo_batch_ed25519_eddsa_verify(_ptr(r_xy), _ptr(ri) if ri is not None else None, _ptr(pk_xy),
                                        _ptr(pi) if pi is not None else None, _ptr(s), _ptr(k), _ptr(out), n, nthreads)
    return out


def secp256k1_schnorr_batch_verify(pk_xy, pk_inf, r_xy, r_inf, s, a, e):
    """-> (result, sides (16,) uint64, sides_inf (2,) uint8)."""
    pk_xy, r_xy, s, a, e = _u64(pk_xy), _u64(r_xy), _u64(s), _u64(a), _u64(e)
    n = s.size // 4
    pi = np.ascontiguousarray(np.asarray(pk_inf, dtype=np.uint8)) if pk_inf is not None else None
    ri = np.ascontiguousarray(np.asarray(r_inf, dtype=np.uint8)) if r_inf is not None else None
    sides = np.zeros(16, dtype=np.uint64)
    sinf = np.zeros(2, dtype=np.uint8)
    rc = lib().fo_secp256k1_schnorr_batch_verify(_ptr(pk_xy), _ptr(pi) if pi is not None else None, _ptr(r_xy),
                                                 _ptr(ri) if ri is not None else None, _ptr(s), _ptr(a), _ptr(e),
                                                 n, _ptr(sides), _ptr(sinf))
    return rc, sides, sinf


def schnorr_batch_verify(curve, pk_xy, pk_inf, r_xy, r_inf, s, a, e):
    """schnorr::batch_verify::<C, D> for curve 0 (secp256k1) / 1 (P-256): -> (result, sides (16,), sides_inf (2,))."""
    if curve == SECP256K1:
        return secp256k1_schnorr_batch_verify(pk_xy, pk_inf, r_xy, r_inf, s, a, e)
    if curve != P256:
        raise ValueError("schnorr batch_verify: curve 0 or 1")
    pk_xy, r_xy, s, a, e = _u64(pk_xy), _u64(r_xy), _u64(s), _u64(a), _u64(e)
    n = s.size // 4
    pi = np.ascontiguousarray(np.asarray(pk_inf, dtype=np.uint8)) if pk_inf is not None else None
    ri = np.ascontiguousarray(np.asarray(r_inf, dtype=np.uint8)) if r_inf is not None else None
    sides = np.zeros(16, dtype=np.uint64)
    sinf = np.zeros(2, dtype=np.uint8)
    rc = lib().fo_p256_schnorr_batch_verify(_ptr(pk_xy), _ptr(pi) if pi is not None else None, _ptr(r_xy),
                                            _ptr(ri) if ri is not None else None, _ptr(s), _ptr(a), _ptr(e), n,
                                            _ptr(sides), _ptr(sinf))
    return rc, sides, sinf


def ed25519_schnorr_batch_verify(pk_xy, pk_inf, r_xy, r_inf, s, a, e):
    """schnorr::batch_verify::<Ed25519, D> under the release profile (wrapping u128 sums in the scalar Mul):
    -> (result 0 / 1 / 2 = the reference panics in to_affine, sides (16,), sides_inf (2,), debug_build_panics 0 / 1)."""
    pk_xy, r_xy, s, a, e = _u64(pk_xy), _u64(r_xy), _u64(s), _u64(a), _u64(e)
    n = s.size // 4
    pi = np.ascontiguousarray(np.asarray(pk_inf, dtype=np.uint8)) if pk_inf is not None else None
    ri = np.ascontiguousarray(np.asarray(r_inf, dtype=np.uint8)) if r_inf is not None else None
    sides = np.zeros(16, dtype=np.uint64)
    sinf = np.zeros(2, dtype=np.uint8)
    dbg = np.zeros(1, dtype=np.uint8)
    rc = lib().fo_ed25519_schnorr_batch_verify(_ptr(pk_xy), _ptr(pi) if pi is not None else None, _ptr(r_xy),
                                               _ptr(ri) if ri is not None else None, _ptr(s), _ptr(a), _ptr(e), n,
                                               _ptr(sides), _ptr(sinf), _ptr(dbg))
    return rc, sides, sinf, int(dbg[0])


def ed25519_scalar_mul_release(a, b):
    """impl Mul for Scalar (ed25519.rs:1256-1376), release profile: -> (product (4,) uint64, overflowed 0 / 1)."""
    a, b = _u64(a), _u64(b)
    out = np.zeros(4, dtype=np.uint64)
    ovf = lib().fo_ed25519_scalar_mul_release(_ptr(a), _ptr(b), _ptr(out))
    return out, int(ovf)


def batch_schnorr_verify(curve, pk_xy, pk_inf, r_xy, r_inf, s, e, nthreads=1):
    """Schnorr::<C, D>::verify per signature (schnorr.rs:90-140) from the point computation on: (n,) uint8 status
    (1 true, 0 false, 2 = the reference panics)."""
    pk_xy, r_xy, s, e = _u64(pk_xy), _u64(r_xy), _u64(s), _u64(e)
    n = s.size // 4
    pi = np.ascontiguousarray(np.asarray(pk_inf, dtype=np.uint8)) if pk_inf is not None else None
    ri = np.ascontiguousarray(np.asarray(r_inf, dtype=np.uint8)) if r_inf is not None else None
    out = np.zeros(n, dtype=np.uint8)
    lib().fo_batch_schnorr_verify(curve, _ptr(pk_xy), _ptr(pi) if pi is not None else None, _ptr(r_xy),
                                  _ptr(ri) if ri is not None else None, _ptr(s), _ptr(e), _ptr(out), n, nthreads)
    return out


def batch_compress(curve, xy, inf=None):
    """PointAffine::to_bytes of each (x, y, infinity): (n, 33) uint8."""
    xy = _u64(xy)
    n = xy.size // 8
    fl = np.ascontiguousarray(np.asarray(inf, dtype=np.uint8)) if inf is not None else None
    out = np.zeros((n, 33), dtype=np.uint8)
    lib().fo_batch_compress(curve, _ptr(xy), _ptr(fl) if fl is not None else None, _ptr(out), n)
    return out


def _bytes(a, width):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint8)).reshape(-1, width)


def batch_decompress(curve, data33):
    """PointAffine::from_bytes per element -> (xy (n,8), inf (n,), ok (n,))."""
    b = _bytes(data33, 33)
    n = b.shape[0]
    xy, inf, ok = np.zeros((n, 8), dtype=np.uint64), np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
    lib().fo_batch_decompress(curve, _ptr(b), _ptr(xy), _ptr(inf), _ptr(ok), n)
    return xy, inf, ok


def batch_encode_uncompressed(curve, xy, inf=None):
    p = _u64(xy).reshape(-1, 8)
    n = p.shape[0]
    fl = np.ascontiguousarray(np.asarray(inf, dtype=np.uint8)) if inf is not None else None
    out = np.zeros((n, 65), dtype=np.uint8)
    lib().fo_batch_encode_uncompressed(curve, _ptr(p), _ptr(fl) if fl is not None else None, _ptr(out), n)
    return out


def batch_decode_uncompressed(curve, data65):
    b = _bytes(data65, 65)
    n = b.shape[0]
    xy, inf, ok = np.zeros((n, 8), dtype=np.uint64), np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
    lib().fo_batch_decode_uncompressed(curve, _ptr(b), _ptr(xy), _ptr(inf), _ptr(ok), n)
    return xy, inf, ok
