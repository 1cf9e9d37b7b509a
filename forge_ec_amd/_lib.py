"""ctypes binding of include/fecgpu.h.  No fallback: if libfecgpu.so is missing or no gfx950
GPU is usable, importing works but every compute call raises."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(HERE, "libfecgpu.so")

SECP256K1, P256, ED25519 = 0, 1, 2
POINT_LIMBS = {SECP256K1: 12, P256: 12, ED25519: 16}
F_ADD, F_SUB, F_MUL, F_SQR, F_NEG = 0, 1, 2, 3, 4
P_ADD, P_DOUBLE, P_NEGATE, P_DOUBLE_TRAIT = 0, 1, 2, 3

# every symbol include/fecgpu.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "fec_point_limbs", "fec_ctx_create", "fec_ctx_create_multi", "fec_ctx_device_count", "fec_ctx_destroy", "fec_ctx_wipe", "fec_ctx_check", "fec_ctx_debug_force_fault", "fec_ctx_set_fixed_prefix_bits", "fec_ctx_build_fixed_prefix", "fec_ctx_set_fixed_prefix_after", "fec_ctx_set_fixed_prefix_budget", "fec_ctx_fixed_prefix_bits", "fec_ctx_set_side_stream_max", "fec_generator", "fec_generator_dev", "fec_batch_mul", "fec_batch_mul_fixed",
    "fec_batch_double_mul", "fec_batch_to_affine", "fec_batch_to_affine_dev", "fec_ecdsa_verify_secp256k1", "fec_ecdsa_verify_secp256k1_dev", "fec_ecdsa_verify_p256", "fec_ecdsa_verify_p256_dev", "fec_eddsa_verify_ed25519", "fec_eddsa_verify_ed25519_dev", "fec_ecdsa_batch_verify", "fec_batch_ecdh", "fec_batch_ecdh_dev", "fec_batch_validate_point", "fec_batch_validate_point_dev", "fec_multi_scalar_mul", "fec_schnorr_batch_verify_secp256k1", "fec_schnorr_batch_verify", "fec_schnorr_batch_verify_ed25519", "fec_schnorr_verify", "fec_schnorr_verify_dev", "fec_batch_compress", "fec_batch_compress_dev", "fec_batch_decompress", "fec_batch_encode_uncompressed", "fec_batch_decode_uncompressed", "fec_field_op", "fec_point_op", "fec_batch_mul_dev",
    "fec_batch_mul_fixed_dev", "fec_batch_double_mul_dev", "fec_multi_batch_mul_dev", "fec_multi_batch_mul_fixed_dev", "fec_multi_batch_double_mul_dev", "fec_ctx_set_chunk", "fec_ctx_set_timing",
    "fec_ctx_last_kernel_ms", "fec_measure_peak_mad32", "fec_ctx_device_info", "fec_strerror",
]
# include/fecgpu_canon.h: the canonical-math mode (NOT reference parity)
CANON_ABI_SYMBOLS = [
    "fec_canon_mul_base", "fec_canon_mul_base_dev", "fec_canon_mul", "fec_canon_mul_dev", "fec_canon_field_op",
    "fec_canon_double_mul", "fec_canon_double_mul_dev", "fec_canon_ecdsa_verify", "fec_canon_ecdsa_verify_dev",
    "fec_canon_bip340_verify", "fec_canon_bip340_verify_dev", "fec_canon_eddsa_verify", "fec_canon_eddsa_verify_dev",
    "fec_canon_scalar_op",
]
F_INV = 5


class FecError(RuntimeError):
    def __init__(self, status, what=""):
        self.status = status
        msg = "fecgpu error %d" % status
        try:
            msg += ": " + lib().fec_strerror(status).decode()
        except Exception:
            pass
        if what:
            msg += " (%s)" % what
        super().__init__(msg)


_lib = None


def lib():
    """Load libfecgpu.so.  Raises (loudly) when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(
            "forge_ec_amd: %s is missing -- build it with `python -m forge_ec_amd.build` "
            "(there is no CPU fallback)" % SO_PATH)
    # PyTorch (device memory / streams / torch.distributed plumbing) bundles its own HIP runtime
    # under the same SONAME.  Device pointers and streams handed to the *_dev entry points must
    # come from the runtime instance libfecgpu.so itself uses, so when torch is importable it is
    # loaded first and libfecgpu.so binds to its libamdhip64; without torch the system ROCm
    # runtime is used.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = ctypes.CDLL(SO_PATH)
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    L.fec_point_limbs.argtypes = [ci]
    L.fec_point_limbs.restype = ci
    L.fec_ctx_create.argtypes = [ctypes.POINTER(vp), ci]
    L.fec_ctx_create.restype = ci
    L.fec_ctx_create_multi.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(ci), ci]
    L.fec_ctx_create_multi.restype = ci
    L.fec_ctx_device_count.argtypes = [vp]
    L.fec_ctx_device_count.restype = ci
    L.fec_ctx_wipe.argtypes = [vp]
    L.fec_ctx_wipe.restype = ci
    L.fec_ctx_check.argtypes = [vp]
    L.fec_ctx_check.restype = ci
    L.fec_ctx_debug_force_fault.argtypes = [vp, ci]
    L.fec_ctx_debug_force_fault.restype = ci
    L.fec_ctx_set_fixed_prefix_bits.argtypes = [vp, ctypes.c_uint]
    L.fec_ctx_set_fixed_prefix_bits.restype = ci
    L.fec_ctx_fixed_prefix_bits.argtypes = [vp, ci]
    L.fec_ctx_fixed_prefix_bits.restype = ci
    L.fec_ctx_build_fixed_prefix.argtypes = [vp, ci]
    L.fec_ctx_build_fixed_prefix.restype = ci
    L.fec_ctx_set_fixed_prefix_after.argtypes = [vp, sz]
    L.fec_ctx_set_fixed_prefix_after.restype = ci
    L.fec_ctx_set_fixed_prefix_budget.argtypes = [vp, ctypes.c_uint]
    L.fec_ctx_set_fixed_prefix_budget.restype = ci
    L.fec_ctx_set_side_stream_max.argtypes = [vp, sz]
    L.fec_ctx_set_side_stream_max.restype = ci
    # device-resident shards of a multi-device ctx: arrays of per-device pointers / counts
    L.fec_multi_batch_mul_dev.argtypes = [vp, ci, vp, vp, vp, vp, vp, ci, vp]
    L.fec_multi_batch_mul_dev.restype = ci
    L.fec_multi_batch_mul_fixed_dev.argtypes = [vp, ci, vp, vp, vp, vp, vp, ci, vp]
    L.fec_multi_batch_mul_fixed_dev.restype = ci
    L.fec_multi_batch_double_mul_dev.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, ci, vp]
    L.fec_multi_batch_double_mul_dev.restype = ci
    L.fec_ctx_destroy.argtypes = [vp]
    L.fec_ctx_destroy.restype = None
    L.fec_generator.argtypes = [vp, ci, vp]
    L.fec_generator.restype = ci
    L.fec_generator_dev.argtypes = [vp, ci]
    L.fec_generator_dev.restype = vp
    L.fec_batch_mul.argtypes = [vp, ci, vp, vp, vp, sz]
    L.fec_batch_mul_fixed.argtypes = [vp, ci, vp, vp, vp, sz]
    L.fec_batch_double_mul.argtypes = [vp, ci, vp, vp, vp, vp, sz]
    L.fec_batch_decompress.argtypes = [vp, ci, vp, vp, vp, vp, sz]
    L.fec_batch_decode_uncompressed.argtypes = [vp, ci, vp, vp, vp, vp, sz]
    L.fec_batch_encode_uncompressed.argtypes = [vp, ci, vp, vp, vp, sz]
    L.fec_field_op.argtypes = [vp, ci, ci, vp, vp, vp, sz]
    L.fec_multi_scalar_mul.argtypes = [vp, ci, vp, vp, vp, sz]
    L.fec_multi_scalar_mul.restype = ci
    L.fec_ecdsa_verify_secp256k1.argtypes = [vp, vp, vp, vp, vp, vp, vp, sz]
    L.fec_ecdsa_verify_secp256k1.restype = ci
    L.fec_ecdsa_verify_secp256k1_dev.argtypes = [vp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.fec_ecdsa_verify_secp256k1_dev.restype = ci
    L.fec_ecdsa_verify_p256.argtypes = [vp, vp, vp, vp, vp, vp, vp, sz]
    L.fec_ecdsa_verify_p256.restype = ci
    L.fec_ecdsa_verify_p256_dev.argtypes = [vp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.fec_ecdsa_verify_p256_dev.restype = ci
    L.fec_batch_validate_point.argtypes = [vp, ci, vp, vp, vp, sz]
    L.fec_batch_validate_point.restype = ci
    L.fec_batch_validate_point_dev.argtypes = [vp, ci, vp, vp, vp, sz, vp]
    L.fec_batch_validate_point_dev.restype = ci
    L.fec_batch_ecdh.argtypes = [vp, ci, vp, vp, vp, vp, vp, sz]
    L.fec_batch_ecdh.restype = ci
    L.fec_batch_ecdh_dev.argtypes = [vp, ci, vp, vp, vp, vp, vp, sz, vp]
    L.fec_batch_ecdh_dev.restype = ci
    L.fec_ecdsa_batch_verify.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, sz, vp, vp]
    L.fec_ecdsa_batch_verify.restype = ci
    L.fec_eddsa_verify_ed25519.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, sz]
    L.fec_eddsa_verify_ed25519.restype = ci
    L.fec_eddsa_verify_ed25519_dev.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.fec_eddsa_verify_ed25519_dev.restype = ci
    L.fec_schnorr_batch_verify_secp256k1.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, sz, vp, vp, vp]
    L.fec_schnorr_batch_verify_secp256k1.restype = ci
    L.fec_schnorr_batch_verify.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp, sz, vp, vp, vp]
    L.fec_schnorr_batch_verify.restype = ci
    L.fec_schnorr_batch_verify_ed25519.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, sz, vp, vp, vp, vp]
    L.fec_schnorr_batch_verify_ed25519.restype = ci
    L.fec_schnorr_verify.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp, sz]
    L.fec_schnorr_verify.restype = ci
    L.fec_schnorr_verify_dev.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.fec_schnorr_verify_dev.restype = ci
    L.fec_batch_compress.argtypes = [vp, ci, vp, vp, vp, sz]
    L.fec_batch_compress.restype = ci
    L.fec_batch_compress_dev.argtypes = [vp, ci, vp, vp, vp, sz, vp]
    L.fec_batch_compress_dev.restype = ci
    L.fec_batch_to_affine.argtypes = [vp, ci, vp, vp, vp, sz]
    L.fec_batch_to_affine.restype = ci
    L.fec_batch_to_affine_dev.argtypes = [vp, ci, vp, vp, vp, sz, vp]
    L.fec_batch_to_affine_dev.restype = ci
    L.fec_point_op.argtypes = [vp, ci, ci, vp, vp, vp, sz]
    L.fec_batch_mul_dev.argtypes = [vp, ci, vp, vp, vp, sz, vp]
    L.fec_batch_mul_fixed_dev.argtypes = [vp, ci, vp, vp, vp, sz, vp]
    L.fec_batch_double_mul_dev.argtypes = [vp, ci, vp, vp, vp, vp, sz, vp]
    for n in ("fec_batch_mul", "fec_batch_mul_fixed", "fec_batch_double_mul", "fec_batch_to_affine", "fec_batch_to_affine_dev", "fec_ecdsa_verify_secp256k1", "fec_ecdsa_verify_secp256k1_dev", "fec_multi_scalar_mul", "fec_field_op",
              "fec_point_op", "fec_batch_mul_dev", "fec_batch_mul_fixed_dev",
              "fec_batch_double_mul_dev"):
        getattr(L, n).restype = ci
    L.fec_ctx_set_chunk.argtypes = [vp, sz]
    L.fec_ctx_set_chunk.restype = ci
    L.fec_ctx_set_timing.argtypes = [vp, ci]
    L.fec_ctx_set_timing.restype = ci
    L.fec_ctx_last_kernel_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_char_p)]
    L.fec_ctx_last_kernel_ms.restype = ci
    L.fec_measure_peak_mad32.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
    L.fec_measure_peak_mad32.restype = ci
    L.fec_ctx_device_info.argtypes = [vp, ctypes.c_char_p, sz, ctypes.POINTER(ci), ctypes.POINTER(ci)]
    L.fec_ctx_device_info.restype = ci
    L.fec_strerror.argtypes = [ci]
    L.fec_strerror.restype = ctypes.c_char_p
    L.fec_canon_mul_base.argtypes = [vp, ci, vp, vp, vp, sz]
    L.fec_canon_mul_base_dev.argtypes = [vp, ci, vp, vp, vp, sz, vp]
    L.fec_canon_mul.argtypes = [vp, ci, vp, vp, vp, vp, sz]
    L.fec_canon_mul_dev.argtypes = [vp, ci, vp, vp, vp, vp, sz, vp]
    L.fec_canon_field_op.argtypes = [vp, ci, ci, vp, vp, vp, sz]
    L.fec_canon_double_mul.argtypes = [vp, ci, vp, vp, vp, vp, vp, sz]
    L.fec_canon_double_mul_dev.argtypes = [vp, ci, vp, vp, vp, vp, vp, sz, vp]
    L.fec_canon_ecdsa_verify.argtypes = [vp, ci, vp, vp, vp, vp, vp, sz]
    L.fec_canon_ecdsa_verify_dev.argtypes = [vp, ci, vp, vp, vp, vp, vp, sz, vp]
    L.fec_canon_scalar_op.argtypes = [vp, ci, ci, vp, vp, vp, vp, sz]
    for name in ("fec_canon_bip340_verify", "fec_canon_eddsa_verify"):
        getattr(L, name).argtypes = [vp, vp, vp, vp, vp, vp, sz]
        getattr(L, name + "_dev").argtypes = [vp, vp, vp, vp, vp, vp, sz, vp]
    for n in CANON_ABI_SYMBOLS:
        getattr(L, n).restype = ci
    _lib = L
    return L
