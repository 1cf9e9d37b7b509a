"""forge_ec_amd -- MI355X (gfx950) backend for forge-ec's batched scalar-multiplication path.

The compute path is libfecgpu.so (hand-written HIP, C ABI in include/fecgpu.h).  This package is
the host-side mirror of the reference's trait surface over that ABI; it has no CPU fallback and
never imports the oracle.
"""
from ._lib import ED25519, P256, SECP256K1, POINT_LIMBS, FecError, lib  # noqa: F401
from .curves import CURVES, Context, Ed25519Curve, P256Curve, Secp256k1  # noqa: F401
