"""
Point decoding and the uncompressed encoding on the GPU (fec_batch_decompress,
fec_batch_decode_uncompressed, fec_batch_encode_uncompressed) against the committed fixtures
(tests/golden/decode_vectors.json, from the Python model) and against the C oracle on seeded inputs.
Bit-exact, including WHICH inputs the reference rejects (None).
"""
import json
import os

import numpy as np
import pytest

import vectors as V

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _cases():
    with open(os.path.join(HERE, "golden", "decode_vectors.json")) as f:
        return json.load(f)["cases"]


def _same(got, want, what):
    for g, w, name in zip(got, want, ("xy", "inf", "ok")):
        if not np.array_equal(g, w):
            bad = np.nonzero((g != w).reshape(g.shape[0], -1).any(axis=1))[0]
            raise AssertionError("%s: %s differs in %d rows, first %d: got %s want %s" % (what, name, len(bad), bad[0], g[bad[0]], w[bad[0]]))


@pytest.mark.parametrize("curve", [0, 1, 2])
def test_decode_fixtures_on_the_gpu(gpu_ctx, curve):
    cases = _cases()
    for op, width, fn in (("decompress", 33, gpu_ctx.batch_decompress),
                          ("decode_uncompressed", 65, gpu_ctx.batch_decode_uncompressed)):
        cs = [c for c in cases if c["curve"] == curve and c["op"] == op]
        data = np.frombuffer(bytes.fromhex("".join(c["in"] for c in cs)), dtype=np.uint8).reshape(-1, width)
        xy, inf, ok = fn(curve, data)
        want_xy = np.array([c["x"] + c["y"] for c in cs], dtype=np.uint64)
        _same((xy, inf, ok), (want_xy, np.array([c["inf"] for c in cs], dtype=np.uint8),
                              np.array([c["ok"] for c in cs], dtype=np.uint8)), "%s curve %d" % (op, curve))
    cs = [c for c in cases if c["curve"] == curve and c["op"] == "encode_uncompressed"]
    out = gpu_ctx.batch_encode_uncompressed(curve, np.array([c["x"] + c["y"] for c in cs], dtype=np.uint64),
                                            np.array([c["inf"] for c in cs], dtype=np.uint8))
    for i, c in enumerate(cs):
        assert out[i].tobytes().hex() == c["out"], (curve, i)


@pytest.mark.parametrize("curve", [0, 1, 2])
def test_decode_matches_oracle_on_seeded_inputs(gpu_ctx, oracle, curve):
    rng = np.random.default_rng(900 + curve)
    n = 3000
    top = 32 if curve == 2 else 1          # position of the most significant byte of x
    raw = rng.integers(0, 256, size=(n, 33), dtype=np.uint8)
    raw[:, 0] = rng.choice([0, 2, 3, 2, 3, 4, 7], size=n)
    raw[::2, top] &= 0x3F                  # half of the x values below p
    raw[5::11, 1:] = 0                     # x = 0
    raw[7::13, 1:] = 0xFF                  # x = 2^256 - 1
    _same(gpu_ctx.batch_decompress(curve, raw), oracle.batch_decompress(curve, raw), "decompress curve %d" % curve)
    # uncompressed: encodings of real (reference-arithmetic) affine points, garbage, identities
    q = V.points(600, curve, 910)
    xy, inf = gpu_ctx.batch_to_affine(curve, q)
    enc = gpu_ctx.batch_encode_uncompressed(curve, xy, inf)
    assert np.array_equal(enc, oracle.batch_encode_uncompressed(curve, xy, inf))
    junk = rng.integers(0, 256, size=(600, 65), dtype=np.uint8)
    junk[:, 0] = rng.choice([0, 4, 4, 4, 5], size=600)
    junk[::2, top] &= 0x3F
    junk[::2, 32 + top] &= 0x3F
    both = np.concatenate([enc, junk])
    _same(gpu_ctx.batch_decode_uncompressed(curve, both), oracle.batch_decode_uncompressed(curve, both),
          "decode_uncompressed curve %d" % curve)
    # ragged and empty batches
    for m in (0, 1, 63, 257):
        got = gpu_ctx.batch_decompress(curve, raw[:m])
        _same(got, tuple(a[:m] for a in oracle.batch_decompress(curve, raw)), "ragged %d" % m)


def test_identity_round_trips(gpu_ctx):
    for curve in (0, 1, 2):
        xy = np.zeros((3, 8), dtype=np.uint64)
        inf = np.ones(3, dtype=np.uint8)
        enc = gpu_ctx.batch_encode_uncompressed(curve, xy, inf)
        assert not enc.any()
        dxy, dinf, ok = gpu_ctx.batch_decode_uncompressed(curve, enc)
        assert ok.all() and dinf.all() and not dxy.any()
        c33 = gpu_ctx.batch_compress(curve, xy, inf)
        dxy, dinf, ok = gpu_ctx.batch_decompress(curve, c33)
        assert ok.all() and dinf.all() and not dxy.any()
