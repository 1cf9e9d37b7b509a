"""Rehearsal of bench.py's N > 1 control flow on a box WITHOUT GPUs: the same file, the same flags and the same
self-launch, with a CPU / gloo platform whose "context" computes with the oracle.

    python tests/bench_rehearsal.py --gpus 2 --steps 2 --warmup 1 --log2-batch 6 --gather both

runs bench.main() with that platform.  Without RANK in the environment bench.main() starts the ranks itself --
torch.distributed.run on THIS script -- exactly as `python bench.py --gpus N` does on a GPU node, so the launcher,
both timed regions, the double-buffered ResultGather, the max-reduce over ranks and the one JSON line on rank 0 are all
exercised with two real ranks.  Test infrastructure (it imports the oracle); bench.py knows nothing about it.
"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _view(ptr, rows, limbs):
    """The caller's (rows, limbs) uint64 array behind a raw address (CPU tensors: data_ptr() is host memory)."""
    if rows == 0:
        return np.zeros((0, limbs), dtype=np.uint64)
    buf = (ctypes.c_uint64 * (rows * limbs)).from_address(ptr)
    return np.ctypeslib.as_array(buf).reshape(rows, limbs)


class OracleContext:
    """The methods of forge_ec_amd.Context that bench.py calls, answered by the C oracle on host memory."""

    def __init__(self):
        from oracle import c_oracle
        import forge_ec_amd as F
        self.O, self.limbs = c_oracle, F.POINT_LIMBS
        self.gens = {c: np.ascontiguousarray(c_oracle.generator(c)) for c in (0, 1, 2)}
        self.bits = 0
        self.ms, self.timing = 0.0, False

    def generator_dev(self, curve):
        return self.gens[curve].ctypes.data

    def _timed(self, f):
        t0 = time.perf_counter()
        r = f()
        self.ms = (time.perf_counter() - t0) * 1e3
        return r

    def batch_mul_dev(self, curve, k, p, out, n, stream=None):
        L = self.limbs[curve]
        _view(out, n, L)[:] = self._timed(lambda: self.O.batch_mul(curve, _view(k, n, 4).copy(), _view(p, n, L).copy(), nthreads=2))

    def batch_mul_fixed_dev(self, curve, k, base, out, n, stream=None):
        L = self.limbs[curve]
        b = _view(base, 1, L)[0].copy()
        _view(out, n, L)[:] = self._timed(lambda: self.O.batch_mul_fixed(curve, _view(k, n, 4).copy(), b, nthreads=2))

    def batch_double_mul_dev(self, curve, u1, u2, q, out, n, stream=None):
        L = self.limbs[curve]
        _view(out, n, L)[:] = self._timed(lambda: self.O.batch_double_mul(curve, _view(u1, n, 4).copy(), _view(u2, n, 4).copy(),
                                                                         _view(q, n, L).copy(), nthreads=2))

    def set_fixed_prefix_bits(self, bits):
        self.bits = int(bits)

    def build_fixed_prefix(self, curve):
        return None

    def fixed_prefix_bits(self, curve):
        return self.bits

    def set_timing(self, on):
        self.timing = bool(on)

    def last_kernel_ms(self):
        return self.ms, "oracle (rehearsal: no GPU kernel ran)"

    def check(self):
        return None

    def measure_peak_mad32(self):
        return 1.0

    def device_info(self):
        return {"name": "CPU rehearsal (gloo, oracle as the compute)", "compute_units": 256}


class GlooOraclePlatform:
    backend = "gloo"

    def __init__(self, local_rank):
        import torch
        self.torch, self.local_rank = torch, local_rank
        self.device = torch.device("cpu")

    def init_process_group(self, dist):
        dist.init_process_group(backend=self.backend)

    def context(self):
        return OracleContext()

    def new_stream(self):
        return 1   # any non-null handle: the oracle context ignores it

    def synchronize(self):
        return None


if __name__ == "__main__":
    import bench
    bench.main(platform_factory=GlooOraclePlatform, script=os.path.abspath(__file__))
