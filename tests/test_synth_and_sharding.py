"""CPU tests of the host logic: seeded inputs are deterministic and canonical; the N>1 path
(contiguous sharding + all-gather of result shards) is exercised with gloo, world_size 2, the
oracle standing in for the per-rank compute."""
import os
import socket

import numpy as np
import pytest

import vectors as V


def test_synth_is_deterministic_and_canonical():
    for curve in (0, 1, 2):
        a = V.scalars(500, curve, 5)
        b = V.scalars(500, curve, 5)
        assert np.array_equal(a, b)
        assert not np.array_equal(a, V.scalars(500, curve, 6))
        bound = V.ORDER[curve] if curve != 2 else V.PRIME[2]
        assert all(0 < V.int_of(r) < bound for r in a)
        p = V.points(200, curve, 7)
        assert p.shape == (200, V.POINT_LIMBS[curve])
        for row in p:
            for c in range(V.POINT_LIMBS[curve] // 4):
                assert V.int_of(row[4 * c:4 * c + 4]) < V.PRIME[curve]


def test_shard_ranges_cover_exactly():
    from forge_ec_amd.dist import shard_range, shard_sizes
    for n in (0, 1, 7, 8, 1000, 1 << 20, (1 << 22) + 5):
        for world in (1, 2, 3, 8):
            edges = [shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            assert sum(shard_sizes(n, world)) == n
            assert max(shard_sizes(n, world)) - min(shard_sizes(n, world)) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, curve, q, dst="all"):
    import torch
    import torch.distributed as dist
    from forge_ec_amd.dist import ResultGather, shard_range
    from oracle import c_oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        k = V.scalars(n, curve, 601)
        p = V.points(n, curve, 602)
        lo, hi = shard_range(n, rank, world)
        local = c_oracle.batch_mul(curve, k[lo:hi], p[lo:hi])  # stand-in for the rank's GPU shard
        g = ResultGather(n, V.POINT_LIMBS[curve], torch.device("cpu"), dst=None if dst == "all" else dst)
        for _ in range(2):  # the object is reused step after step (bench.py double-buffers two of them)
            g.start(torch.from_numpy(local.view(np.int64)))
            full = g.finish()
        q.put((rank, None if full is None else full.numpy().view(np.uint64).copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,dst", [(10, "all"), (11, "all"), (10, 0), (11, 0), (11, 1), (16, 0)])
def test_two_rank_gloo_gather_reassembles_the_batch(oracle, n, dst):
    """world_size 2 over gloo: the shard + gather path of bench.py, in both forms -- gather to the
    consumer rank (the default: only rank `dst` receives the batch) and all-gather.  This IS the strong-scaling
    split of `bench.py --scaling strong`: ONE global batch of n elements (the same data whatever the world size),
    every rank keeps its contiguous dist.shard_range and the gathered batch must equal the unsharded result --
    ragged (n = 11) and even (n = 16 = 2^4, the shape of the two 8-GPU BASELINE configurations)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    curve = 0
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, curve, q, dst)) for r in range(2)]
    for pr in procs:
        pr.start()
    results = dict(q.get(timeout=120) for _ in range(2))
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    want = oracle.batch_mul(curve, V.scalars(n, curve, 601), V.points(n, curve, 602))
    for r in range(2):
        if dst == "all" or dst == r:
            assert np.array_equal(results[r], want)
        else:
            assert results[r] is None  # a non-consumer rank holds no copy of the batch
