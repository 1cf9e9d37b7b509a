"""
Multi-GPU sharding for the batched scalar-multiplication path: one process per GPU,
torch.distributed (backend "nccl" = RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The path shards trivially -- scalar-muls are independent -- so there is NO collective on the
compute path.  The only exchange is the one BASELINE.json's north_star names: gathering the
result shards.  By default they are gathered TO THE CONSUMER rank (dist.gather: on RCCL a grouped
send/recv, i.e. every peer writes its block to rank `dst` over its own xGMI link -- the direct
pattern SURVEY.md section 8e asks for; a ring all-gather would make every rank receive 7 x 96 MiB per
step that only rank 0 reads).  dst=None selects the all-gather (every rank gets the batch).  The
collective is issued async so it overlaps the next batch's kernel.
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous shard [lo, hi) of n units for `rank` of `world` (sizes differ by at most 1)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return (n * rank) // world, (n * (rank + 1)) // world


def shard_sizes(n, world):
    return [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]


_GATHER_OK = {}


def _gather_supported(device, group, dst):
    """One tiny dist.gather at construction time (a collective: every rank of the group runs it): does this backend
    implement gather to one rank?  All ranks see the same answer, so they all pick the same collective afterwards."""
    key = (dist.get_backend(group), str(device).split(":")[0])
    if key not in _GATHER_OK:
        try:
            world, rank = dist.get_world_size(group), dist.get_rank(group)
            probe = torch.zeros(1, dtype=torch.int64, device=device)
            blocks = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)] if rank == dst else None
            dst_global = dst if group is None else dist.get_global_rank(group, dst)
            dist.gather(probe, gather_list=blocks, dst=dst_global, group=group)
            _GATHER_OK[key] = True
        except (RuntimeError, NotImplementedError):
            _GATHER_OK[key] = False
    return _GATHER_OK[key]


class ResultGather:
    """Gather of the per-rank result shards into the full (n, limbs) result on rank `dst`
    (default 0), or on every rank when dst is None.

    Shards may differ in size by one row; they are padded to the largest shard for the collective
    and trimmed afterwards.  `start()` enqueues the collective and returns immediately (async);
    `finish()` waits and returns the assembled tensor on the consumer rank(s), None elsewhere.
    """

    def __init__(self, n_total, limbs, device, dtype=torch.int64, group=None, dst=0):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if dst is not None and not (0 <= dst < self.world):
            raise ValueError("bad dst rank")
        self.dst = dst
        self.n_total = n_total
        self.limbs = limbs
        self.sizes = shard_sizes(n_total, self.world)
        self.pad = max(self.sizes)
        if dst is not None and not _gather_supported(device, group, dst):
            # (a backend build without gather: every rank then receives the batch -- more traffic, same result on rank dst)
            self.dst = dst = None
        self.consumer = dst is None or dst == self.rank
        # only a consumer holds the assembled batch; the other ranks own one send block
        self.buf = torch.empty((self.world * self.pad, limbs), dtype=dtype, device=device) if self.consumer else None
        self.send = torch.zeros((self.pad, limbs), dtype=dtype, device=device)
        self.handle = None

    def start(self, local):
        if local.shape[0] != self.sizes[self.rank]:
            raise ValueError("local shard has %d rows, expected %d" % (local.shape[0], self.sizes[self.rank]))
        if local.shape[0] == self.pad:
            src = local.contiguous()
        else:
            self.send[:local.shape[0]].copy_(local)
            src = self.send
        if self.dst is None:
            self.handle = dist.all_gather_into_tensor(self.buf, src, group=self.group, async_op=True)
        else:
            blocks = None
            if self.consumer:
                blocks = [self.buf[r * self.pad:(r + 1) * self.pad] for r in range(self.world)]
            dst_global = self.dst if self.group is None else dist.get_global_rank(self.group, self.dst)
            self.handle = dist.gather(src, gather_list=blocks, dst=dst_global, group=self.group, async_op=True)
        return self.handle

    def finish(self):
        if self.handle is not None:
            self.handle.wait()
            self.handle = None
        if not self.consumer:
            return None
        if all(s == self.pad for s in self.sizes):
            return self.buf
        parts = [self.buf[r * self.pad: r * self.pad + self.sizes[r]] for r in range(self.world)]
        return torch.cat(parts, dim=0)
