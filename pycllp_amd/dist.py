"""Multi-GPU sharding of the LP batch: one process per GPU, no data-path collective.

The batch of LPs (fixed A, varying b and c) is embarrassingly parallel: rank r owns a contiguous slice of
the batch axis, A is replicated (every rank builds it from the same data / seed), each rank runs the same
single-GPU solve, and ONE collective at the end gathers the results (RCCL over xGMI when the process
group's backend is "nccl"; "gloo" on CPU for the tests).  The reference has no counterpart (it is single
device: pycllp/solvers/cl.py uses one queue); this is the north_star's "RCCL used only for the final
result gather".
"""
import torch
import torch.distributed as dist


def shard_slices(nproblems, world_size):
    """Contiguous [start, stop) per rank; sizes differ by at most one."""
    base, extra = divmod(int(nproblems), int(world_size))
    out, start = [], 0
    for r in range(world_size):
        size = base + (1 if r < extra else 0)
        out.append((start, start + size))
        start += size
    return out


def gather_batch(local, sizes, group=None, dst=None):
    """Concatenate per-rank tensors (first dim = local batch, possibly ragged) in rank order.

    ``sizes``: local batch size of every rank.  ``dst=None`` -> all ranks get the result (all_gather);
    otherwise only rank ``dst`` does (others return None)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return local
    mx = max(sizes)
    if local.shape[0] != sizes[rank]:
        raise ValueError("rank %d holds %d rows, expected %d" % (rank, local.shape[0], sizes[rank]))
    send = local
    if local.shape[0] < mx:
        pad = torch.zeros((mx - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send = torch.cat([local, pad], dim=0)
    send = send.contiguous()
    if dst is None:
        out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, send, group=group)
        parts = out.view((world, mx) + tuple(local.shape[1:]))
    else:
        bufs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
        dist.gather(send, bufs, dst=dst, group=group)
        if rank != dst:
            return None
        parts = bufs
    if all(s == mx for s in sizes):
        return parts.reshape((world * mx,) + tuple(local.shape[1:])) if dst is None else torch.cat(parts, dim=0)
    return torch.cat([parts[r][:sizes[r]] for r in range(world)], dim=0)


def solve_sharded(local_solve, b_local, c_local, sizes, group=None, fields=("pobj", "dobj", "status", "iters", "x", "y"),
                  dst=None):
    """Run ``local_solve(b_local, c_local) -> dict of tensors`` on this rank's shard, then gather ``fields``.

    Returns (local_result, gathered) where ``gathered`` is a dict on the receiving rank(s), else None."""
    res = local_solve(b_local, c_local)
    gathered = {}
    for k in fields:
        g = gather_batch(res[k], sizes, group=group, dst=dst)
        if g is not None:
            gathered[k] = g
    return res, (gathered if gathered else None)


class PackedGather(object):
    """The path's one collective as ``bench.py`` and a multi-GPU caller issue it: every rank holds the results of a solve
    in ONE packed device allocation (``pycllp_amd.solvers.hip.pack_layout``: pobj | dobj | status | iters | y | x | z), and the
    first ``gather_bytes`` bytes of it travel to rank ``dst`` in a single ``dist.gather``.  The gather is asynchronous and
    double-buffered: ``post(k, packed)`` uses slot ``k % slots`` -- receive buffers on ``dst`` included -- after waiting
    for the gather that last used the slot, so the transfer of step k overlaps the solve of step k+1.

    Ragged shards: build the layout for ``max(sizes)`` LPs on every rank; ``results(k)`` slices rank r's arrays to ``sizes[r]``.
    """

    def __init__(self, layout, world_size, rank, device, dst=0, group=None, slots=2, sizes=None):
        self.layout, self.world, self.rank, self.dst, self.group, self.slots = layout, world_size, rank, dst, group, slots
        self.nbytes = int(layout["_gather_bytes"])
        self.sizes = list(sizes) if sizes is not None else None
        self.recv = None
        if rank == dst:
            self.recv = [[torch.empty(self.nbytes, dtype=torch.uint8, device=device) for _ in range(world_size)]
                         for _ in range(slots)]
        self.pending = [None] * slots

    def wait(self, k=None):
        """Block until the gather of slot ``k % slots`` (all slots when ``k`` is None) has completed."""
        for sl in (range(self.slots) if k is None else [k % self.slots]):
            if self.pending[sl] is not None:
                self.pending[sl].wait()
                self.pending[sl] = None

    def post(self, k, packed, sync=False):
        """Gather the first ``gather_bytes`` bytes of ``packed`` (uint8, on the collective's device) to ``dst``."""
        sl = k % self.slots
        self.wait(sl)
        if packed.dtype != torch.uint8 or packed.numel() < self.nbytes:
            raise ValueError("packed must be a uint8 tensor of at least %d bytes" % self.nbytes)
        work = dist.gather(packed[:self.nbytes], self.recv[sl] if self.rank == self.dst else None, dst=self.dst,
                           group=self.group, async_op=True)
        if sync:
            work.wait()
        else:
            self.pending[sl] = work
        return work

    def results(self, k, names=("pobj", "dobj", "status", "iters", "y", "x")):
        """On ``dst``: the gathered arrays of step ``k`` concatenated in rank order (None elsewhere).  Waits for that gather."""
        self.wait(k)
        if self.rank != self.dst:
            return None
        from .solvers.hip import unpack
        parts = [unpack(buf, self.layout, names=names) for buf in self.recv[k % self.slots]]
        if self.sizes is not None:
            parts = [{f: p[f][:self.sizes[r]] for f in names} for r, p in enumerate(parts)]
        return {f: torch.cat([p[f] for p in parts], dim=0) for f in names}
