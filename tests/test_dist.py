"""CPU: the multi-GPU sharding/gather path rehearsed with world_size-2 gloo process groups.

The local solve on each rank is the ORACLE here (tests may use it as a stand-in; on GPUs the same code path
runs HipDensePrimalNormalSolver.solve_device)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pycllp_amd import problems
from pycllp_amd.dist import shard_slices, gather_batch, solve_sharded


def test_shard_slices_cover_the_batch():
    for B, W in ((10, 2), (7, 3), (3, 8), (65536, 8), (0, 4)):
        sl = shard_slices(B, W)
        assert len(sl) == W and sl[0][0] == 0 and sl[-1][1] == B
        assert all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
        sizes = [b - a for a, b in sl]
        assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, B, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import port as oport
        m, n = 6, 9
        A, b, c = problems.random_dense_arrays(m, n, B, seed=5)
        Ae, be, ce = problems.equality_arrays(A, b, c)
        sl = shard_slices(B, world)
        sizes = [e - s for s, e in sl]
        lo, hi = sl[rank]

        def local_solve(bl, cl):
            r = oport.dense_solve(Ae, bl.numpy(), cl.numpy())
            return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in r.items()}

        res, allg = solve_sharded(local_solve, torch.from_numpy(be[lo:hi]), torch.from_numpy(ce[lo:hi]), sizes)
        _, rootg = solve_sharded(local_solve, torch.from_numpy(be[lo:hi]), torch.from_numpy(ce[lo:hi]), sizes,
                                 fields=("pobj", "status"), dst=0)
        assert (rootg is None) == (rank != 0)
        if rank == 0:
            full = oport.dense_solve(Ae, be, ce)
            ok = all(np.array_equal(allg[k].numpy(), full[k]) for k in ("x", "y", "pobj", "dobj", "status", "iters"))
            ok = ok and np.array_equal(rootg["pobj"].numpy(), full["pobj"]) and allg["x"].shape == (B, n + m)
            open(tmp, "w").write("ok" if ok else "mismatch")
        # ragged plain gather
        t = torch.full((sizes[rank], 2), float(rank))
        g = gather_batch(t, sizes)
        assert g.shape == (B, 2) and float(g[0, 0]) == 0.0 and float(g[-1, 0]) == world - 1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [8, 7])
def test_sharded_solve_two_ranks_gloo(tmp_path, B):
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, _free_port(), B, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def _packed_worker(rank, world, port, B, tmp):
    """The collective bench.py times at N > 1: PackedGather over packed result buffers (oracle as the local solve),
    double-buffered over several steps, ragged shards included."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import port as oport
        from pycllp_amd.dist import PackedGather
        from pycllp_amd.solvers.hip import pack_layout, unpack
        m, n = 5, 7
        N = n + m
        A, b, c = problems.random_dense_arrays(m, n, B, seed=9)
        Ae, be, ce = problems.equality_arrays(A, b, c)
        sl = shard_slices(B, world)
        sizes = [e - s for s, e in sl]
        lo, hi = sl[rank]
        layout = pack_layout(max(sizes), m, N)
        assert layout["_gather_bytes"] % 256 == 0 and layout["x"][0] + layout["x"][1] <= layout["_gather_bytes"]
        pg = PackedGather(layout, world, rank, torch.device("cpu"), sizes=sizes)
        ok = True
        for step in range(3):                      # three steps through two slots: slot 0 is reused
            scale = 1.0 + step                     # a different batch every step
            r = oport.dense_solve(Ae, be[lo:hi] * scale, ce[lo:hi])
            packed = torch.zeros(layout["_total_bytes"], dtype=torch.uint8)
            views = unpack(packed, layout, names=("pobj", "dobj", "status", "iters", "y", "x", "z"))
            for k in ("pobj", "dobj", "status", "iters", "y", "x", "z"):
                views[k][:sizes[rank]] = torch.from_numpy(np.ascontiguousarray(r[k]))
            pg.post(step, packed)
            if step >= 1:                          # results of the PREVIOUS step are complete while this one is in flight
                g = pg.results(step - 1)
                if rank == 0:
                    full = oport.dense_solve(Ae, be * float(step), ce)
                    ok = ok and all(np.array_equal(g[k].numpy(), full[k]) for k in ("x", "y", "pobj", "dobj", "status", "iters"))
                    ok = ok and g["x"].shape == (B, N) and g["status"].dtype == torch.int32
                else:
                    ok = ok and g is None
        pg.wait()
        g = pg.results(2)
        if rank == 0:
            full = oport.dense_solve(Ae, be * 3.0, ce)
            ok = ok and np.array_equal(g["pobj"].numpy(), full["pobj"]) and np.array_equal(g["x"].numpy(), full["x"])
            open(tmp, "w").write("ok" if ok else "mismatch")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [8, 7])
def test_packed_result_gather_two_ranks_gloo(tmp_path, B):
    out = str(tmp_path / "result.txt")
    mp.spawn(_packed_worker, args=(2, _free_port(), B, out), nprocs=2, join=True)
    assert open(out).read() == "ok"
